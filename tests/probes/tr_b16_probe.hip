// What ds_read_b64_tr_b16 returns, established on the device (gfx950): within each 16-lane group, lane (4 r + q) passes the address of four consecutive 16-bit
// elements = row r, columns 4 q .. 4 q + 3 of a 4 x 16 matrix; the hypothesis checked here is that lane i gets back column i of that matrix (rows 0..3).
//   hipcc --offload-arch=gfx950 -O2 tests/probes/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void tr_probe(const unsigned short* in, unsigned short* out, const int* addr_elems)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int e = threadIdx.x; e < 8192; e += 64) lds[e] = in[e];
    __syncthreads();
    const int l = threadIdx.x;
    typedef __attribute__((address_space(3))) f16x4* lp;
    f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds + addr_elems[l]));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main()
{
    unsigned short h_in[8192], h_out[256]; int h_addr[64];
    for (int e = 0; e < 8192; ++e) h_in[e] = (unsigned short)e;
    for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; h_addr[l] = g * 1000 + (i >> 2) * 200 + (i & 3) * 4; }
    unsigned short *d_in, *d_out; int* d_addr;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out); hipMalloc(&d_addr, sizeof h_addr);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice); hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int want = (l >> 4) * 1000 + j * 200 + (l & 15);
            if (h_out[l * 4 + j] != want) ++bad;
        }
    printf("tr_b16 probe: %d of 256 values differ from 'lane i gets column i of the group's 4 x 16 matrix'\n", bad);
    for (int l = 0; l < 64; l += 5) printf("  lane %2d: %5d %5d %5d %5d\n", l, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    return bad != 0;
}
