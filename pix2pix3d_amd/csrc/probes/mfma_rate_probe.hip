// What does the gfx950 matrix pipe sustain when a kernel issues NOTHING but v_mfma_f32_32x32x16_f16 — in the register blocking, launch
// bounds and grid shape of conv3x3_h2_f16_kernel (csrc/conv2d.hip: 4 waves per block, each 2 A fragments x 4 B fragments x 2 K sub-steps =
// 16 MFMAs per step into 8 accumulators of 16 registers, two blocks per CU)?  The round-3 ablations of that kernel stopped at 0.54-0.58 of
// the 2.5 PFLOP/s the part is quoted at with its staging, fragment reads and epilogue compiled out; this probe separates the candidates:
//   * dependency stalls      -> `chains` independent accumulators per wave (1, 2, 4, 8),
//   * waves per SIMD         -> 1, 2 or 4 blocks per CU (launch bounds + an LDS reservation that admits exactly that many),
//   * grid tail              -> `blocks`,
//   * the clock under load   -> every wave stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop, and the
//                               operands come from memory, so the caller chooses zeros / random / activation-like values (the data an MFMA
//                               multiplies sets its power, the power sets the clock: MI355X_MICROARCH.md, "DVFS give-back").
// Results: profiles/round5_*_mfma_rate_probe.*; DESIGN.md section 2.4.
#include "../p3d_common.h"
#include "p3d_probes.h"

namespace p3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int WPS>
__global__ void __launch_bounds__(256, WPS) mfma_rate_kernel(const f32x4* __restrict__ operands, float* __restrict__ sink, uint64_t* __restrict__ stamps, int iters)
{
    extern __shared__ char lds_reservation[];                       // never touched: its size caps the blocks per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = (blockIdx.x * 4 + wave) & 7;                   // eight operand sets: co-resident waves multiply different data
    f32x4 fa[2][2], fb[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[kk][i] = operands[((slot * 2 + kk) * 6 + i) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[kk][j] = operands[((slot * 2 + kk) * 6 + 2 + j) * 64 + lane];
    }
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    constexpr int dummy = 0; (void)dummy;
                    const int c = (i * 4 + j) % CHAINS;
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[kk][i]), __builtin_bit_cast(h8, fb[kk][j]), acc[c], 0, 0, 0);
                }
    }
    // the accumulators must have landed before the closing stamp: read one register of each
    float keep = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[c][r];
    asm volatile("" :: "v"(keep));
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 256 + threadIdx.x] = keep;
    if (lane == 0) {
        stamps[(blockIdx.x * 4 + wave) * 2 + 0] = t1 - t0;
        stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int CHAINS, int WPS>
static int launch_rate(const void* operands, float* sink, uint64_t* stamps, int blocks, int iters, hipStream_t s)
{
    // LDS per block that admits exactly WPS blocks of 4 waves on a 160 KB CU (WPS + 1 do not fit)
    constexpr int lds = WPS == 1 ? 96 * 1024 : WPS == 2 ? 64 * 1024 : 36 * 1024;
    static std::atomic<uint64_t> done{0};
    if (reserve_lds_once((const void*)mfma_rate_kernel<CHAINS, WPS>, lds, done) != hipSuccess) return fail(P3D_ERR_LAUNCH, "probe_mfma_rate: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((mfma_rate_kernel<CHAINS, WPS>), dim3(blocks), dim3(256), lds, s, (const f32x4*)operands, sink, stamps, iters);
    return check_launch("probe_mfma_rate");
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_probe_mfma_rate(const void* operands, float* sink, uint64_t* stamps, int32_t chains, int32_t waves_per_simd, int32_t blocks,
                                   int32_t iters, p3d_stream_t stream)
{
    P3D_REQUIRE(operands && sink && stamps && blocks >= 1 && iters >= 1, "probe_mfma_rate: null pointer or empty launch");
    hipStream_t s = (hipStream_t)stream;
#define P3D_RATE(C, W) if (chains == C && waves_per_simd == W) return launch_rate<C, W>(operands, sink, stamps, blocks, iters, s);
    P3D_RATE(1, 1) P3D_RATE(2, 1) P3D_RATE(4, 1) P3D_RATE(8, 1)
    P3D_RATE(1, 2) P3D_RATE(2, 2) P3D_RATE(4, 2) P3D_RATE(8, 2)
    P3D_RATE(1, 4) P3D_RATE(2, 4) P3D_RATE(4, 4)                      // 8 chains x 16 + 48 fragment registers do not fit 128 registers
#undef P3D_RATE
    return fail(P3D_ERR_UNSUPPORTED, "probe_mfma_rate: chains %d x waves per SIMD %d is not instantiated", chains, waves_per_simd);
}
