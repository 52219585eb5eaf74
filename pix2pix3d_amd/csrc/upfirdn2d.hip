// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, per (n, c) image.
//
// Contract: torch_utils/ops/upfirdn2d.cpp:20-102 and upfirdn2d.cu:33-204 of the reference.
// Formulation used here (derived from the op definition, upfirdn2d.py:169-213): with the virtual
// upsampled+padded signal u[X] = x[(X - pad0) / up] when that division is exact and in range,
//     y[o] = gain * sum_k u[o*down + k] * g[k],   g[k] = flip ? f[k] : f[fw-1-k]
// so only taps k == (pad0 - o*down) mod up contribute (polyphase), and every output is an
// independent gather: HBM traffic is sizeof(T) * (in + out) elements, neighbours share taps
// through L1/L2.  Lanes run along whichever axis has unit stride (W for NCHW, C for
// channels_last) so both layouts are read and written coalesced.  Compile-time (UP, DOWN, FW, FH)
// instances cover the StyleGAN2 cases ([1,3,3,1] outer product: 4x4 with up/down in {1,2});
// everything else takes the runtime-parameter instance.
#include "p3d_common.h"

namespace p3d {

struct UpfirArgs {
    const void* x; const float* f; void* y;
    int in_w, in_h, C, N;
    int64_t isx, isy, isc, isn;
    int fw, fh; int64_t fsx, fsy;
    int out_w, out_h;
    int64_t osx, osy, osc, osn;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip;
    float gain;
    int lanes_on_c;            // 1: fastest thread axis walks channels (channels_last)
    int64_t total;             // out_w * out_h * C * N
};

__device__ __forceinline__ int pos_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(UpfirArgs a)
{
    typedef typename Acc<T>::type S;
    const int upx = UPX ? UPX : a.up_x, upy = UPY ? UPY : a.up_y;
    const int dnx = DNX ? DNX : a.down_x, dny = DNY ? DNY : a.down_y;
    const int fw = FW ? FW : a.fw, fh = FH ? FH : a.fh;
    const T* xp = (const T*)a.x; T* yp = (T*)a.y;

    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ox, oy, c, n; int64_t t = idx;
        if (a.lanes_on_c) { c = (int)(t % a.C); t /= a.C; ox = (int)(t % a.out_w); t /= a.out_w; oy = (int)(t % a.out_h); n = (int)(t / a.out_h); }
        else              { ox = (int)(t % a.out_w); t /= a.out_w; oy = (int)(t % a.out_h); t /= a.out_h; c = (int)(t % a.C); n = (int)(t / a.C); }

        const int bx = ox * dnx - a.pad_x0;          // upsampled-grid coordinate of tap 0
        const int by = oy * dny - a.pad_y0;
        const int kx0 = pos_mod(-bx, upx), ky0 = pos_mod(-by, upy);
        const T* img = xp + c * a.isc + n * a.isn;
        S acc = 0;
        constexpr bool kStatic = (UPX && UPY && FW && FH);
        constexpr int NJX = kStatic ? (FW + (UPX ? UPX : 1) - 1) / (UPX ? UPX : 1) : 0;
        constexpr int NJY = kStatic ? (FH + (UPY ? UPY : 1) - 1) / (UPY ? UPY : 1) : 0;
        auto tap = [&](int kx, int ky) {
            const int ix = (bx + kx) / upx, iy = (by + ky) / upy;      // exact divisions by construction
            if (kx < fw && ky < fh && ix >= 0 && ix < a.in_w && iy >= 0 && iy < a.in_h) {
                const int fx = a.flip ? kx : fw - 1 - kx, fy = a.flip ? ky : fh - 1 - ky;
                acc += ld(img + ix * a.isx + iy * a.isy) * (S)a.f[fx * a.fsx + fy * a.fsy];
            }
        };
        if constexpr (kStatic) {
#pragma unroll
            for (int jy = 0; jy < NJY; ++jy)
#pragma unroll
                for (int jx = 0; jx < NJX; ++jx) tap(kx0 + jx * upx, ky0 + jy * upy);
        } else {
            for (int ky = ky0; ky < fh; ky += upy)
                for (int kx = kx0; kx < fw; kx += upx) tap(kx, ky);
        }
        st(yp + ox * a.osx + oy * a.osy + c * a.osc + n * a.osn, acc * (S)a.gain);
    }
}

template <class T>
static int launch_upfirdn2d(const UpfirArgs& a, hipStream_t s)
{
    const int threads = 256;
    int64_t blocks64 = (a.total + threads - 1) / threads;
    const int64_t cap = (int64_t)kNumCU * 32;
    int blocks = (int)(blocks64 > cap ? cap : blocks64);
    if (blocks < 1) blocks = 1;
#define P3D_UPFIR_CASE(ux, uy, dx, dy, w, h) \
    if (a.up_x == ux && a.up_y == uy && a.down_x == dx && a.down_y == dy && a.fw == w && a.fh == h) { \
        hipLaunchKernelGGL((upfirdn2d_kernel<T, ux, uy, dx, dy, w, h>), dim3(blocks), dim3(threads), 0, s, a); } else
    P3D_UPFIR_CASE(1, 1, 1, 1, 4, 4)
    P3D_UPFIR_CASE(2, 2, 1, 1, 4, 4)
    P3D_UPFIR_CASE(1, 1, 2, 2, 4, 4)
    P3D_UPFIR_CASE(2, 1, 1, 1, 4, 1)
    P3D_UPFIR_CASE(1, 2, 1, 1, 1, 4)
    P3D_UPFIR_CASE(1, 1, 2, 1, 4, 1)
    P3D_UPFIR_CASE(1, 1, 1, 2, 1, 4)
    { hipLaunchKernelGGL((upfirdn2d_kernel<T, 0, 0, 0, 0, 0, 0>), dim3(blocks), dim3(threads), 0, s, a); }
#undef P3D_UPFIR_CASE
    count_launch(FAM_UPFIRDN);
    return check_launch("upfirdn2d");
}

} // namespace p3d

extern "C" int p3d_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                             const int32_t in_size[4], const int64_t in_stride[4],
                             const int32_t f_size[2], const int64_t f_stride[2],
                             const int32_t out_size[4], const int64_t out_stride[4],
                             int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                             int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && f && y, "upfirdn2d: x, f and y must be non-null");
    P3D_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down factors must be >= 1");
    P3D_REQUIRE(f_size[0] >= 1 && f_size[1] >= 1, "upfirdn2d: filter must have at least one tap");
    P3D_REQUIRE(out_size[2] == in_size[2] && out_size[3] == in_size[3], "upfirdn2d: channel/batch mismatch between x and y");
    UpfirArgs a;
    a.x = x; a.f = f; a.y = y;
    a.in_w = in_size[0]; a.in_h = in_size[1]; a.C = in_size[2]; a.N = in_size[3];
    a.isx = in_stride[0]; a.isy = in_stride[1]; a.isc = in_stride[2]; a.isn = in_stride[3];
    a.fw = f_size[0]; a.fh = f_size[1]; a.fsx = f_stride[0]; a.fsy = f_stride[1];
    a.out_w = out_size[0]; a.out_h = out_size[1];
    a.osx = out_stride[0]; a.osy = out_stride[1]; a.osc = out_stride[2]; a.osn = out_stride[3];
    a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
    a.pad_x0 = pad_x0; a.pad_y0 = pad_y0; a.flip = flip ? 1 : 0; a.gain = gain;
    a.lanes_on_c = (a.osc == 1 && a.C > 1 && a.osx != 1) ? 1 : 0;
    a.total = (int64_t)a.out_w * a.out_h * a.C * a.N;
    if (a.total <= 0) return P3D_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case P3D_F32: return launch_upfirdn2d<float>(a, s);
        case P3D_F16: return launch_upfirdn2d<__half>(a, s);
        case P3D_F64: return launch_upfirdn2d<double>(a, s);
    }
    return fail(P3D_ERR_ARGUMENT, "upfirdn2d: unknown dtype %d", dtype);
}
