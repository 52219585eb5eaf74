// filtered_lrelu: bias -> x`up` zero-insertion + FIR -> leaky ReLU with gain and clamp -> FIR + x`down` decimation, in one pass, with the
// bit-packed sign tensor that lets the backward pass be the same kernel (torch_utils/ops/filtered_lrelu.py:180-270).
//
// Stands in for filtered_lrelu_plugin (torch_utils/ops/filtered_lrelu.cpp:20-213 + the 1 100-line template family of
// filtered_lrelu.cu:143-1103, one instantiation per (up, down, filter size, separable?) with its own tile sizes and constant-memory
// filter tables) and for the in-place activation helper filtered_lrelu_act_ (filtered_lrelu.cpp:217-296, filtered_lrelu.cu:1109-1213).
// Here: ONE kernel, run-time geometry.  A work-group owns a 16 x 16 output tile of one (image, channel) plane:
//   1. the input footprint (+ bias, zero outside the image) goes to LDS as fp32;
//   2. every element of the up-sampled footprint ((16-1)*down + fd taps per axis) is the polyphase sum over the input taps that are
//      not zero-stuffing, times up^2 * gain; then sign / slope / clamp — or, in the backward configuration, the saved sign code
//      instead of the comparison — and lands in a second LDS tile;
//   3. the outputs are the down-sampling FIR over that tile.
// Filters are passed as dense 2-D fp32 tables (the host expands separable ones) and live in LDS: no device globals, so concurrent
// streams are safe (the reference's __constant__ tables are not, filtered_lrelu.cu:81-82).
//
// Sign tensor (its byte layout is a contract: it is saved between forward and backward, filtered_lrelu.cpp:93-97,
// filtered_lrelu.cu:480-523): uint8 [N][C][sh][sw/4] contiguous, sh = yh*down - (down-1) + (fd_h-1), sw = the same along x rounded up
// to 16 elements.  Element (x, y) of the up-sampled grid sits in byte ((x + ofs_x) >> 2) + (sw/4) * ((y + ofs_y) + sh * (n*C + c)) at
// bit ((x + ofs_x) & 3) * 2; code 0 = passed, 1 = negative (times slope on the way back), 2 = clamped (gradient zero; replaces the
// sign code).  Written a whole byte at a time by the lane that owns its four elements (the reference ORs 2-bit fields across four
// lanes with shuffles; tiles start on multiples of 16 * down columns, so bytes never straddle two writers' halves unevenly).
#include "p3d_common.h"

namespace p3d {

struct FlreluArgs {
    const void* x; const void* b; void* y; uint8_t* s;
    const float* fu; const float* fd;
    int xw, xh, C, N;  int64_t xs_w, xs_h, xs_c, xs_n;     // element strides
    int yw, yh;        int64_t ys_w, ys_h, ys_c, ys_n;
    int64_t bs;
    int fuw, fuh, fdw, fdh, up, down, px0, py0;
    int s_wb, s_h, sofs_x, sofs_y, sw_limit;                // sign tensor: bytes per row, rows; offsets; valid bytes per row
    float gain, slope, clamp;
    int flip, mode;                                         // mode 0: plain, 1: write signs, 2: read signs
    int tiles_x, tiles_y;
    int IW, IH, UW, UH;                                     // LDS tile sizes
};

constexpr int FL_TILE = 16;

__device__ __forceinline__ int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

template <class T>
__global__ void __launch_bounds__(256) filtered_lrelu_kernel(FlreluArgs a)
{
    extern __shared__ float fl_lds[];
    float* const s_fu = fl_lds;                                   // [fuh][fuw], already in correlation order
    float* const s_fd = s_fu + a.fuh * a.fuw;                     // [fdh][fdw]
    float* const s_in = s_fd + a.fdh * a.fdw;                     // [IH][IW]
    float* const s_up = s_in + a.IH * a.IW;                       // [UH][UW]
    const int tid = threadIdx.x;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int plane = blockIdx.x / ntiles;                        // n * C + c
    const int n = plane / a.C, c = plane - n * a.C;
    const int tile = blockIdx.x - plane * ntiles, ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * FL_TILE, ox0 = tx * FL_TILE;
    const int uy0 = oy0 * a.down, ux0 = ox0 * a.down;            // first up-grid element of this tile
    // first input element any tap of the footprint can touch: padded index j = u + k maps to input (j - pad0) / up
    const int iy0 = floordiv(uy0 - a.py0 + a.up - 1, a.up), ix0 = floordiv(ux0 - a.px0 + a.up - 1, a.up);

    // filters: upfirdn2d correlates with the MIRRORED filter unless flip_filter is set (upfirdn2d.py:199-200)
    for (int e = tid; e < a.fuh * a.fuw; e += 256) s_fu[e] = a.flip ? a.fu[e] : a.fu[a.fuh * a.fuw - 1 - e];
    for (int e = tid; e < a.fdh * a.fdw; e += 256) s_fd[e] = a.flip ? a.fd[e] : a.fd[a.fdh * a.fdw - 1 - e];
    // input footprint + bias (zero outside the image: the reference zero-pads AFTER the bias add only implicitly — taps outside read 0)
    const float bias = a.b ? (float)ld((const T*)a.b + c * a.bs) : 0.f;
    const T* xp = (const T*)a.x + n * a.xs_n + c * a.xs_c;
    for (int e = tid; e < a.IH * a.IW; e += 256) {
        const int ry = e / a.IW, rx = e - ry * a.IW;
        const int iy = iy0 + ry, ix = ix0 + rx;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)a.xh && (unsigned)ix < (unsigned)a.xw) v = (float)ld(xp + iy * a.xs_h + ix * a.xs_w) + bias;
        s_in[e] = v;
    }
    __syncthreads();

    // up-sampled footprint, four consecutive x per thread (= one sign byte)
    const int uw4 = (a.UW + 3) >> 2;
    const float scale = (float)a.up * (float)a.up * a.gain;
    for (int e = tid; e < a.UH * uw4; e += 256) {
        const int ry = e / uw4, q = e - ry * uw4;
        const int uy = uy0 + ry;
        // rows: taps ky = ky0 + s * up are the ones that land on real input rows
        int ky0 = (a.py0 - uy) % a.up; if (ky0 < 0) ky0 += a.up;
        const int iyr = (uy + ky0 - a.py0) / a.up - iy0;          // LDS row of the first such tap (exact division by construction)
        unsigned codes = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rx = q * 4 + j;                            // (the last byte of a row may run up to 3 elements past UW: computed for the
            const int ux = ux0 + rx;                             //  sign tensor — whole bytes only — but not kept in the tile)
            int kx0 = (a.px0 - ux) % a.up; if (kx0 < 0) kx0 += a.up;
            const int ixr = (ux + kx0 - a.px0) / a.up - ix0;
            float acc = 0.f;
            for (int ky = ky0, sy = 0; ky < a.fuh; ky += a.up, ++sy)
                for (int kx = kx0, sx = 0; kx < a.fuw; kx += a.up, ++sx)
                    acc = fmaf(s_in[(iyr + sy) * a.IW + ixr + sx], s_fu[ky * a.fuw + kx], acc);
            float v = acc * scale;
            unsigned code = 0;
            if (a.mode == 1) {                                   // forward, signs kept (filtered_lrelu.cu:497-508)
                code = __float_as_uint(v) >> 31;
                if (code) v *= a.slope;
                if (fabsf(v) > a.clamp) { code = 2; v = fminf(fmaxf(v, -a.clamp), a.clamp); }
            } else if (a.mode == 2) {                            // backward configuration: apply the saved code (:566-576)
                const int sx = ux + a.sofs_x, sy = uy + a.sofs_y;
                if ((unsigned)(sx >> 2) < (unsigned)a.sw_limit && sx >= 0 && (unsigned)sy < (unsigned)a.s_h) {
                    const unsigned byte = a.s[(sx >> 2) + (int64_t)a.s_wb * (sy + (int64_t)a.s_h * plane)];
                    const unsigned cd = (byte >> ((sx & 3) << 1)) & 3u;
                    if (cd & 1u) v *= a.slope;
                    if (cd & 2u) v = 0.f;
                }
            } else {
                if (v < 0.f) v *= a.slope;
                v = fminf(fmaxf(v, -a.clamp), a.clamp);
            }
            codes |= code << (j * 2);
            if (rx < a.UW) s_up[ry * a.UW + rx] = v;
        }
        if (a.mode == 1) {
            // neighbouring tiles overlap by fd - down columns / rows and write the same bytes with the same values
            const int sxb = (ux0 + q * 4 + a.sofs_x) >> 2, sy = uy + a.sofs_y;   // sofs_x is a multiple of 4 in this mode (host-checked)
            if ((unsigned)sxb < (unsigned)a.sw_limit && (unsigned)sy < (unsigned)a.s_h)
                a.s[sxb + (int64_t)a.s_wb * (sy + (int64_t)a.s_h * plane)] = (uint8_t)codes;
        }
    }
    __syncthreads();

    // outputs: down-sampling FIR over the activated footprint
    const int oy = oy0 + (tid >> 4), ox = ox0 + (tid & 15);
    if (oy < a.yh && ox < a.yw) {
        const float* up0 = s_up + (tid >> 4) * a.down * a.UW + (tid & 15) * a.down;
        float acc = 0.f;
        for (int ky = 0; ky < a.fdh; ++ky)
            for (int kx = 0; kx < a.fdw; ++kx)
                acc = fmaf(up0[ky * a.UW + kx], s_fd[ky * a.fdw + kx], acc);
        st((T*)a.y + n * a.ys_n + c * a.ys_c + oy * a.ys_h + ox * a.ys_w, acc);
    }
}

// ---- activation only, in place (the generic route of filtered_lrelu.py:225-231) ---------------------------------------------------------
struct FlreluActArgs {
    void* x; uint8_t* s;
    int xw, xh, C, N; int64_t xs_w, xs_h, xs_c, xs_n;
    int s_w, s_h, sofs_x, sofs_y;                            // sign tensor width in ELEMENTS (a multiple of 16), rows
    float gain, slope, clamp; int mode;
};

template <class T>
__global__ void __launch_bounds__(256) filtered_lrelu_act_kernel(FlreluActArgs a)
{
    typedef typename Acc<T>::type acc_t;
    // one thread = four consecutive x of one row = one sign byte; write mode covers the sign tensor's extent, the others x's
    const int w4 = ((a.mode == 1 ? a.s_w : a.xw) + 3) >> 2;
    const int rows = a.mode == 1 ? a.s_h : a.xh;
    const int64_t total = (int64_t)w4 * rows * a.C * a.N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % w4);
        const int y = (int)((e / w4) % rows);
        const int plane = (int)(e / ((int64_t)w4 * rows));
        const int n = plane / a.C, c = plane - n * a.C;
        unsigned codes = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = q * 4 + j;
            if (x >= a.xw || y >= a.xh) continue;
            T* pv = (T*)a.x + n * a.xs_n + c * a.xs_c + y * a.xs_h + x * a.xs_w;
            acc_t v = ld(pv) * (acc_t)a.gain;
            if (a.mode == 1) {                                                  // (:1140-1149: this kernel tests v < 0, not the sign bit)
                unsigned code = 0;
                if (v < 0) { v *= (acc_t)a.slope; code = 1; }
                if (fabs((double)v) > (double)a.clamp) { v = v < 0 ? -(acc_t)a.clamp : (acc_t)a.clamp; code = 2; }
                codes |= code << (j * 2);
            } else if (a.mode == 2) {
                const unsigned sx = (unsigned)(x + a.sofs_x), sy = (unsigned)(y + a.sofs_y);
                if (sx < (unsigned)a.s_w && sy < (unsigned)a.s_h) {
                    const unsigned byte = a.s[(sx >> 2) + (int64_t)(a.s_w >> 2) * (sy + (int64_t)a.s_h * plane)];
                    const unsigned cd = (byte >> ((sx & 3) << 1)) & 3u;
                    if (cd & 1u) v *= (acc_t)a.slope;
                    if (cd & 2u) v = 0;
                }
            } else {
                if (v < 0) v *= (acc_t)a.slope;
                if (fabs((double)v) > (double)a.clamp) v = v < 0 ? -(acc_t)a.clamp : (acc_t)a.clamp;
            }
            st(pv, v);
        }
        if (a.mode == 1) a.s[q + (int64_t)(a.s_w >> 2) * (y + (int64_t)a.s_h * plane)] = (uint8_t)codes;
    }
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, uint8_t* s, void* y, int dtype,
                                  const int32_t x_size[4], const int64_t x_stride[4], const int32_t y_size[4], const int64_t y_stride[4], int64_t b_stride,
                                  int32_t fu_w, int32_t fu_h, int32_t fd_w, int32_t fd_h, int32_t up, int32_t down, int32_t pad_x0, int32_t pad_y0,
                                  int32_t s_width_bytes, int32_t s_height, int32_t s_ofs_x, int32_t s_ofs_y, int32_t sw_limit,
                                  float gain, float slope, float clamp, int32_t flip_filters, int32_t sign_mode, p3d_stream_t stream)
{
    P3D_REQUIRE(x && fu && fd && y && x_size && x_stride && y_size && y_stride, "filtered_lrelu: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32, "filtered_lrelu: x must be float16 or float32");
    P3D_REQUIRE(up >= 1 && down >= 1 && fu_w >= 1 && fu_h >= 1 && fd_w >= 1 && fd_h >= 1, "filtered_lrelu: up, down and the filter sizes must be at least 1");
    P3D_REQUIRE(sign_mode >= 0 && sign_mode <= 2 && (sign_mode == 0 || s), "filtered_lrelu: sign_mode 1 / 2 needs the sign tensor");
    for (int i = 0; i < 4; ++i) P3D_REQUIRE(x_size[i] >= 1 && y_size[i] >= 1, "filtered_lrelu: empty tensor");
    P3D_REQUIRE(x_size[2] == y_size[2] && x_size[3] == y_size[3], "filtered_lrelu: x and y must agree in batch and channels");
    FlreluArgs a{};
    a.x = x; a.b = b; a.y = y; a.s = s; a.fu = fu; a.fd = fd;
    a.xw = x_size[0]; a.xh = x_size[1]; a.C = x_size[2]; a.N = x_size[3];
    a.xs_w = x_stride[0]; a.xs_h = x_stride[1]; a.xs_c = x_stride[2]; a.xs_n = x_stride[3];
    a.yw = y_size[0]; a.yh = y_size[1];
    a.ys_w = y_stride[0]; a.ys_h = y_stride[1]; a.ys_c = y_stride[2]; a.ys_n = y_stride[3];
    a.bs = b_stride;
    a.fuw = fu_w; a.fuh = fu_h; a.fdw = fd_w; a.fdh = fd_h; a.up = up; a.down = down; a.px0 = pad_x0; a.py0 = pad_y0;
    a.s_wb = s_width_bytes; a.s_h = s_height; a.sofs_x = s_ofs_x; a.sofs_y = s_ofs_y; a.sw_limit = sw_limit;
    a.gain = gain; a.slope = slope; a.clamp = clamp; a.flip = flip_filters; a.mode = sign_mode;
    a.tiles_x = ceil_div(a.yw, FL_TILE); a.tiles_y = ceil_div(a.yh, FL_TILE);
    a.UW = (FL_TILE - 1) * down + fd_w; a.UH = (FL_TILE - 1) * down + fd_h;
    a.IW = (((a.UW + 3) & ~3) + fu_w - 2) / up + 2; a.IH = (a.UH + fu_h - 2) / up + 2;      // (width: whole sign bytes, see the kernel)
    const size_t lds = (size_t)(fu_w * fu_h + fd_w * fd_h + a.IW * a.IH + a.UW * a.UH) * sizeof(float);
    // "no specialised kernel": same meaning as the plugin's return code -1 (filtered_lrelu.cpp:56-60) — the caller takes the generic route
    // (gfx950 has 160 KB of LDS per CU; more than 64 KB per block is an opt-in per kernel and device, taken once below)
    constexpr size_t kLdsMax = 160 * 1024;
    if (lds > kLdsMax) return fail(P3D_ERR_UNSUPPORTED, "filtered_lrelu: tiles of %zu B do not fit %zu B of LDS (filters %dx%d / %dx%d, up %d, down %d)", lds, kLdsMax, fu_w, fu_h, fd_w, fd_h, up, down);
    if (sign_mode == 1 && (s_ofs_x & 3)) return fail(P3D_ERR_UNSUPPORTED, "filtered_lrelu: sign offset x = %d is not byte aligned in write mode", s_ofs_x);
    const int64_t blocks = (int64_t)a.C * a.N * a.tiles_x * a.tiles_y;
    P3D_REQUIRE(blocks < (1ll << 31), "filtered_lrelu: too many tiles");
    hipStream_t st_ = (hipStream_t)stream;
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> once_h{0}, once_f{0};
        const hipError_t e = dtype == P3D_F16 ? reserve_lds_once((const void*)filtered_lrelu_kernel<__half>, (int)kLdsMax, once_h)
                                              : reserve_lds_once((const void*)filtered_lrelu_kernel<float>, (int)kLdsMax, once_f);
        if (e != hipSuccess) return fail(P3D_ERR_LAUNCH, "filtered_lrelu: cannot reserve %zu B of LDS: %s", kLdsMax, hipGetErrorString(e));
    }
    if (dtype == P3D_F16) hipLaunchKernelGGL(filtered_lrelu_kernel<__half>, dim3((unsigned)blocks), dim3(256), lds, st_, a);
    else                  hipLaunchKernelGGL(filtered_lrelu_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st_, a);
    count_launch(FAM_FLRELU);
    return check_launch("filtered_lrelu");
}

extern "C" int p3d_filtered_lrelu_act(void* x, uint8_t* s, int dtype, const int32_t x_size[4], const int64_t x_stride[4],
                                      int32_t s_width, int32_t s_height, int32_t s_ofs_x, int32_t s_ofs_y, float gain, float slope, float clamp,
                                      int32_t sign_mode, p3d_stream_t stream)
{
    P3D_REQUIRE(x && x_size && x_stride, "filtered_lrelu_act: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32 || dtype == P3D_F64, "filtered_lrelu_act: x must be float16, float32 or float64");
    P3D_REQUIRE(sign_mode >= 0 && sign_mode <= 2 && (sign_mode == 0 || s), "filtered_lrelu_act: sign_mode 1 / 2 needs the sign tensor");
    P3D_REQUIRE(sign_mode == 0 || (s_width % 16 == 0 && s_height >= 1), "filtered_lrelu_act: sign width must be a multiple of 16 elements");
    FlreluActArgs a{};
    a.x = x; a.s = s;
    a.xw = x_size[0]; a.xh = x_size[1]; a.C = x_size[2]; a.N = x_size[3];
    a.xs_w = x_stride[0]; a.xs_h = x_stride[1]; a.xs_c = x_stride[2]; a.xs_n = x_stride[3];
    a.s_w = s_width; a.s_h = s_height; a.sofs_x = s_ofs_x; a.sofs_y = s_ofs_y;
    a.gain = gain; a.slope = slope; a.clamp = clamp; a.mode = sign_mode;
    const int w4 = ((sign_mode == 1 ? s_width : a.xw) + 3) >> 2;
    const int64_t total = (int64_t)w4 * (sign_mode == 1 ? s_height : a.xh) * a.C * a.N;
    if (total <= 0) return P3D_OK;
    const int blocks = (int)((total + 255) / 256 < 16 * kNumCU ? (total + 255) / 256 : 16 * kNumCU);
    hipStream_t st_ = (hipStream_t)stream;
    if (dtype == P3D_F16)      hipLaunchKernelGGL(filtered_lrelu_act_kernel<__half>, dim3(blocks), dim3(256), 0, st_, a);
    else if (dtype == P3D_F32) hipLaunchKernelGGL(filtered_lrelu_act_kernel<float>, dim3(blocks), dim3(256), 0, st_, a);
    else                       hipLaunchKernelGGL(filtered_lrelu_act_kernel<double>, dim3(blocks), dim3(256), 0, st_, a);
    count_launch(FAM_FLRELU);
    return check_launch("filtered_lrelu_act");
}
