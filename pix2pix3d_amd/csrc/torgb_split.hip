// Wide ToRGB of the backbone on split activations, with the skip-image sum, in one pass (bf16x3), gfx950.
//     y = clamp(sum_c x[c] * wm[o][c] + bias[o])            (ToRGBLayer, training/networks_stylegan2.py:355-359: 1x1 modulated conv, no demodulation)
//     img = upsample2d(img_prev, f) + y                      (SynthesisBlock, :453-459)
// The block's 96 tri-plane channels leave a 128- or 256-channel layer at 128^2 / 256^2: a GEMM with K = Ci whose cost is reading x once and
// writing the image once.  On the generic convolution kernel it was a 4..8-step K loop per 128 x 128 tile (106 us at 128 -> 96 @256^2 x 4,
// 2.2 TB/s), and the upsampling launch then read that image back to add its predecessor (59 us more).  Here:
//   * x arrives as split K rows ([32 x bf16 hi | 32 x bf16 lo] per 32 channels, csrc/conv2d.hip XS) and goes from global memory straight
//     into MFMA A fragments: lane (pixel = lane & 31, k-group = lane >> 5) loads the 16-byte hi and lo pieces of its 8 channels per k-step —
//     every byte of a pixel's row is read exactly once, by two lanes;
//   * the image's modulated weights (p3d_modulate_weights, P3D_F32_BF16X3, [N][Co][1][Ci]) sit in LDS in fragment order (one conflict-free
//     ds_read_b128 per B fragment);
//   * a wave owns 32 pixels x all Co <= 96 channels (three 32x32 accumulators), the NEXT tile's A fragments load while this one multiplies;
//   * the epilogue adds bias, clamps, adds the x2-upsampled predecessor image — the 2 x 2 taps of upfirdn2d(up = 2, 4x4 filter, pad 2,
//     gain 4) summed in that kernel's order, so the result equals "ToRGB, then p3d_upfirdn2d_acc" to the last bit of the skip term — and
//     stores 128 bytes per half-wave (a pixel's 32 consecutive channels).  The predecessor's taps are requested a group of four pixels ahead
//     (24 values x 3 channel blocks in registers): a first version that loaded them where it used them ran one memory round trip per VALUE
//     (309 us on the 256^2 layer; the loads may alias the stores, so the compiler kept them in program order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "p3d_common.h"
#include "../../include/p3d_hip.h"

namespace p3d {

typedef float tf4 __attribute__((ext_vector_type(4)));
typedef float tf16 __attribute__((ext_vector_type(16)));
typedef __bf16 tb8 __attribute__((ext_vector_type(8)));

struct TorgbSplitArgs {
    const float* x;        // [N][H*W][Ci] split rows
    const float* wm;       // [N][Co][Ci]  split rows (weight * styles)
    const float* bias;     // [Co] or null
    float* y;              // [N][H][W][Co] fp32
    const float* prev;     // [N][H/2][W/2][Co] fp32 or null
    float fr[4][4];        // the upsampling filter as upfirdn2d_cl_kernel builds it: f[3 - kx][3 - ky-th row] * gain
    int H, W, Co;
    float clamp;           // < 0: off
};

template <int KS, int NT>  // 16-channel k-steps: Ci = 16 * KS; threads per block (256 with 48 KB of weights: two blocks per CU; 512 with 96 KB: one)
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1) torgb_wide_split_kernel(TorgbSplitArgs a)
{
    constexpr int Ci = KS * 16, NKC = KS / 8;                                   // A fragments travel in chunks of 8 k-steps (64 registers), two chunks in flight
    extern __shared__ __attribute__((aligned(16))) char tw_lds[];
    tf4* const wl = (tf4*)tw_lds;                                               // [(j * KS + ks) * 2 + hl][64 lanes]
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kg = lane >> 5;
    const int nj = a.Co / 32;
    {
        const char* wn = (const char*)(a.wm + (int64_t)n * a.Co * Ci);
#pragma unroll 6                                                                // (rolled, each of the twelve iterations was a memory round trip of its own: a fifth of the 128^2 launch)
        for (int e = tid; e < nj * KS * 2 * 64; e += NT) {
            const int ln = e & 63, frag = e >> 6, hl = frag & 1, ks = (frag >> 1) % KS, j = frag / (2 * KS);
            const int co = j * 32 + (ln & 31);
            wl[e] = *(const tf4*)(wn + (int64_t)co * Ci * 4 + (ks >> 1) * 128 + (ks & 1) * 32 + (ln >> 5) * 16 + hl * 64);
        }
    }
    __syncthreads();
    const int HW = a.H * a.W, ntiles = HW / 32;                                 // (host: W % 32 == 0: a tile is 32 consecutive pixels of ONE image row)
    const char* const xn = (const char*)(a.x + (int64_t)n * HW * Ci);
    float* __restrict__ const yn = a.y + (int64_t)n * HW * a.Co;
    const int PH = a.H / 2, PW = a.W / 2;
    const float* __restrict__ const pn = a.prev ? a.prev + (int64_t)n * PH * PW * a.Co : nullptr;
    float bias[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const bool bok = a.bias && j < nj; const float v = (bok ? a.bias : a.x)[bok ? j * 32 + col : 0]; bias[j] = bok ? v : 0.f; }      // (one batch, no branch)

    auto load_chunk = [&](tf4 (&ah)[8], tf4 (&al)[8], int tile, int kc) {
        const char* px = xn + ((int64_t)tile * 32 + col) * Ci * 4 + kg * 16 + kc * 512;       // 8 k-steps = four 128-byte rows
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ah[q] = *(const tf4*)(px + (q >> 1) * 128 + (q & 1) * 32);
            al[q] = *(const tf4*)(px + (q >> 1) * 128 + (q & 1) * 32 + 64);
        }
    };
    tf16 acc[3];
    auto clear_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    auto mma_chunk = [&](const tf4 (&ah)[8], const tf4 (&al)[8], int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nj) {
                    const int ks = kc * 8 + q;
                    const tb8 bh = __builtin_bit_cast(tb8, wl[((j * KS + ks) * 2 + 0) * 64 + lane]);
                    const tb8 bl = __builtin_bit_cast(tb8, wl[((j * KS + ks) * 2 + 1) * 64 + lane]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tb8, ah[q]), bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tb8, ah[q]), bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tb8, al[q]), bh, acc[j], 0, 0, 0);
                }
            }
    };
    auto store_tile = [&](int tile) {
        // accumulator element r = 4 g + e of lane (col, kg): pixel m = 8 g + 4 kg + e of the tile, channel 32 j + col.
        // Skip term = upfirdn2d_cl_kernel<float, 2, 1, 4> (pad 2) at that pixel: taps (jy, jx) in {0, 1}^2 of prev rows iyb + jy, columns ixb + jx with weights
        // fr[2 jy + ky0][2 jx + kx0], ky0 = (oy - 2) & 1, kx0 = (ox - 2) & 1, iyb = (oy - 2 + ky0) >> 1, ixb likewise.  The tile starts at a multiple of 32, so for
        // E = ox0 + 8 g + 4 kg: kx0 = e & 1 and ixb = E / 2 - 1 + {0, 1, 1, 2}[e]: the four pixels of a group read FOUR prev columns E/2 - 1 .. E/2 + 2 of two rows.
        // Those 24 values (x 3 channel blocks) are requested one group AHEAD and consumed from registers: one memory round trip per tile is exposed, not 48.
        const int p0 = tile * 32, oy = p0 / a.W, ox0 = p0 - oy * a.W;
        const int by = oy - 2, ky0 = by & 1, iyb = (by + ky0) >> 1;
        float fw[2][4];                                                         // the two filter rows this output row uses (uniform over the tile)
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) fw[jy][kx] = ky0 ? a.fr[2 * jy + 1][kx] : a.fr[2 * jy][kx];
        auto load_group = [&](float (&pv)[2][4][3], int g) {
            const int base = (ox0 + 8 * g + 4 * kg) >> 1;
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int iy = iyb + jy, ix = base - 1 + c;
                    const bool ok = (iy >= 0) & (iy < PH) & (ix >= 0) & (ix < PW);
                    const float* src = pn + ((int64_t)(ok ? iy : 0) * PW + (ok ? ix : 0)) * a.Co + col;
#pragma unroll
                    for (int j = 0; j < 3; ++j) pv[jy][c][j] = (ok && j < nj) ? src[j * 32] : 0.f;      // (an out-of-image tap contributes fma(0, w, up) == up)
                }
        };
        auto emit_group = [&](const float (&pv)[2][4][3], int g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 8 * g + 4 * kg + e, c0 = (e + (e & 1)) >> 1;
                float* const dst = yn + ((int64_t)p0 + m) * a.Co + col;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < nj) {
                        float v = acc[j][4 * g + e] + bias[j];
                        if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                        if (pn) {
                            float up = 0.f;
#pragma unroll
                            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                                for (int jx = 0; jx < 2; ++jx) up = fmaf(pv[jy][c0 + jx][j], fw[jy][2 * jx + (e & 1)], up);
                            v = up + v;
                        }
                        dst[j * 32] = v;
                    }
                }
            }
        };
        float pa[2][4][3], pb[2][4][3];
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j) { pa[jy][c][j] = 0.f; pb[jy][c][j] = 0.f; }
        if (pn) load_group(pa, 0);
        if (pn) load_group(pb, 1);
        emit_group(pa, 0);
        if (pn) load_group(pa, 2);
        emit_group(pb, 1);
        if (pn) load_group(pb, 3);
        emit_group(pa, 2);
        emit_group(pb, 3);
    };

    const int nwaves = gridDim.x * (NT / 64);
    int tile = blockIdx.x * (NT / 64) + wave;
    tf4 h0[8], l0[8], h1[8], l1[8];
    if (tile < ntiles) load_chunk(h0, l0, tile, 0);
    if constexpr (NKC == 1) {                                                   // one chunk per tile: the two buffers alternate between tiles
        while (tile < ntiles) {
            if (tile + nwaves < ntiles) load_chunk(h1, l1, tile + nwaves, 0);
            clear_acc(); mma_chunk(h0, l0, 0); store_tile(tile);
            tile += nwaves;
            if (tile >= ntiles) break;
            if (tile + nwaves < ntiles) load_chunk(h0, l0, tile + nwaves, 0);
            clear_acc(); mma_chunk(h1, l1, 0); store_tile(tile);
            tile += nwaves;
        }
    } else {                                                                    // two chunks per tile: chunk 1 loads under chunk 0, the next tile's chunk 0 under chunk 1
        while (tile < ntiles) {
            load_chunk(h1, l1, tile, 1);
            clear_acc(); mma_chunk(h0, l0, 0);
            if (tile + nwaves < ntiles) load_chunk(h0, l0, tile + nwaves, 0);
            mma_chunk(h1, l1, 1); store_tile(tile);
            tile += nwaves;
        }
    }
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_torgb_wide_split(const void* x_split, const void* wmod_split, const float* bias, float* y_nhwc, const float* prev_nhwc, const float* f4x4_host,
                                    int32_t n_img, int32_t h, int32_t w, int32_t ci, int32_t co, float clamp, p3d_stream_t stream)
{
    P3D_REQUIRE(x_split && wmod_split && y_nhwc, "torgb_wide_split: null pointer");
    P3D_REQUIRE(!prev_nhwc || f4x4_host, "torgb_wide_split: the skip image needs its upsampling filter");
    P3D_REQUIRE(n_img >= 1 && n_img < 65536 && h >= 1 && w >= 1, "torgb_wide_split: bad sizes");
    if (!(ci == 128 || ci == 256) || co % 32 != 0 || co < 32 || co > 96 || w % 32 != 0 || (prev_nhwc && ((h | w) & 1)))
        return fail(P3D_ERR_UNSUPPORTED, "torgb_wide_split: needs Ci in {128, 256}, Co in {32, 64, 96}, W %% 32 == 0 (got %d, %d, %d x %d)", ci, co, h, w);
    P3D_REQUIRE(((((uintptr_t)x_split) | ((uintptr_t)wmod_split) | ((uintptr_t)y_nhwc)) & 15u) == 0, "torgb_wide_split: pointers must be 16-byte aligned");
    TorgbSplitArgs a{};
    a.x = (const float*)x_split; a.wm = (const float*)wmod_split; a.bias = bias; a.y = y_nhwc; a.prev = prev_nhwc; a.H = h; a.W = w; a.Co = co; a.clamp = clamp;
    if (prev_nhwc) {
        for (int ky = 0; ky < 4; ++ky)                                            // (f4x4_host: sixteen floats in HOST memory, row-major)
            for (int kx = 0; kx < 4; ++kx) a.fr[ky][kx] = f4x4_host[(3 - kx) + (3 - ky) * 4] * 4.f;      // upfirdn2d_cl_kernel: fr[ky][kx] = f[fx * fsx + fy * fsy] * gain, fx = 3 - kx, fy = 3 - ky
    }
    const int lds = (co / 32) * (ci / 16) * 2 * 64 * 16;                           // 48 KB (Ci = 128) / 96 KB (Ci = 256) at Co = 96
    const int ntiles = h * w / 32;
    const int wpb = ci == 128 ? 4 : 8;                                             // waves per block
    int blocks = (ntiles + wpb - 1) / wpb;
    const int cap = (kNumCU * (ci == 128 ? 2 : 1) + n_img - 1) / n_img;            // eight waves per CU either way: two per SIMD, each with two fragment chunks in flight
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    static std::atomic<uint64_t> done8{0}, done16{0};
    // the reservation is made once per device and must cover every later call: the kernel's maximum (Co = 96), not this call's Co
    constexpr int kLdsMax8 = 3 * (128 / 16) * 2 * 64 * 16, kLdsMax16 = 3 * (256 / 16) * 2 * 64 * 16;      // 48 KB, 96 KB
    if (ci == 128) {
        if (reserve_lds_once((const void*)torgb_wide_split_kernel<8, 256>, kLdsMax8, done8) != hipSuccess) return fail(P3D_ERR_LAUNCH, "torgb_wide_split: cannot reserve %d bytes of LDS", kLdsMax8);
        hipLaunchKernelGGL((torgb_wide_split_kernel<8, 256>), dim3(blocks, n_img), dim3(256), lds, s, a);
    } else {
        if (reserve_lds_once((const void*)torgb_wide_split_kernel<16, 512>, kLdsMax16, done16) != hipSuccess) return fail(P3D_ERR_LAUNCH, "torgb_wide_split: cannot reserve %d bytes of LDS", kLdsMax16);
        hipLaunchKernelGGL((torgb_wide_split_kernel<16, 512>), dim3(blocks, n_img), dim3(512), lds, s, a);
    }
    count_launch(FAM_CONV);
    return check_launch("torgb_wide_split");
}
