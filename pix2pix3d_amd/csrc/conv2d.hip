// Modulated 3x3 convolution of the StyleGAN2 synthesis layers as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces, for fp16 channels-last activations, the chain the reference runs per layer in inference
// (training/networks_stylegan2.py:34-91 fused branch + :313-332): per-sample weight modulation/demodulation
// (elementwise torch ops) -> reshape to one grouped conv (cuDNN through conv2d_gradfix.py:127-129) -> + noise ->
// bias_act.  Here:
//   p3d_modulate_weights   w[O,I,k,k] fp32, styles[N,I] -> per-sample fp16 weights [N][O][k*k][I] (tap-major K),
//                          demodulation folded in (fp32 math, one rounding to fp16) — one launch, one pass.
//   p3d_conv2d_nhwc_f16    y[n, p, o] = epilogue( sum_{tap, i} x[n, p + d(tap), i] * wm[n][o][tap][i] ), fp32 accumulate
//                          on v_mfma_f32_32x32x16_f16; epilogue = (+ noise[p] * strength) (+ bias[o]) lrelu * gain, clamp.
// The same kernel runs the stride-2 transposed conv of the upsampling layers (conv2d_resample.py:114-127) as four
// polyphase sub-problems: output parity class (py, px) only sees taps with ky == py, kx == px (mod 2), i.e. a dense
// conv over the low-res grid with 4 / 2 / 2 / 1 taps written to strided output positions — no zero-stuffing, no
// col2im.
//
// Tiling: 256 threads = 4 waves in a 2 x 2 grid, block tile 128 pixels x 128 output channels, K step 64 input
// channels of one tap.  A (pixels x K) and B (channels x K) tiles are staged through LDS in 16-byte chunks with an
// XOR swizzle (chunk ^= row & 7) so the column-slice ds_read_b128 of the MFMA fragments are conflict-free; the next
// K step's global loads are issued before the current step's MFMAs and written to the other LDS buffer after them
// (register-staged double buffering, one barrier per step).  A rows are gathered per pixel (zero-filled at the
// image border), so a tile may straddle image rows.  blockIdx.x walks pixel tiles fastest with the image index in
// blockIdx.z, so concurrently resident blocks share one image's weight panel (1.2 MB, L2-resident).
#include "p3d_common.h"
#include <stdlib.h>

namespace p3d {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr int BM = 128, BN = 128;          // block tile; the K step is one 128-byte row: 64 halfs or 32 floats

// ---- "bf16x3": fp32 convolution on the bf16 matrix cores --------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and the five big backbone layers are bound by it (117 of 157 TFLOP/s).  With
// every fp32 operand written as hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significand bits), x * w = xh*wh + xh*wl + xl*wh up to
// 2^-16 relative — three bf16 MFMAs with fp32 accumulation instead of one fp32 MFMA at 1/16 of their rate.  Activations stay fp32 in
// HBM and LDS and are split in registers (v_cvt_pk_bf16_f32, VALU slots beside the MFMAs); weights are split once by
// p3d_modulate_weights into 128-byte K rows [32 x hi | 32 x lo], i.e. the SAME bytes per row as 32 fp32 values, so staging is
// untouched.  Selected by dtype P3D_F32_BF16X3; exact fp32 remains the default (P3D_F32).
#ifndef P3D_BF16_TERMS
#define P3D_BF16_TERMS 3
#endif
constexpr int kBf16Terms = P3D_BF16_TERMS;     // 3: xh*wh + xh*wl + xl*wh (term order: hh, hl, lh); 4 adds xl*wl (measured: no accuracy gain)
#include "bf16_split.h"

struct ConvTap { int dy, dx, widx; };

struct ConvArgs {
    const void* x;         // [N][H][W][Ci]           fp16 or fp32 (kernel template)
    const void* w;         // [N or 1][Co][KT][Ci]    (KT = taps in the weight tensor: 9 for 3x3, 1 for 1x1)
    void* y;               // [N][OH][OW][Co]
    const float* bias;     // [Co] or null
    const float* noise;    // [OH][OW] or null
    const float* noise_strength;   // device scalar (used when noise != null)
    const void* zeros;     // >= 128 bytes of zeros, 16-byte aligned: source of out-of-image / out-of-range rows
    int N, H, W, Ci, Co, KT;
    int64_t w_img_stride;  // elements between images' weight panels (0: shared weights)
    int OH, OW;            // full output size
    int osy, osx;          // output pixel = (i * osy + ooy, j * osx + oox)
    int isy, isx;          // input pixel of tap (dy, dx) = (i * isy + dy, j * isx + dx)   (2 for the stride-2 'down' convolution)
    int ncls;              // sub-problems solved by this launch (1: plain conv; 4: parity classes of the stride-2 transposed conv)
    int cls_major;         // != 0 (generic kernel, ncls > 1): blockIdx.z = class * N + image instead of image * ncls + class — the classes are ordered by tap count
                           // (4, 2, 2, 1), so the blocks with the longest K loops are dispatched first and the launch's tail is made of the shortest ones
    struct Cls { int SH, SW, ooy, oox, ntaps; ConvTap taps[9]; } cls[4];     // plain conv uses cls[0] with up to 9 taps; transposed classes have <= 4
    int act;               // 0: none (linear), 1: lrelu(0.2)
    float gain, clamp;     // clamp < 0: off
    int fold;              // != 0 (generic kernel, one class, shared weights): GEMM rows run over ALL images' pixels, m = n * SH * SW + pixel,
                           // so a batch of small images fills the 128-row tiles that one image cannot
    int ksplit;            // > 1 (generic kernel): the K steps of a tile are dealt to `ksplit` work-groups (blockIdx.z % ksplit) whose fp32
    float* partial;        // partial tiles go to partial[split][z][Mpad][CoP] and are summed + finished by splitk_epilogue_kernel — the
                           // low-resolution layers (K = 4608, a handful of tiles) otherwise run 144 serial K steps on 8 of 256 CUs
    // fused ToRGB (conv3x3_h2_f16_kernel with Co == 128, no noise): the block's finished [256 pixels][128 channels] tile is contracted with
    // the image's modulated 1x1 weights and ADDED into the fp32 NCHW skip image — the layer's output is not read back by a ToRGB launch
    const float* rgb_w;    // [N][rgb_co][128] fp32 (weight * styles), or null
    const float* rgb_bias; // [rgb_co] or null
    float* rgb_out;        // [N][rgb_co][H][W] fp32, accumulated into
    int rgb_co;            // <= 8
    float rgb_clamp;       // on (sum + bias) before the accumulation; < 0: off
    int store_narrow;      // A/B only (P3D_CONV_STORE4=1): the generic kernel's fp32 epilogue with one 4-byte store per value instead of quad-transposed 16-byte ones
    int cb_loop;           // conv3x3_h2_f16_kernel<false> with the fused ToRGB: channel blocks of 128 one work-group walks (grid.y = Co / 128 / cb_loop); 0 / 1 = one
    const float* oscale;   // [N][Co] or null (generic kernel, fp32 tensors): the accumulator is multiplied by oscale[image][channel] before the
                           // rest of the epilogue — the demodulation coefficient of the SHARED-weight form of the modulated convolution
    const float* iscale;   // [N][Ci] or null (generic kernel, bf16x3 on plain fp32 activations: conv2d_nhwc_kernel<float, true, false, false, true>): every activation is
                           // multiplied by iscale[image][channel] on its way from LDS into the MFMA fragments — the style modulation of the SHARED-weight form
                           // (networks_stylegan2.py:70-79: x * styles), without a pass over x of its own
    int y_split;           // != 0 (conv3x3_halo_kernel<float, true, .>): y leaves in the bf16x3 K-row layout — per 32 channels [32 x bf16 hi | 32 x bf16 lo],
                           // the same 128 bytes as 32 floats — which the next bf16x3 layer reads without splitting anything (XS below)
};

static thread_local const float* tl_in_scale = nullptr;      // ConvArgs::iscale of the call in flight on this thread (p3d_conv2d_nhwc_scaled_in)

// 16-B slot of (row, chunk).  Two 128-byte tile rows share one 256-byte LDS bank row, so the XOR key is (row >> 1) & 7:
// the 16 rows a ds_read_b128 lane group touches then land on 16 distinct slots (row & 7 would leave a 2-way conflict).
__device__ __forceinline__ int swz(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

template <class T> struct ConvTraits;
template <> struct ConvTraits<__half> { static constexpr int BK = 64; };      // elements per 128-byte K row
template <> struct ConvTraits<float>  { static constexpr int BK = 32; };

// XS (bf16x3 only): the activations arrive ALREADY split — K rows of [32 x bf16 hi | 32 x bf16 lo] written by the producing layer's epilogue
// (ConvArgs::y_split, fir4_cl_fused_kernel) — so A fragments are two plain 16-byte LDS reads like the weights': the in-register split (8
// conversions + 8 subtractions + the conversion -> MFMA guard per fragment, repeated by every tap and every wave that reads the pixel) cost
// the bf16x3 kernels 13-21 % (profiles/round3_ae_*).  Staging is unchanged: the rows have the same bytes either way.
// CO64: the launch has at most 64 output channels (one column block, half of its weight tile zeros).  The four waves then take 32 GEMM rows each against both
// 32-column tiles instead of a 64 x 64 quadrant each — two of which would multiply the zero half (the 128 -> 64 channel transposed convolutions of the
// discriminators' data gradient ran at 47 TFLOP/s in fp32 where their 128-column neighbours run at 90).
// ISC: ConvArgs::iscale.  The scales a tile can need — its image's row, or with the batch folded into the GEMM rows the rows of the (few) images its 128 GEMM rows
// belong to — sit in an 8 KB LDS table (host: images per tile x Ci <= kIscaleFloats), read 32 bytes per fragment piece next to the activation itself.
constexpr int kIscaleFloats = 2048;
template <class T, bool BF3 = false, bool XS = false, bool CO64 = false, bool ISC = false, bool X6 = false>
__global__ void __launch_bounds__(256, 2) conv2d_nhwc_kernel(ConvArgs a)
{
    static_assert(!X6 || (sizeof(T) == 4 && !BF3 && !XS && !ISC), "bf16x6 is an arithmetic of the plain fp32 kernel");
    static_assert(!ISC || (BF3 && !XS && !CO64), "the input scale rides on the in-register split of plain fp32 activations");
    static_assert(!BF3 || sizeof(T) == 4, "bf16x3 is a formulation of the fp32 convolution");
    static_assert(!XS || BF3, "pre-split activations are the bf16x3 kernels' input format");
    constexpr int BK = ConvTraits<T>::BK;
    constexpr int EPC = 16 / sizeof(T);                                        // elements per 16-byte chunk
    __shared__ __attribute__((aligned(16))) f32x4 lds[2][2][BM * 8];          // [buffer][A|B][row*8 + chunk], 16-byte slots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(!CO64 || !BF3, "the 64-column form is instantiated for the fp32-layout kernels only");
    constexpr int NI = CO64 ? 1 : 2;                                           // 32-row tiles per wave
    const int rbase = CO64 ? wave * 32 : (wave >> 1) * 64, wn = CO64 ? 0 : wave & 1;   // wave's first tile row / 64-column half
    const int split = blockIdx.z % a.ksplit, zz = blockIdx.z / a.ksplit;
    const int n = a.fold ? 0 : (a.cls_major ? zz % a.N : zz / a.ncls);
    const ConvArgs::Cls& kc = a.cls[a.fold ? 0 : (a.cls_major ? zz / a.N : zz - n * a.ncls)];
    // XCD-aware tile order: workgroup L of a launch lands on XCD L % 8, each XCD with its own L2.  Consecutive slots
    // of one XCD get the output-channel blocks of the SAME pixel tile (they share the A operand), and pixel tiles
    // stride over XCDs, so an A neighbourhood is fetched into one L2 only.  Falls back to the plain order when the
    // grid does not split evenly.
    int mt = blockIdx.x, cb = blockIdx.y;
    {
        const int nmt = gridDim.x, ncb = gridDim.y, L = blockIdx.x + blockIdx.y * nmt;
        if ((nmt & 7) == 0) { const int q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    }
    const int m0 = mt * BM, co0 = cb * BN;
    const int MI = kc.SH * kc.SW;                                              // GEMM rows of one image
    const int M = a.fold ? MI * a.N : MI;
    if (m0 >= M) return;                                                       // classes of one launch differ slightly in size
    const T* xin = (const T*)a.x + (int64_t)n * a.H * a.W * a.Ci;
    const T* wgt = (const T*)a.w + (int64_t)n * a.w_img_stride;

    // staging assignment: 8 threads cover one 128-byte row (64 halfs); 32 rows per pass, 4 passes for 128 rows.
    // Everything lane-dependent is computed once: per K step a DMA source costs one add, four compares and a select.
    const int chunk = tid & 7, srow = tid >> 3;
    const int src_chunk = chunk ^ ((srow >> 1) & 7);               // ((srow + 32p) >> 1) & 7 == (srow >> 1) & 7
    int pi[4], pj[4], poff[4], woff[4];
    bool pok[4], wok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + srow + 32 * p;
        pok[p] = m < M;
        int mm = pok[p] ? m : 0;
        const int img = a.fold ? mm / MI : 0;                                               // folded: the row's own image
        mm -= img * MI;
        pi[p] = mm / kc.SW; pj[p] = mm - pi[p] * kc.SW;
        pi[p] *= a.isy; pj[p] *= a.isx;                                                     // from here on: the tap-(0,0) input pixel
        poff[p] = (((img * a.H + pi[p]) * a.W + pj[p]) * a.Ci + src_chunk * EPC) * (int)sizeof(T);   // < 2^31: one image's activations (the folded batch is small by construction)
        const int co = co0 + srow + 32 * p;
        wok[p] = co < a.Co;
        woff[p] = (co * a.KT * a.Ci + src_chunk * EPC) * (int)sizeof(T);
    }
    const int kchunks = a.Ci / BK;
    const int ntaps = kc.ntaps;

    // Direct global -> LDS staging (global_load_lds_dwordx4): each wave instruction deposits 64 x 16 B = eight 128-byte
    // rows linearly at a wave-uniform LDS base, so the XOR swizzle is applied on the SOURCE side: the lane that fills
    // LDS slot (row, pos) fetches chunk pos ^ ((row >> 1) & 7) of that row.  Rows outside the image (or past the end of the
    // pixel / channel range) read from a page of zeros instead.
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const char* const xin_b = (const char*)xin;
    const char* const wgt_b = (const char*)wgt;
    auto stage = [&](int cc, int t, int buf) {                     // taps innermost: a block re-reads its channel chunk while it is L2-hot
        const ConvTap tp = kc.taps[t];
        const int xs = ((tp.dy * a.W + tp.dx) * a.Ci + cc * BK) * (int)sizeof(T);
        const int ws = (tp.widx * a.Ci + cc * BK) * (int)sizeof(T);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int iy = pi[p] + tp.dy, ix = pj[p] + tp.dx;
            const bool ok = pok[p] & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
            const char* src = ok ? xin_b + (poff[p] + xs) : (const char*)a.zeros;
            const int row0 = (wave * 8 + 32 * p) * 8;      // first 16-B slot of this wave's 8-row group
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&lds[buf][0][row0], 16, 0, 0);
            const char* wsrc = wok[p] ? wgt_b + (woff[p] + ws) : (const char*)a.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr)wsrc, (lds_ptr)&lds[buf][1][row0], 16, 0, 0);
        }
    };

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K steps of this work-group: all of them, or its share of the split
    const int nsteps = kchunks * ntaps;
    const int steps_per = (nsteps + a.ksplit - 1) / a.ksplit;
    const int s_begin = split * steps_per;
    int s_end = s_begin + steps_per;
    if (s_end > nsteps) s_end = nsteps;
    int cc = s_begin / ntaps, t = s_begin - cc * ntaps;
    if (s_begin < s_end) stage(cc, t, 0);
    const int frow = lane & 31, fk = lane >> 5;                                 // fragment row / k-group of this lane
    __shared__ __attribute__((aligned(16))) float isc_tab[ISC ? kIscaleFloats : 4];
    int isc_off[2] = {0, 0};                                                    // table offset of the image of this lane's fragment rows (i = 0, 1)
    if constexpr (ISC) {                                                        // (the table's loads share the first tile's round trip: one rendezvous for both — they came one after the other)
        const int img0 = a.fold ? m0 / MI : n;
        const int img1 = a.fold ? min((min(m0 + BM, M) - 1) / MI, a.N - 1) : n;
        for (int e = tid; e < (img1 - img0 + 1) * a.Ci; e += 256) isc_tab[e] = a.iscale[(int64_t)img0 * a.Ci + e];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(m0 + rbase + i * 32 + frow, M - 1);
            isc_off[i] = (a.fold ? m / MI - img0 : 0) * a.Ci + 8 * fk;
        }
    }
    __syncthreads();
    int pa[NI][4], pb[2][4];                                                    // fragment slots of this lane (buffer 0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (i < NI) pa[i][kk] = swz(rbase + i * 32 + frow, kk * 2 + fk);
            pb[i][kk] = swz(wn * 64 + i * 32 + frow, kk * 2 + fk);
        }
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
            const int cc_cur = cc;                                              // the channel chunk of THIS step's tile (ISC)
            {                                                                   // next tile streams into the other buffer under the MFMAs
                if (++t == ntaps) { t = 0; ++cc; }
                if (s + 1 < s_end) stage(cc, t, buf ^ 1);
            }
            if constexpr (BF3) {                                                // 2 x 16 channels: lane (frow, fk) owns channels 16 m + 8 fk + 0..7
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ra = rbase + i * 32 + frow, rb = wn * 64 + i * 32 + frow;
                        if constexpr (XS) {
                            ah[i] = __builtin_bit_cast(bf8, lds[buf][0][swz(ra, 2 * m + fk)]);
                            al[i] = __builtin_bit_cast(bf8, lds[buf][0][swz(ra, 4 + 2 * m + fk)]);
                        } else if constexpr (ISC) {                             // x * styles in registers: the same fp32 product scale_input() would have stored
                            const float* sp = isc_tab + isc_off[i] + cc_cur * 32 + 16 * m;
                            f32x4 p0 = lds[buf][0][swz(ra, 4 * m + 2 * fk)] * *(const f32x4*)sp, p1 = lds[buf][0][swz(ra, 4 * m + 2 * fk + 1)] * *(const f32x4*)(sp + 4);
                            asm volatile("" : "+v"(p0), "+v"(p1));              // the ROUNDED product is what gets split (no contraction into the split's subtraction)
                            split_bf16x8(p0, p1, ah[i], al[i]);
                        } else
                        split_bf16x8(lds[buf][0][swz(ra, 4 * m + 2 * fk)], lds[buf][0][swz(ra, 4 * m + 2 * fk + 1)], ah[i], al[i]);
                        bh[i] = __builtin_bit_cast(bf8, lds[buf][1][swz(rb, 2 * m + fk)]);          // weight row = [32 hi | 32 lo] bf16
                        bl[i] = __builtin_bit_cast(bf8, lds[buf][1][swz(rb, 4 + 2 * m + fk)]);
                    }
#pragma unroll
                    for (int term = 0; term < kBf16Terms; ++term)               // term outermost: consecutive MFMAs never share an accumulator
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term >= 2 ? al[i] : ah[i], (term & 1) ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
                }
            } else if constexpr (X6) {                                          // fp32 tiles on both sides, split in registers: 2 x 16 channels, six bf16 MFMAs per product tile
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf8 ah[NI], am[NI], al[NI], bh[2], bm[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i < NI) { const int ra = rbase + i * 32 + frow; split3_bf16x8(lds[buf][0][swz(ra, 4 * m + 2 * fk)], lds[buf][0][swz(ra, 4 * m + 2 * fk + 1)], ah[i], am[i], al[i]); }
                        const int rb = wn * 64 + i * 32 + frow;
                        split3_bf16x8(lds[buf][1][swz(rb, 4 * m + 2 * fk)], lds[buf][1][swz(rb, 4 * m + 2 * fk + 1)], bh[i], bm[i], bl[i]);
                    }
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, ah[i], am[i], al[i]), P3D_X6_B(term, bh[j], bm[j], bl[j]), acc[i][j], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {                                    // 4 x 32 bytes of K per 128-byte row
                f32x4 fa[NI], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i < NI) fa[i] = lds[buf][0][pa[i][kk]];
                    fb[i] = lds[buf][1][pb[i][kk]];
                }
#pragma unroll
                for (int e = 0; e < (sizeof(T) == 2 ? 1 : 4); ++e)              // fp32: e outermost, so consecutive MFMAs never share an accumulator
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if constexpr (sizeof(T) == 2) {                     // 8 halfs per lane = one 32x32x16 step
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[i]), __builtin_bit_cast(h8, fb[j]), acc[i][j], 0, 0, 0);
                            } else {                                            // 4 floats per lane = four 32x32x2 steps; K order permuted identically in A and B
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                            }
                        }
            }
            __syncthreads();                                                    // drains the LDS-DMA (vmcnt) and fences the buffer swap
            buf ^= 1;
        }

    if (a.ksplit > 1) {                                                         // split K: raw fp32 partial tile, finished by splitk_epilogue_kernel
        const int Mpad = gridDim.x * BM, CoP = gridDim.y * BN;
        float* out = a.partial + ((int64_t)split * (gridDim.z / a.ksplit) + zz) * Mpad * CoP;
        // (four-byte stores, one 128-byte line per half-wave: the quad-transposed 16-byte form measured 0.4 % SLOWER on the line here, profiles/round6_f_*)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = co0 + wn * 64 + j * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    out[(int64_t)row * CoP + col] = acc[i][j][r];
                }
            }
        return;
    }

    // ---- epilogue: accumulator element (row = (r&3) + 8(r>>2) + 4*fk, col = frow) of each 32x32 tile -------------
    const float ns = a.noise ? a.noise_strength[0] : 0.f;
    if constexpr (sizeof(T) == 2) {
        // fp16: a lane holds ONE channel of 64 output pixels — as direct stores, 64 two-byte writes.  The tile is transposed
        // through LDS instead ([128 pixels][128 channels], pitch 136 halfs so the fk = 1 half-wave lands 16 banks away) and
        // leaves as 16-byte stores, 16 lanes per pixel = that pixel's whole 256-byte channel run; the pixel -> (y, x)
        // division is done 8 times per lane, not 64.  (The loop's last barrier has retired every fragment read.)
        constexpr int OP = 136;
        __half* const ot = (__half*)lds;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + frow, co = co0 + cl;
            const float b = (a.bias && co < a.Co) ? a.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    float v = acc[i][j][r] + b;
                    if (!a.noise) {                                             // (with noise the activation waits for the read-back pass)
                        if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                        v *= a.gain;
                        if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                    }
                    ot[ml * OP + cl] = __float2half(v);
                }
        }
        __syncthreads();
        __half* const yout = (__half*)a.y + (int64_t)n * a.OH * a.OW * a.Co;
        const bool vec_ok = ((a.Co & 7) == 0) && ((((uintptr_t)a.y) & 15u) == 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid, ml = idx >> 4, ch = idx & 15;
            int m = m0 + ml;
            const int co = co0 + ch * 8;
            if (m >= M || co >= a.Co) continue;
            const int img = a.fold ? m / MI : 0;
            m -= img * MI;
            const int si = m / kc.SW, sj = m - si * kc.SW;
            const int oy = si * a.osy + kc.ooy + img * a.OH, ox = sj * a.osx + kc.oox;      // folded: images are stacked along y in the output
            if (oy >= a.OH * (img + 1) || ox >= a.OW) continue;
            f32x4 pk = *(const f32x4*)(ot + ml * OP + ch * 8);
            if (a.noise) {                                                      // rare here (the big noisy layers take the halo kernels)
                const float nz = a.noise[(int64_t)(oy - img * a.OH) * a.OW + ox] * ns;
                h8 hv = __builtin_bit_cast(h8, pk);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)hv[e] + nz;
                    if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                    v *= a.gain;
                    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                    hv[e] = (_Float16)v;
                }
                pk = __builtin_bit_cast(f32x4, hv);
            }
            __half* dst = yout + ((int64_t)oy * a.OW + ox) * a.Co + co;
            if (vec_ok) *(f32x4*)dst = pk;
            else {
                const h8 hv = __builtin_bit_cast(h8, pk);
#pragma unroll
                for (int e = 0; e < 8; ++e) if (co + e < a.Co) ((_Float16*)dst)[e] = hv[e];
            }
        }
    } else {
        int opix[NI][16];                                                        // output pixel offset of each accumulator row (or -1)
        // one division for the wave's first row, then every row by carry (its 32 rows span 64 consecutive m): the 32 runtime
        // divisions this replaces were a fifth of a short-K (1x1) block's life
        const int mb = m0 + rbase;
        const int imgb = a.fold ? mb / MI : 0;
        const int sib = (mb - imgb * MI) / kc.SW, sjb = (mb - imgb * MI) - sib * kc.SW;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                int si = sib, sj = sjb + d, img = imgb;
                while (sj >= kc.SW) { sj -= kc.SW; ++si; }
                while (a.fold && si >= kc.SH) { si -= kc.SH; ++img; }
                const int oy = si * a.osy + kc.ooy, ox = sj * a.osx + kc.oox;
                opix[i][r] = (mb + d < M && oy < a.OH && ox < a.OW) ? (img * a.OH + oy) * a.OW + ox : -1;    // folded: + the row's image
            }
        if constexpr (sizeof(T) == 4) {
            if (((a.Co & 3) == 0) && ((((uintptr_t)a.y) & 15u) == 0) && !a.store_narrow) {
                // 16-byte stores: a lane holds ONE channel of four consecutive GEMM rows in registers 4 q .. 4 q + 3 and the four lanes of a quad four consecutive
                // channels; the quad transposes its 4 x 4 values in registers (two DPP stages) and lane 4 m + t then stores channels 4 m .. 4 m + 3 of row t — 16
                // store instructions per lane and tile instead of 64 four-byte ones (conv3x3_r2_bf16x3_kernel's epilogue: the same change was -9 % of that kernel)
                const int t = lane & 3, m4 = (lane & 31) >> 2;
                const bool odd1 = lane & 1, odd2 = lane & 2;
                float bj2[2];                                                       // (both column blocks' biases in one batch)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = co0 + wn * 64 + j * 32 + frow;
                    const bool bok = a.bias && co < a.Co;
                    bj2[j] = (bok ? a.bias : (const float*)a.zeros)[bok ? co : 0];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cq = co0 + wn * 64 + j * 32, co = cq + frow;
                    const bool cok = co < a.Co;
                    const float b = bj2[j];
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            unsigned w[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int px = opix[i][4 * q + e];
                                float v = acc[i][j][4 * q + e];
                                if (px >= 0 && cok) {
                                    if (a.oscale) v *= a.oscale[(a.fold ? px / (a.OH * a.OW) : n) * a.Co + co];
                                    if (a.noise) v = fmaf(a.noise[a.fold ? px % (a.OH * a.OW) : px], ns, v);
                                }
                                v += b;
                                if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                                v *= a.gain;
                                if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                                w[e] = __float_as_uint(v);
                            }
                            quad_transpose4(w, odd1, odd2);
                            const int px0 = opix[i][4 * q], px1 = opix[i][4 * q + 1], px2 = opix[i][4 * q + 2], px3 = opix[i][4 * q + 3];
                            const int mine = odd2 ? (odd1 ? px3 : px2) : (odd1 ? px1 : px0);
                            if (mine >= 0 && cq + 4 * m4 < a.Co) {
                                typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
                                *(u32x4e*)((float*)a.y + ((int64_t)n * a.OH * a.OW + mine) * a.Co + cq + 4 * m4) = u32x4e{w[0], w[1], w[2], w[3]};
                            }
                        }
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = co0 + wn * 64 + j * 32 + frow;
            if (co >= a.Co) continue;
            const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (opix[i][r] < 0) continue;
                    float v = acc[i][j][r];
                    if (a.oscale) v *= a.oscale[(a.fold ? opix[i][r] / (a.OH * a.OW) : n) * a.Co + co];
                    if (a.noise) v = fmaf(a.noise[a.fold ? opix[i][r] % (a.OH * a.OW) : opix[i][r]], ns, v);
                    v += b;
                    if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                    v *= a.gain;
                    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                    st((T*)a.y + ((int64_t)n * a.OH * a.OW + opix[i][r]) * a.Co + co, v);
                }
        }
    }
}

// ---- finish of the split-K launches: sum the partial tiles, then the epilogue and the store of conv2d_nhwc_kernel ----------------------
// One thread = one GEMM row (output pixel) x 4 consecutive output channels of one z-slice (image x class, or the folded batch).
template <class T>
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(ConvArgs a, int Mpad, int CoP, int zcount)
{
    const int co4 = (a.Co + 3) / 4;
    const int64_t per_z = (int64_t)Mpad * co4;
    const float ns = a.noise ? a.noise_strength[0] : 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per_z * zcount; e += (int64_t)gridDim.x * blockDim.x) {
        const int zz = (int)(e / per_z);
        const int64_t rem = e - zz * per_z;
        int m = (int)(rem / co4);
        const int co = (int)(rem - (int64_t)m * co4) * 4;
        int n = a.fold ? 0 : (a.cls_major ? zz % a.N : zz / a.ncls);
        const ConvArgs::Cls& kc = a.cls[a.fold ? 0 : (a.cls_major ? zz / a.N : zz - n * a.ncls)];
        const int MI = kc.SH * kc.SW;
        if (m >= (a.fold ? MI * a.N : MI)) continue;
        const float* src = a.partial + ((int64_t)zz * Mpad + m) * CoP + co;
        const int64_t split_stride = (int64_t)zcount * Mpad * CoP;
        if (a.fold) { n = m / MI; m -= n * MI; }
        const int si = m / kc.SW, sj = m - si * kc.SW;
        const int oy = si * a.osy + kc.ooy, ox = sj * a.osx + kc.oox;
        if (oy >= a.OH || ox >= a.OW) continue;
        // the row's noise, scales and biases: requested ahead of the partial tiles (one batch, no load between two stores); the partial tiles eight / four / two at a time,
        // ADDED in k order as before.  As written first — `for k: sum += load`, then `if (oscale) v *= load; if (bias) v += load; store` per channel — a thread ran
        // ksplit + 8 dependent memory round trips, which is what these 5 - 15 us launches on the low-resolution layers' critical path consisted of.
        const float nz = a.noise ? a.noise[(int64_t)oy * a.OW + ox] * ns : 0.f;
        float osv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.oscale) {
#pragma unroll
            for (int c = 0; c < 4; ++c) osv[c] = a.oscale[n * a.Co + min(co + c, a.Co - 1)];
        }
        if (a.bias) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bsv[c] = a.bias[min(co + c, a.Co - 1)];
        }
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= a.ksplit; k += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(src + (k + u) * split_stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        if (k + 4 <= a.ksplit) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const f32x4*)(src + (k + u) * split_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += t[u];
            k += 4;
        }
        if (k + 2 <= a.ksplit) {
            const f32x4 t0 = *(const f32x4*)(src + k * split_stride), t1 = *(const f32x4*)(src + (k + 1) * split_stride);
            sum += t0; sum += t1;
            k += 2;
        }
        if (k < a.ksplit) sum += *(const f32x4*)(src + k * split_stride);
        T* dst = (T*)a.y + (((int64_t)n * a.OH + oy) * a.OW + ox) * a.Co + co;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (co + c >= a.Co) break;
            float v = sum[c];
            if (a.oscale) v *= osv[c];
            v += nz;
            if (a.bias) v += bsv[c];
            if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
            v *= a.gain;
            if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
            st(dst + c, v);
        }
    }
}

// Epilogue of the 8 x 16-patch kernels (conv3x3_halo_kernel, conv3x3_halo_x6p_kernel): tile i of a wave is patch rows prow0 + 2 i, + 1, column block j is
// channels co0 + 64 wn + 32 j + (lane & 31); accumulator register r of lane (frow, fk) is tile row (r & 3) + 8 (r >> 2) + 4 fk.
template <class T, bool BF3>
__device__ __forceinline__ void halo_epilogue(const ConvArgs& a, f32x16 (&acc)[2][2], int n, int oy0, int ox0, int co0, int prow0, bool co64, int wn, int lane)
{
    const int frow = lane & 31, fk = lane >> 5;
    const float ns = a.noise ? a.noise_strength[0] : 0.f;
    // this lane's 32 noise values (one per accumulator row of its two tiles; the same for every column block) in ONE batch of unconditional loads — read where they were
    // used, under `if (noise && inside)`, each was a branch with its own wait: 32 dependent memory round trips in the epilogue of every work-group of a layer with noise
    // (the exact-fp32 and bf16x6 inference legs' 3x3 layers)
    float nzr[2][16];
    {
        const float* const np = a.noise ? a.noise : (const float*)a.zeros;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int oy = oy0 + prow0 + i * 2 + (mrow >> 4), ox = ox0 + (mrow & 15);
                const bool ok = a.noise && oy < a.H && ox < a.W;
                nzr[i][r] = np[ok ? (int64_t)oy * a.W + ox : 0];
            }
    }
    float bjh[2];                                                               // (and the two column blocks' biases)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + wn * 64 + j * 32 + frow;
        const bool bok = a.bias && co < a.Co;
        bjh[j] = (bok ? a.bias : (const float*)a.zeros)[bok ? co : 0];
    }
    if (!a.y_split && !a.store_narrow && (a.Co & 3) == 0 && ((((uintptr_t)a.y) & 15u) == 0)) {
        // Wide stores (quad_transpose4, p3d_common.h): after the 4 x 4 transpose lane 4 m + t holds channels 4 m .. 4 m + 3 of pixel t of the four consecutive pixels in
        // registers 4 q .. 4 q + 3 — one 16-byte store (fp32) or, with the two row tiles i = 0, 1 packed into one dword per value, two 8-byte ones (fp16) where the
        // lane = channel layout stores one value per instruction (with the generic kernel's: 272.9 -> 270.5 ms per training iteration, profiles/round6_j_*).
        const bool odd1 = lane & 1, odd2 = lane & 2;
        const int t = lane & 3, m4 = frow >> 2;
        T* const yimg = (T*)a.y + (int64_t)n * a.H * a.W * a.Co;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cq = co0 + wn * 64 + j * 32, co = cq + frow;
            const bool cok = co < a.Co;
            const float b = bjh[j];
            float fin[2][16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mrow = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const int oy = oy0 + prow0 + i * 2 + (mrow >> 4), ox = ox0 + (mrow & 15);
                    float v = (co64 && i == 1) ? 0.f : acc[i][j][r];
                    if (a.noise && oy < a.H && ox < a.W) v = fmaf(nzr[i][r], ns, v);
                    v += b;
                    if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                    v *= a.gain;
                    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                    fin[i][r] = v;
                }
            const bool chan_ok = cq + 4 * m4 < a.Co;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mrow = t + 8 * q + 4 * fk;                           // this lane's pixel after the transpose
                const int ox = ox0 + (mrow & 15);
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (co64 && i == 1) continue;
                        unsigned w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(fin[i][4 * q + e]);
                        quad_transpose4(w, odd1, odd2);
                        const int oy = oy0 + prow0 + i * 2 + (mrow >> 4);
                        if (chan_ok && oy < a.H && ox < a.W) {
                            typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
                            *(u32x4h*)(yimg + ((int64_t)oy * a.W + ox) * a.Co + cq + 4 * m4) = u32x4h{w[0], w[1], w[2], w[3]};
                        }
                    }
                } else {
                    unsigned w[4];                                              // (value of row tile 0 | value of row tile 1 << 16) per register
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
                        hp2 pk; pk[0] = (_Float16)fin[0][4 * q + e]; pk[1] = (_Float16)fin[1][4 * q + e];
                        w[e] = __builtin_bit_cast(unsigned, pk);
                    }
                    quad_transpose4(w, odd1, odd2);
                    typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (co64 && i == 1) continue;
                        const unsigned sel = i ? 0x07060302u : 0x05040100u;
                        const u32x2h out = {__builtin_amdgcn_perm(w[1], w[0], sel), __builtin_amdgcn_perm(w[3], w[2], sel)};
                        const int oy = oy0 + prow0 + i * 2 + (mrow >> 4);
                        if (chan_ok && oy < a.H && ox < a.W) *(u32x2h*)(yimg + ((int64_t)oy * a.W + ox) * a.Co + cq + 4 * m4) = out;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + wn * 64 + j * 32 + frow;
        if (co >= a.Co) continue;
        const float b = bjh[j];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (co64 && i == 1) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = (r & 3) + 8 * (r >> 2) + 4 * fk;             // accumulator row within the 32-row tile
                const int oy = oy0 + prow0 + i * 2 + (mrow >> 4), ox = ox0 + (mrow & 15);
                if (oy >= a.H || ox >= a.W) continue;
                float v = acc[i][j][r];
                if (a.noise) v = fmaf(nzr[i][r], ns, v);
                v += b;
                if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                v *= a.gain;
                if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                if constexpr (BF3) {
                    if (a.y_split) {                                            // this lane's channel inside its 32-channel K row: hi at [frow], lo at [32 + frow]
                        __bf16* row = (__bf16*)((float*)a.y + (((int64_t)n * a.H + oy) * a.W + ox) * a.Co + (co - frow));
                        const __bf16 hv = (__bf16)v;
                        row[frow] = hv;
                        row[32 + frow] = (__bf16)(v - (float)hv);
                        continue;
                    }
                }
                st((T*)a.y + (((int64_t)n * a.H + oy) * a.W + ox) * a.Co + co, v);
            }
        }
    }
}

// ---- 3x3 "same" convolution with halo reuse ---------------------------------------------------------------------------
// Same GEMM tiling as above, but the A operand of a block — an 8 x 16 pixel patch — is staged ONCE per 128-byte channel
// chunk as a (8+2) x (16+2) pixel slab; the nine taps then read shifted windows of that slab straight from LDS.  Global /
// L2 traffic per K step drops from 32 KB (A + B tile) to 16 KB of weights + 1/9 of a 23 KB slab, and every activation
// byte is fetched from HBM once per output-channel block instead of nine times.
constexpr int PH = 8, PW = 16;                      // pixel patch of a block (PH * PW == BM)
constexpr int SLAB_W = PW + 2, SLAB_ROWS = (PH + 2) * (PW + 2);          // 18, 180 slab pixels
constexpr int SLAB_SLOTS = ((SLAB_ROWS + 7) / 8) * 8 * 8;                // padded to whole 8-row DMA groups, 16-byte slots

template <class T, bool BF3 = false, bool XS = false, bool X6 = false>
__global__ void __launch_bounds__(256, 2) conv3x3_halo_kernel(ConvArgs a)
{
    static_assert(!BF3 || sizeof(T) == 4, "bf16x3 is a formulation of the fp32 convolution");
    static_assert(!XS || BF3, "pre-split activations are the bf16x3 kernels' input format");
    constexpr int BK = ConvTraits<T>::BK;
    constexpr int EPC = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) f32x4 slab[2][SLAB_SLOTS];       // [buffer][slab pixel * 8 + chunk]
    __shared__ __attribute__((aligned(16))) f32x4 wt[2][BN * 8];             // [buffer][co row * 8 + chunk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Co <= 64 (the 64-channel layers at 512^2 of the discriminators and the label-map Encoder): the block's 128-column weight tile is half zeros, and
    // the 2 x 2 quadrant mapping left the waves of the upper column half multiplying them (55 instead of 115 TFLOP/s in fp32, profiles/round4_m_*).  There
    // the four waves take a QUARTER of the patch's pixels each (one 32-row tile) against the 64 live columns: half the MFMAs, all of them useful.
    const bool co64 = a.Co <= 64;
    const int wm = co64 ? 0 : wave >> 1, wn = co64 ? 0 : wave & 1;
    const int prow0 = co64 ? wave * 2 : (wave >> 1) * 4;                     // first patch row of this wave's tile i = 0 (tile i adds 2 rows)
    const int n = blockIdx.z;
    const int tiles_x = (a.W + PW - 1) / PW;
    int mt = blockIdx.x, cb = blockIdx.y;
    {
        const int nmt = gridDim.x, ncb = gridDim.y, L = blockIdx.x + blockIdx.y * nmt;
        if ((nmt & 7) == 0) { const int q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    }
    const int ty = mt / tiles_x, tx = mt - ty * tiles_x;
    const int oy0 = ty * PH, ox0 = tx * PW, co0 = cb * BN;
    const T* xin = (const T*)a.x + (int64_t)n * a.H * a.W * a.Ci;
    const T* wgt = (const T*)a.w + (int64_t)n * a.w_img_stride;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    const int pos = tid & 7, grow = tid >> 3;                               // DMA lane: slot position / row within the 32-row pass
    const int kchunks = a.Ci / BK;
    auto stage_slab = [&](int cc, int buf) {
#pragma unroll
        for (int p = 0; p < (SLAB_SLOTS / 8 + 31) / 32; ++p) {              // 6 passes of 32 slab rows
            const int row = grow + 32 * p;
            if (row >= SLAB_SLOTS / 8) break;                               // wave-uniform: rows of a wave are contiguous groups of 8
            const int sr = row / SLAB_W, sc = row - sr * SLAB_W;
            const int iy = oy0 - 1 + sr, ix = ox0 - 1 + sc;
            const bool ok = (row < SLAB_ROWS) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
            const int src_chunk = pos ^ ((row >> 1) & 7);
            const T* src = ok ? xin + ((int64_t)iy * a.W + ix) * a.Ci + cc * BK + src_chunk * EPC : (const T*)a.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&slab[buf][(row - (grow & 7)) * 8], 16, 0, 0);
        }
    };
    auto stage_w = [&](int cc, int t, int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = grow + 32 * p, co = co0 + row;
            const int src_chunk = pos ^ ((row >> 1) & 7);
            const T* src = (co < a.Co) ? wgt + ((int64_t)co * 9 + t) * a.Ci + cc * BK + src_chunk * EPC : (const T*)a.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&wt[buf][(wave * 8 + 32 * p) * 8], 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment rows of this lane: MFMA row r of tile i is patch pixel (py, px) = (wm*4 + i*2 + r/16, r%16)
    const int frow = lane & 31, fk = lane >> 5;
    int arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) arow[i] = (prow0 + ((co64 && i == 1) ? 0 : i * 2) + (frow >> 4) + 1) * SLAB_W + (frow & 15) + 1;    // slab pixel under tap (0,0); (co64: tile 1 is not
                                                                                                                                  // used — its fragment reads just stay inside the slab)

    stage_slab(0, 0);
    stage_w(0, 0, 0);
    __syncthreads();
    const int ksteps = 9 * kchunks;
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cc = ks / 9, t = ks - cc * 9;
        const int wb = ks & 1, sb = cc & 1;
        if (ks + 1 < ksteps) {
            const int cc1 = (ks + 1) / 9, t1 = (ks + 1) - cc1 * 9;
            stage_w(cc1, t1, wb ^ 1);
            if (t == 0 && cc + 1 < kchunks) stage_slab(cc + 1, sb ^ 1);     // next chunk's slab streams in under this chunk's nine taps
        }
        const int toff = (t / 3 - 1) * SLAB_W + (t % 3 - 1);
        if constexpr (BF3) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int sr = arow[i] + toff, key = (sr >> 1) & 7, rb = wn * 64 + i * 32 + frow;
                    if constexpr (XS) {
                        ah[i] = __builtin_bit_cast(bf8, slab[sb][sr * 8 + ((2 * m + fk) ^ key)]);
                        al[i] = __builtin_bit_cast(bf8, slab[sb][sr * 8 + ((4 + 2 * m + fk) ^ key)]);
                    } else
                    split_bf16x8(slab[sb][sr * 8 + ((4 * m + 2 * fk) ^ key)], slab[sb][sr * 8 + ((4 * m + 2 * fk + 1) ^ key)], ah[i], al[i]);
                    bh[i] = __builtin_bit_cast(bf8, wt[wb][swz(rb, 2 * m + fk)]);
                    bl[i] = __builtin_bit_cast(bf8, wt[wb][swz(rb, 4 + 2 * m + fk)]);
                }
#pragma unroll
                for (int term = 0; term < kBf16Terms; ++term)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (co64 && i == 1) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term >= 2 ? al[i] : ah[i], (term & 1) ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (X6) {
            static_assert(!X6 || (sizeof(T) == 4 && !BF3 && !XS), "bf16x6 is an arithmetic of the plain fp32 kernel");
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int sr = arow[i] + toff, key = (sr >> 1) & 7, rb = wn * 64 + i * 32 + frow;
                    split3_bf16x8(slab[sb][sr * 8 + ((4 * m + 2 * fk) ^ key)], slab[sb][sr * 8 + ((4 * m + 2 * fk + 1) ^ key)], ah[i], am[i], al[i]);
                    split3_bf16x8(wt[wb][swz(rb, 4 * m + 2 * fk)], wt[wb][swz(rb, 4 * m + 2 * fk + 1)], bh[i], bm[i], bl[i]);
                }
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (co64 && i == 1) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, ah[i], am[i], al[i]), P3D_X6_B(term, bh[j], bm[j], bl[j]), acc[i][j], 0, 0, 0);
                    }
            }
        } else
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sr = arow[i] + toff;
                fa[i] = slab[sb][sr * 8 + ((kk * 2 + fk) ^ ((sr >> 1) & 7))];
                fb[i] = wt[wb][swz(wn * 64 + i * 32 + frow, kk * 2 + fk)];
            }
#pragma unroll
            for (int e = 0; e < (sizeof(T) == 2 ? 1 : 4); ++e)                  // fp32: e outermost, so consecutive MFMAs never share an accumulator
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (co64 && i == 1) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (sizeof(T) == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[i]), __builtin_bit_cast(h8, fb[j]), acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                        }
                    }
                }
        }
        __syncthreads();
    }

    halo_epilogue<T, BF3>(a, acc, n, oy0, ox0, co0, prow0, co64, wn, lane);
}

// ---- bf16x6 on operands that are split ONCE per work-group ("x6p") -------------------------------------------------------------
// conv3x3_halo_kernel<float, false, false, true> keeps fp32 tiles in LDS and splits every fragment it reads: a slab value is split again by each of
// the nine taps and by both waves that share its rows, a weight by both waves that share its columns — 288 vector instructions per 48 MFMAs, and the
// counters show the kernel bound by them (vector issue 0.50 against 0.37 of the matrix pipe, profiles/round6_z_kernel_pmc_train6.txt).  Pre-split
// tiles do not fit the two-blocks-per-CU budget at 32-channel K rows (three 64-byte pieces per row: 114 KB), so this kernel takes 16-channel K rows:
//   * LDS rows are 96 bytes, [hi | mid | lo] x 16 bf16, 16-byte position 2 piece + (half ^ key), key = (row >> 3) & 1: the 16 rows one ds_read_b128
//     cycle serves start on 16 different bank quads (6 r mod 16 takes every even value twice, eight rows apart — where the key differs);
//   * operands travel global -> registers -> split3_bf16x8 -> LDS: a thread owns 8 consecutive channels of one weight row per tap (two 16-byte loads
//     issued two taps ahead, 36 vector instructions, three ds_write_b128) and of at most two slab pixels per chunk — every value is split exactly once;
//   * a tap is 12 fragment reads and 24 MFMAs per wave with no vector work between them; two slab buffers, two weight slots, one barrier per tap;
//     60 KB per block.
// Same products and the same six terms per product as the in-register form; the summation order over K differs (16-channel chunks outermost).
constexpr int X6_ROW = 96;
constexpr int X6_SLAB_BYTES = 184 * X6_ROW;               // 180 slab pixels (+ 4 rows of padding nobody reads)
constexpr int X6_W_BYTES = BN * X6_ROW;
constexpr int X6_W_BASE = 2 * X6_SLAB_BYTES;
constexpr int X6_LDS = X6_W_BASE + 2 * X6_W_BYTES;        // 59 904

__device__ __forceinline__ void x6_put(char* dst, const f32x4& v0, const f32x4& v1)      // dst: the row's hi piece at this thread's (half ^ key) position
{
    bf8 hi, mid, lo;
    split3_bf16x8(v0, v1, hi, mid, lo);
    *(bf8*)dst = hi; *(bf8*)(dst + 32) = mid; *(bf8*)(dst + 64) = lo;
}

template <bool CO64>                                                           // Co <= 64: four row tiles against the 64 live columns (see conv3x3_halo_kernel)
__global__ void __launch_bounds__(256, 2) conv3x3_halo_x6p_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds_b[X6_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool co64 = CO64;
    const int wn = co64 ? 0 : wave & 1;
    const int prow0 = co64 ? wave * 2 : (wave >> 1) * 4;
    const int n = blockIdx.z;
    const int tiles_x = (a.W + PW - 1) / PW;
    int mt = blockIdx.x, cb = blockIdx.y;
    {
        const int nmt = gridDim.x, ncb = gridDim.y, L = blockIdx.x + blockIdx.y * nmt;
        if ((nmt & 7) == 0) { const int q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    }
    const int ty = mt / tiles_x, tx = mt - ty * tiles_x;
    const int oy0 = ty * PH, ox0 = tx * PW, co0 = cb * BN;
    const float* const xin = (const float*)a.x + (int64_t)n * a.H * a.W * a.Ci;
    const float* const wgt = (const float*)a.w + (int64_t)n * a.w_img_stride;
    const float* const zeros = (const float*)a.zeros;

    // producer roles.  Weights: row tid >> 1 of the 128-row tile, channels 8 (tid & 1) .. + 7 of the 16-channel chunk.
    const int wrow = tid >> 1, whalf = tid & 1;
    const bool wok = co0 + wrow < a.Co;
    const float* const wsrc = wgt + (int64_t)(co0 + wrow) * 9 * a.Ci + whalf * 8;                           // + tap * Ci + chunk * 16
    char* const wdst = lds_b + X6_W_BASE + wrow * X6_ROW + ((whalf ^ ((wrow >> 3) & 1)) << 4);             // + slot * X6_W_BYTES
    // slab: items tid and tid + 256 of the 360 (pixel, half) pairs
    const float* ssrc[2]; char* sdst[2]; bool sval[2], sin[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int item = tid + 256 * r, q = item >> 1, half = item & 1;
        const int sr = q / SLAB_W, sc = q - sr * SLAB_W;
        const int iy = oy0 - 1 + sr, ix = ox0 - 1 + sc;
        sval[r] = item < 2 * SLAB_ROWS;
        sin[r] = sval[r] & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        ssrc[r] = xin + ((int64_t)iy * a.W + ix) * a.Ci + half * 8;                                        // + chunk * 16
        sdst[r] = lds_b + q * X6_ROW + ((half ^ ((q >> 3) & 1)) << 4);                                      // + buffer * X6_SLAB_BYTES
    }
    auto load_w = [&](f32x4 (&v)[2], int cc, int t) {
        const float* p = wok ? wsrc + (int64_t)t * a.Ci + cc * 16 : zeros;
        v[0] = *(const f32x4*)p; v[1] = *(const f32x4*)(p + 4);
    };
    auto load_s = [&](f32x4 (&v)[2], int r, int cc) {
        const float* p = sin[r] ? ssrc[r] + cc * 16 : zeros;
        v[0] = *(const f32x4*)p; v[1] = *(const f32x4*)(p + 4);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fk = lane >> 5;
    int arow[2];                                                               // slab pixel under tap (0, 0) of this lane's row in tile i
#pragma unroll
    for (int i = 0; i < 2; ++i) arow[i] = (prow0 + ((co64 && i == 1) ? 0 : i * 2) + (frow >> 4) + 1) * SLAB_W + (frow & 15) + 1;
    int preB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + frow; preB[j] = X6_W_BASE + rb * X6_ROW + ((fk ^ ((rb >> 3) & 1)) << 4); }

    const int nchunks = a.Ci / 16, kpairs = nchunks / 2;                       // (host: Ci % 32 == 0)
    f32x4 wq[2][2], sq[2];
    {   // prologue: slab of chunk 0, weights of tap 0; taps 1 and 2 in flight
        f32x4 s1[2];
        load_s(sq, 0, 0); load_s(s1, 1, 0); load_w(wq[0], 0, 0);
        x6_put(sdst[0], sq[0], sq[1]);
        if (sval[1]) x6_put(sdst[1], s1[0], s1[1]);
        x6_put(wdst, wq[0][0], wq[0][1]);
        load_w(wq[1], 0, 1); load_w(wq[0], 0, 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int cp = 0; cp < kpairs; ++cp) {
        const bool more = cp + 1 < kpairs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = cp * 2 + h;
            const bool next_chunk = (h == 0) || more;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int st = h * 9 + t;                                       // 0..17, compile-time: weight slot st & 1, register set (st + 1) & 1 holds tap st + 1
                const int toff = (t / 3 - 1) * SLAB_W + (t % 3 - 1);
                bf8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = arow[i] + toff;
                    const char* pa = lds_b + h * X6_SLAB_BYTES + q * X6_ROW + ((fk ^ ((q >> 3) & 1)) << 4);
                    ah[i] = *(const bf8*)pa; am[i] = *(const bf8*)(pa + 32); al[i] = *(const bf8*)(pa + 64);
                    const char* pb = lds_b + preB[i] + (st & 1) * X6_W_BYTES;
                    bh[i] = *(const bf8*)pb; bm[i] = *(const bf8*)(pb + 32); bl[i] = *(const bf8*)(pb + 64);
                }
                // tap st + 1's weights: registers -> slot (st + 1) & 1 (its readers left at the last barrier); tap st + 3's loads take the registers over
                {
                    const int t1 = t + 1;                                       // tap st + 1 = (cc, t + 1) or (cc + 1, 0)
                    if (t1 < 9 || next_chunk) x6_put(wdst + ((st + 1) & 1) * X6_W_BYTES, wq[(st + 1) & 1][0], wq[(st + 1) & 1][1]);
                    const int t3 = t + 3;
                    if (t3 < 9) load_w(wq[(st + 1) & 1], cc, t3);
                    else if (next_chunk) load_w(wq[(st + 1) & 1], cc + 1, t3 - 9);
                }
                if (next_chunk) {                                               // next chunk's slab into the other buffer, one item at a time
                    if (t == 0) load_s(sq, 0, cc + 1);
                    if (t == 2) { x6_put(sdst[0] + (h ^ 1) * X6_SLAB_BYTES, sq[0], sq[1]); load_s(sq, 1, cc + 1); }
                    if (t == 4 && sval[1]) x6_put(sdst[1] + (h ^ 1) * X6_SLAB_BYTES, sq[0], sq[1]);
                }
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (co64 && i == 1) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, ah[i], am[i], al[i]), P3D_X6_B(term, bh[j], bm[j], bl[j]), acc[i][j], 0, 0, 0);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    halo_epilogue<float, false>(a, acc, n, oy0, ox0, co0, prow0, co64, wn, lane);
}

// ---- 3x3 conv, 16 x 16 patch, three-slot weight ring, 64-byte K rows: TWO blocks per CU (fp16) -----------------------------------
// The two-stage kernels above wait for the whole prefetch at every barrier, so a K step costs one L2 round trip however little
// compute it holds.  This kernel keeps the halo-slab idea and
//   * takes a 16 x 16 pixel patch per block (weight traffic per FLOP is half the 8 x 16 kernel's); 4 waves, each 64 pixels x all
//     128 output channels (8 accumulator tiles: 12 fragment reads per 16 MFMAs),
//   * rings the weight tiles through THREE LDS buffers with ONE barrier per step (details at the loop),
//   * keeps two slab buffers; the next channel chunk's slab is issued at tap 0, eight steps before its first use,
//   * computes everything lane-dependent ONCE (DMA source offsets, fragment addresses): a K step then costs a handful of scalar
//     adds — the first version spent ~110 VALU + ~90 SALU instructions per 16 MFMAs on addresses and was issue-bound,
//   * uses 64-byte K rows (32 channels): a chunk's slab is 23 KB, a weight tile 8 KB, 70 KB per block, so TWO blocks share a CU and
//     one's prologue (slab from HBM) and epilogue hide under the other's main loop.  The first cut of this pipeline (512 threads,
//     128-byte rows, 131 KB, one block per CU) exposed both — 25 % of a block's life — and measured 840-926 TFLOP/s on the SR layers
//     where this one gives 906-938.
// 64-byte rows put FOUR rows in a 256-byte bank row, so the slab pitch is 20 pixels (a multiple of 4) and the XOR key of a row is
// (column >> 2) & 3: the 16 rows one ds_read_b128 cycle serves (8 pixels of a patch row, 8 of the next) then cover all 16 bank quads.
constexpr int QH = 16, QW = 16;                     // pixel patch (QH * QW = 256)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

constexpr int H2_PITCH = 20;
constexpr int H2_SLAB_ROWS = 18 * H2_PITCH;                     // 360 rows of 64 bytes
constexpr int H2_SLAB_PIECES = (H2_SLAB_ROWS * 64 + 1023) / 1024;   // 23 DMA pieces (16 rows each)
constexpr int H2_SLAB_BUF = H2_SLAB_PIECES * 1024;              // 23552
constexpr int H2_WT_BYTES = BN * 64;                            // 8192
constexpr int H2_WT_BASE = 2 * H2_SLAB_BUF;                     // 47104
constexpr int H2_RGB_BASE = H2_WT_BASE + 3 * H2_WT_BYTES;       // 71680 (>= the 69632-byte epilogue stage): partial ToRGB sums [8][256 pixels] fp32 between the passes of a
constexpr int H2_LDS = H2_RGB_BASE + 8 * 256 * 4;               // work-group that walks several channel blocks (ConvArgs::cb_loop) — 79872; two work-groups a CU: 156 of 160 KB

// TR (p3d_conv3x3_torgb_f16 with y == null: the last block of a super-resolution head, whose activations only its ToRGB reads): the MFMA operands
// are SWAPPED — weights as A, pixels as B — so an accumulator tile is y^T: a lane holds 16 channels of ONE pixel ((r & 3) + 8 (r >> 2) + 4 fk of its
// 32-channel group).  After bias / activation / clamp and the fp16 rounding those registers ARE the B fragments of the ToRGB contraction over
// channels (the k <-> channel permutation is put into the ToRGB weights' A fragments), and its result tile has the pixel in the lane again: the
// epilogue needs no LDS image, no rendezvous and no store of y — 128 two-byte LDS stores, 16 LDS reads, 16 global stores and three block-wide
// barriers per thread become 16 MFMAs and a handful of read-modify-writes of the skip image.
// CBL (p3d_conv3x3_torgb_f16 at Co = 256): the work-group walks ConvArgs::cb_loop channel blocks of its patch, see `ncbi` below.  A build of its own so that the one-block
// kernels keep their register allocation (with the loop around the pipeline the compiler hoists the epilogue's address arithmetic above it: 241 -> 256 registers + spills).
template <bool TR, bool CBL = false>
__global__ void __launch_bounds__(256, 2) conv3x3_h2_f16_kernel(ConvArgs a)
{
    static_assert(!(TR && CBL), "the channel-block loop is a form of the LDS-image epilogue");
    // ONE __shared__ object on purpose: with two, hipcc drains vmcnt to 0 before the first ds_read of every step and the counted
    // waits below are moot (cdna_hip_programming.md, 'three .s-level traps' (a))
    __shared__ __attribute__((aligned(16))) char lds_b[H2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z;
    const int tiles_x = (a.W + QW - 1) / QW;
    int mt = blockIdx.x, cb = blockIdx.y;
    {
        const int nmt = gridDim.x, ncb = gridDim.y, L = blockIdx.x + blockIdx.y * nmt;
        if ((nmt & 7) == 0) { const int q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    }
    const int ty0 = mt / tiles_x, tx0 = mt - ty0 * tiles_x;
    const int oy0 = ty0 * QH, ox0 = tx0 * QW;
    // fused ToRGB of a layer with more than 128 channels: this work-group takes `ncbi` channel blocks one after the other (the whole pipeline per block) and carries
    // the ToRGB contraction's partial sums from pass to pass in LDS — bias, clamp and the read-modify-write of the skip image happen once, after the last
    const int ncbi = CBL ? a.cb_loop : 1;
    int co0 = cb * ncbi * BN;
    const char* const xin_b = (const char*)((const __half*)a.x + (int64_t)n * a.H * a.W * a.Ci);
    const char* const wgt_b = (const char*)((const __half*)a.w + (int64_t)n * a.w_img_stride);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int pos = lane & 3, prow = lane >> 2;                                // DMA lane: 16-byte position / row within its 16-row piece
    const int kpairs = a.Ci / 64;                                              // the loop body covers TWO 32-channel chunks (18 taps)

    unsigned woff[2];                                                          // weight pieces 2 * wave, 2 * wave + 1 (16 rows each) of the channel block at hand
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int row = (wave * 2 + p2) * 16 + prow;
        woff[p2] = (unsigned)((row * 9 * a.Ci + (pos ^ ((row >> 2) & 3)) * 8) * 2);
    }
    unsigned soff[6]; bool sok[6];                                             // slab pieces wave, wave + 4, ...  (6, or 5 for wave 3)
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int q = (wave + 4 * g) * 16 + prow;
        const int sy = q / H2_PITCH, sx = q - sy * H2_PITCH;
        const int iy = oy0 - 1 + sy, ix = ox0 - 1 + sx;
        sok[g] = (q < H2_SLAB_ROWS) & (sx < 18) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        soff[g] = (unsigned)(((iy * a.W + ix) * a.Ci + (pos ^ ((sx >> 2) & 3)) * 8) * 2);
    }
    const int nslab = (wave == 3) ? 5 : 6;
    auto stage_slab = [&](int cc, int buf) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g < nslab) {
                const char* src = sok[g] ? xin_b + soff[g] + cc * 64 : (const char*)a.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds_b + buf * H2_SLAB_BUF + (wave + 4 * g) * 1024), 16, 0, 0);
            }
        }
    };
    const char* wgt_cb = wgt_b + (int64_t)co0 * 9 * a.Ci * 2;                   // (uniform) the channel block's first weight row
    auto stage_w = [&](int cc, int t, int slot) {
        const char* base = wgt_cb + (t * a.Ci + cc * 32) * 2;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
            __builtin_amdgcn_global_load_lds((glb_ptr)(base + woff[p2]), (lds_ptr)(lds_b + H2_WT_BASE + slot * H2_WT_BYTES + (wave * 2 + p2) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][4];
    // fragment addresses (LDS byte offsets).  A: pixel (wave*4 + 2i + frow/16, frow%16) of the patch at tap (ty, tx) is slab row
    // (that pixel row + ty) * PITCH + column + tx; its 16-byte piece c sits at position c ^ key with the key taken from the slab
    // COLUMN.  Only tx changes the key: 3 x 2 lane constants; i, ty, the slab buffer and the ring slot are instruction immediates.
    const int frow = lane & 31, fk = lane >> 5;
    const int acol = frow & 15;
    int preA[3][2], preB[2];
#pragma unroll
    for (int tx2 = 0; tx2 < 3; ++tx2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            preA[tx2][kk] = ((wave * 4 + (frow >> 4)) * H2_PITCH + acol + tx2) * 64 + (((kk * 2 + fk) ^ (((acol + tx2) >> 2) & 3)) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) preB[kk] = H2_WT_BASE + frow * 64 + (((kk * 2 + fk) ^ ((frow >> 2) & 3)) << 4);

    f32x4 fa[2][2], fb[2][4];
    auto load_frags = [&](int buf, int t, int kk, f32x4* pa, f32x4* pb) {      // buf, t, kk are compile-time after unrolling
        // issued as opaque asm so that the compiler's own s_waitcnt (always lgkmcnt(0) here) stays out of the way: the waits are
        // the counted ones written next to the MFMAs below
        const int ty2 = t / 3, tx2 = t - ty2 * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[i]) : "v"(preA[tx2][kk]), "n"(buf * H2_SLAB_BUF + (i * 2 + ty2) * H2_PITCH * 64) : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pb[j]) : "v"(preB[kk]), "n"((t % 3) * H2_WT_BYTES + j * 32 * 64) : "memory");
    };

    // Software pipeline.  Fragments are double-buffered in registers (the reads of sub-step kk+1 fly under the MFMAs of kk).  ONE
    // block-wide rendezvous per step, at its last sub-step, after the step's last fragment reads have landed (lgkmcnt(0)): everyone
    // is then done with tile ks, so its ring slot takes the DMA of tile ks+3 at once, and the same point waits for tile ks+1 — issued
    // TWO steps earlier, while the group issued one step earlier (tile ks+2, plus a slab after tap 0) stays in flight under a counted
    // vmcnt.  Two full steps of flight out of three slots (PMC on a one-step version: 46 % of the wave cycles waiting at vmcnt(0) +
    // barrier).  Nine taps and three slots: the slot index is t % 3, a compile-time constant of the unrolled body.
    // TR: everything the epilogue reads from global memory apart from the skip image — the ToRGB weights (as fp16 A fragments, rows < 8), the layer's bias, the ToRGB's — is
    // requested HERE, ahead of the pipeline's first DMA, and parked in the 8 KB of LDS behind the weight ring that only the channel-block loop uses; the epilogue then starts
    // from LDS.  (Loaded where they were used, each under its null / row-bound branch and therefore waited for inside it, they were 12 dependent memory round trips of every
    // work-group's epilogue, and the read-modify-writes of the skip image 8 more: tools/sessions/gpu_round6_zi.sh.)
    constexpr int TRP_BW = H2_RGB_BASE, TRP_BIAS = TRP_BW + 16 * 9 * 16, TRP_RB = TRP_BIAS + 512;       // [k-step 8][k group 2][row 8 + a zero row] x 16 B; [128] floats; [8] floats
    f32x4 prm_w = {0.f, 0.f, 0.f, 0.f}; float prm_b = 0.f, prm_rb = 0.f;
    if constexpr (TR) {
        const int row = tid >> 5;
        prm_w = *(const f32x4*)(a.rgb_w + ((int64_t)n * a.rgb_co + min(row, a.rgb_co - 1)) * 128 + (tid & 31) * 4);
        if (row >= a.rgb_co) prm_w = f32x4{0.f, 0.f, 0.f, 0.f};
        prm_b = (a.bias ? a.bias : (const float*)a.zeros)[a.bias ? (tid & 127) : 0];
        const bool rbok = a.rgb_bias && (tid & 7) < a.rgb_co;
        prm_rb = (a.rgb_bias ? a.rgb_bias : (const float*)a.zeros)[rbok ? (tid & 7) : 0];
        if (!rbok) prm_rb = 0.f;
    }
#pragma nounroll
    for (int cbi = 0; cbi < ncbi; ++cbi, co0 += BN, wgt_cb += (int64_t)BN * 9 * a.Ci * 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    stage_slab(0, 0);
    stage_w(0, 0, 0);
    stage_w(0, 1, 1);
    stage_w(0, 2, 2);
    wait_vmcnt<2>();                                                            // slab 0, tiles 0 and 1 (tile 2 stays in flight)
    if constexpr (TR) {                                                         // (the parameter loads are older than the DMA pieces: they have landed)
        typedef _Float16 hp4 __attribute__((ext_vector_type(4)));
        const int row = tid >> 5, c0 = (tid & 31) * 4, off = c0 & 15;
        // element e of lane (row, k group) of k-step s is channel 16 s + (e & 3) + 8 (e >> 2) + 4 (k group): a run of four channels from 16 s + off is k group (off >> 2) & 1, e = 4 (off >> 3) ..
        *(hp4*)(lds_b + TRP_BW + (((c0 >> 4) * 2 + ((off >> 2) & 1)) * 9 + row) * 16 + (off >> 3) * 8) = hp4{(_Float16)prm_w[0], (_Float16)prm_w[1], (_Float16)prm_w[2], (_Float16)prm_w[3]};
        if (tid < 16) *(f32x4*)(lds_b + TRP_BW + (tid * 9 + 8) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid < 128) *(float*)(lds_b + TRP_BIAS + tid * 4) = prm_b;
        else if (tid < 136) *(float*)(lds_b + TRP_RB + (tid & 7) * 4) = prm_rb;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0, fa[0], fb[0]);
    for (int cp = 0; cp < kpairs; ++cp) {
        const bool more = cp + 1 < kpairs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = cp * 2 + h;
            const bool next_chunk = (h == 0) || more;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    if (kk == 0) { load_frags(h, t, 1, fa[nxt], fb[nxt]); asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); }
                    else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        // what may stay in flight: the group issued one step ago = tile ks+2 (if any) + the slab issued at tap 0
                        const bool t2 = (t < 7) || next_chunk;                  // tile ks+2 exists
                        if (t == 1 && next_chunk) { if (nslab == 6) wait_vmcnt<8>(); else wait_vmcnt<7>(); }
                        else if (t2) wait_vmcnt<2>();
                        else wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        if (t < 6) stage_w(cc, t + 3, t % 3);                   // tile ks+3 -> the slot tile ks just left
                        else if (next_chunk) stage_w(cc + 1, t - 6, t % 3);
                        if (t == 0 && next_chunk) stage_slab(cc + 1, h ^ 1);
                        if (t < 8) load_frags(h, t + 1, 0, fa[nxt], fb[nxt]);
                        else if (next_chunk) load_frags(h ^ 1, 0, 0, fa[nxt], fb[nxt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fb[cur][j]), __builtin_bit_cast(h8, fa[cur][i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[cur][i]), __builtin_bit_cast(h8, fb[cur][j]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if constexpr (TR) {
        // acc[i][j][r] = y^T: channel 32 j + (r & 3) + 8 (r >> 2) + 4 fk of pixel wave * 64 + i * 32 + frow.  (host: Co == 128, one channel block, no noise)
        // ToRGB weights as A fragments: lane (row o = frow, k group fk) of k-step (j, half) holds w[o][32 j + 16 half + (e & 3) + 8 (e >> 2) + 4 fk], e = 0 .. 7 —
        // the channels the pixel lanes' registers 8 half + e carry — rounded to fp16 like the reference's fp16 layer does
        h8 bw[8];                                                               // (parked in LDS by the kernel's first instructions; rows >= 8 read the zero row)
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) bw[sidx] = *(const h8*)(lds_b + TRP_BW + ((sidx * 2 + fk) * 9 + (frow < 8 ? frow : 8)) * 16);
        f32x16 rr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) rr[i][e] = 0.f;
        // the skip image's old values: requested before the activation arithmetic and the ToRGB MFMAs, added behind them (lanes without a pixel / channel read the image's first element)
        float* dstp[2][4]; bool dok[2][4]; float old[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + wave * 4 + 2 * i + (frow >> 4), ox = ox0 + (frow & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = r + 4 * fk;
                dok[i][r] = oy < a.H && ox < a.W && o < a.rgb_co;
                dstp[i][r] = dok[i][r] ? a.rgb_out + (((int64_t)n * a.rgb_co + o) * a.H + oy) * a.W + ox : a.rgb_out;
                old[i][r] = *dstp[i][r];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const f32x4*)(lds_b + TRP_BIAS + (j * 32 + 8 * q + 4 * fk) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[q * 4 + e] = b4[e];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                h8 yb[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + b[r];
                    if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                    v *= a.gain;
                    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                    yb[r >> 3][r & 7] = (_Float16)v;
                }
                rr[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[j * 2], yb[0], rr[i], 0, 0, 0);
                rr[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[j * 2 + 1], yb[1], rr[i], 0, 0, 0);
            }
        }
        // rr[i][r]: image channel (r & 3) + 8 (r >> 2) + 4 fk of pixel (wave * 4 + 2 i + (frow >> 4), frow & 15) of the patch: channels 0 .. 7 are r = 0 .. 3 of the two halves
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = rr[i][r] + *(const float*)(lds_b + TRP_RB + (r + 4 * fk) * 4);
                if (a.rgb_clamp >= 0.f) v = fminf(fmaxf(v, -a.rgb_clamp), a.rgb_clamp);
                if (dok[i][r]) *dstp[i][r] = old[i][r] + v;
            }
        return;
    }
    __syncthreads();                                                            // every wave is done with the slabs and tiles
    // (CBL: opaque copies of everything the epilogue derives its addresses from, so that they are computed HERE — with the loop around the pipeline the compiler otherwise
    // hoists ~70 loop-invariant epilogue values above it and spills them: 147 MB of scratch traffic per launch)
    int oy0e = oy0, ox0e = ox0, tid_e = tid;
    if constexpr (CBL) asm volatile("" : "+s"(oy0e), "+s"(ox0e), "+v"(tid_e));
    const int lane_e = CBL ? (tid_e & 63) : lane, wave_e = CBL ? __builtin_amdgcn_readfirstlane(tid_e >> 6) : wave;
    {   // ---- epilogue scope: tid / lane / wave / frow / fk below are the copies
    const int tid = tid_e, lane = lane_e, wave = wave_e, frow = lane_e & 31, fk = lane_e >> 5;

    // Epilogue through LDS (the slabs are free now): each lane holds ONE channel of 64 pixels, which as direct stores is 64 two-byte
    // writes per lane.  Instead the finished tile is laid out [256 pixels][128 channels] (pitch 136 halfs: the fk = 1 half-wave lands
    // 16 banks away) and leaves as 16-byte stores, 16 lanes per pixel = the pixel's whole 256-byte channel run.
    constexpr int OP = 136;
    __half* const ot = (__half*)lds_b;
    float* const nz = (float*)(ot + 256 * OP);                                   // the tile's noise: bytes 69632 .. 70656 < H2_LDS
    const float ns = a.noise ? a.noise_strength[0] : 0.f;
    // the four column blocks' biases in ONE batch of unconditional loads (an absent / out-of-range one reads the zeros page): as `cond ? load : 0` inside the loop each
    // was a branch with its own wait — four dependent memory round trips per pass (and eight more for the ToRGB weights below)
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + j * 32 + frow;
        const bool bok = a.bias && co < a.Co;
        bj[j] = (bok ? a.bias : (const float*)a.zeros)[bok ? co : 0];
    }
    if (a.noise) {
        const int oy = oy0e + (tid >> 4), ox = ox0e + (tid & 15);
        nz[tid] = (oy < a.H && ox < a.W) ? a.noise[(int64_t)oy * a.W + ox] * ns : 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = j * 32 + frow;
        const float b = bj[j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                float v = acc[i][j][r];
                if (a.noise) v += nz[p];
                v += b;
                if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                v *= a.gain;
                if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                ot[p * OP + cl] = __float2half(v);
            }
    }
    __syncthreads();
    if (a.y) {                                                                   // (null: p3d_conv3x3_torgb_f16 on a block whose x nobody reads — the tile only feeds the ToRGB below)
        __half* const yout = (__half*)a.y + (int64_t)n * a.H * a.W * a.Co;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid, p = idx >> 4, ch = idx & 15;
            const int oy = oy0e + (p >> 4), ox = ox0e + (p & 15), co = co0 + ch * 8;
            if (oy < a.H && ox < a.W && co < a.Co)
                *(f32x4*)(yout + ((int64_t)oy * a.W + ox) * a.Co + co) = *(const f32x4*)(ot + p * OP + ch * 8);
        }
    }
    if (a.rgb_out) {                                                             // (host: Co == 128, one channel block, no noise)
        // ToRGB of the finished tile on the matrix cores, as torgb_nhwc_kernel does it: A = the tile's pixels (rows of `ot`), B = the image's
        // modulated 1x1 weights rounded to fp16 (what the reference's fp16 layer multiplies by), 32 padded output columns.
        const int col = lane & 31, kg = lane >> 5;
        h8 bw[8];
        {
            const bool wok = col < a.rgb_co;                                    // (columns past the image's channels: the last row is read and discarded — no branch, sixteen loads in flight)
            const float* const wrow = a.rgb_w + ((int64_t)n * a.rgb_co + (wok ? col : a.rgb_co - 1)) * a.Co + co0 + kg * 8;
            f32x4 w0[8], w1[8];
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) { w0[sidx] = *(const f32x4*)(wrow + sidx * 16); w1[sidx] = *(const f32x4*)(wrow + sidx * 16 + 4); }
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx)
#pragma unroll
                for (int e = 0; e < 4; ++e) { bw[sidx][e] = wok ? (_Float16)w0[sidx][e] : (_Float16)0.f; bw[sidx][4 + e] = wok ? (_Float16)w1[sidx][e] : (_Float16)0.f; }
        }
        f32x16 rr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) rr[i][e] = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const h8 av = *(const h8*)(ot + (wave * 64 + i * 32 + col) * OP + sidx * 16 + kg * 8);
                rr[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bw[sidx], rr[i], 0, 0, 0);
            }
        }
        // hand-over buffer [rgb_co][256 pixels]: its own region behind the slabs and the ring, so that the sums of this channel block survive the next pass's staging
        float* const ro = (float*)(lds_b + H2_RGB_BASE);
        const bool last = cbi + 1 == ncbi;
        if (col < a.rgb_co) {
            const float b = (last && a.rgb_bias) ? a.rgb_bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* slot = ro + col * 256 + wave * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;      // (written and read back by the same lane)
                    float v = rr[i][e] + b;
                    if (cbi > 0) v += *slot;
                    if (last && a.rgb_clamp >= 0.f) v = fminf(fmaxf(v, -a.rgb_clamp), a.rgb_clamp);
                    *slot = v;
                }
        }
        if (last) {
            __syncthreads();
            const int p = tid, oy = oy0e + (p >> 4), ox = ox0e + (p & 15);
            if (oy < a.H && ox < a.W) {                                          // (rgb_co <= 8: every channel's old value requested at once, then added and stored)
                float* const d0 = a.rgb_out + ((int64_t)n * a.rgb_co * a.H + oy) * a.W + ox;
                const int64_t cs = (int64_t)a.H * a.W;
                float old[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) old[o] = d0[o < a.rgb_co ? o * cs : 0];
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < a.rgb_co) d0[o * cs] = old[o] + ro[o * 256 + p];
            }
        }
    }
    }   // ---- epilogue scope
    if (cbi + 1 < ncbi) {                                                       // the next pass stages over the tile image: every wave done with it, every store of y retired
        wait_vmcnt<0>();                                                        // (the counted waits of the pipeline count loads only)
        __syncthreads();
    }
    }   // channel blocks of this work-group
}

// ---- the same pipeline for the bf16x3 form of the fp32 layers, on activations that arrive split (XS) -----------------------------------
// conv3x3_halo_kernel<float, true, true> waits for its whole prefetch at every tap (two-stage pipeline); with the split gone from its loop it is
// LDS reads + MFMAs and that wait is what is left.  conv3x3_h2_f16_kernel's structure carries over byte for byte when an LDS row is taken to be
// the 16-channel HALF of a split K row — [16 x bf16 hi | 16 x bf16 lo], 64 bytes — because its two K sub-steps (16-byte positions 0, 1 and 2, 3)
// are then exactly the hi and the lo fragments: same slab (18 x 20 rows), same three-slot weight ring (8 KB tiles), same swizzle, same
// fragment addresses, 70 KB, two blocks per CU.  What differs: the DMA source of a row is two 32-byte runs of the 128-byte source row, a tap is
// 24 MFMAs (hi x hi, hi x lo, lo x hi) on 12 fragment reads, and the epilogue stores from registers.  Weights: p3d_modulate_weights(P3D_F32_BF16X3).
//
// TR (p3d_conv3x3_torgb_split: the LAST block of the tri-plane backbone, Co = 128, whose activations only its wide ToRGB reads): conv3x3_h2_f16_kernel<TR>'s idea on
// the bf16x3 pipeline.  The MFMA operands are swapped (weights as A, pixels as B), so an accumulator tile is y^T and a lane ends up with 16 channels of ONE pixel per
// 32-channel group; after noise / bias / activation / clamp those registers, split into (hi, lo) exactly as the stored form would have been, ARE the B fragments of the
// ToRGB contraction over channels (the k <-> channel permutation goes into the ToRGB weights' A fragments, gathered once per work-group into LDS in fragment order).
// The ToRGB result has the pixel in the lane again: bias, clamp, the x2-upsampled predecessor image (torgb_wide_split_kernel's taps in its order) and 16-byte stores of
// the [N][H][W][rgb_co] image follow from registers.  The layer's 134 MB of split activations (256^2 x 128 channels x batch 4) are neither written nor read back, and
// the ToRGB launch (92 us) is gone; the price is 144 more MFMAs per wave behind the K loop's 1 728.
struct WideRgbTail {
    const float* prev;     // [N][H/2][W/2][rgb_co] fp32 or null: the skip image of the block below, upsampled x2 here
    float fr[4][4];        // the upsampling filter as torgb_wide_split_kernel takes it (fr[ky][kx] = f[3 - ky][3 - kx] * 4)
};

template <bool TR>
__global__ void __launch_bounds__(256, 2) conv3x3_r2_bf16x3_kernel(ConvArgs a, WideRgbTail tail)
{
    // ONE __shared__ object on purpose: with two, hipcc drains vmcnt to 0 before the first ds_read of every step and the counted
    // waits below are moot (cdna_hip_programming.md, 'three .s-level traps' (a))
    __shared__ __attribute__((aligned(16))) char lds_b[H2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z;
    const int tiles_x = (a.W + QW - 1) / QW;
    int mt = blockIdx.x, cb = blockIdx.y;
    {
        const int nmt = gridDim.x, ncb = gridDim.y, L = blockIdx.x + blockIdx.y * nmt;
        if ((nmt & 7) == 0) { const int q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    }
    const int ty0 = mt / tiles_x, tx0 = mt - ty0 * tiles_x;
    const int oy0 = ty0 * QH, ox0 = tx0 * QW, co0 = cb * BN;
    const char* const xin_b = (const char*)((const float*)a.x + (int64_t)n * a.H * a.W * a.Ci);        // split rows: 128 bytes per pixel and 32 channels
    const char* const wgt_b = (const char*)((const float*)a.w + (int64_t)n * a.w_img_stride);
    // 16-byte piece q of a 64-byte LDS row [16 hi | 16 lo] inside its 128-byte source row [32 hi | 32 lo]: hi pieces at 16 q, lo pieces at 64 + 16 (q - 2)
    auto piece = [](int q) { return 16 * q + (q >= 2 ? 32 : 0); };
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int pos = lane & 3, prow = lane >> 2;                                // DMA lane: 16-byte position / row within its 16-row piece
    const int kpairs = a.Ci / 32;                                              // the loop body covers the TWO 16-channel halves of a 32-channel row (18 taps)

    unsigned woff[2];                                                          // weight pieces 2 * wave, 2 * wave + 1 (16 rows each)
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int row = (wave * 2 + p2) * 16 + prow;
        woff[p2] = (unsigned)((co0 + row) * 9 * a.Ci * 4 + piece(pos ^ ((row >> 2) & 3)));
    }
    unsigned soff[6]; bool sok[6];                                             // slab pieces wave, wave + 4, ...  (6, or 5 for wave 3)
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int q = (wave + 4 * g) * 16 + prow;
        const int sy = q / H2_PITCH, sx = q - sy * H2_PITCH;
        const int iy = oy0 - 1 + sy, ix = ox0 - 1 + sx;
        sok[g] = (q < H2_SLAB_ROWS) & (sx < 18) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        soff[g] = (unsigned)((iy * a.W + ix) * a.Ci * 4 + piece(pos ^ ((sx >> 2) & 3)));
    }
    const int nslab = (wave == 3) ? 5 : 6;
    auto stage_slab = [&](int cc, int buf) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g < nslab) {
                const char* src = sok[g] ? xin_b + soff[g] + (cc >> 1) * 128 + (cc & 1) * 32 : (const char*)a.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds_b + buf * H2_SLAB_BUF + (wave + 4 * g) * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_w = [&](int cc, int t, int slot) {
        const char* base = wgt_b + t * a.Ci * 4 + (cc >> 1) * 128 + (cc & 1) * 32;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
            __builtin_amdgcn_global_load_lds((glb_ptr)(base + woff[p2]), (lds_ptr)(lds_b + H2_WT_BASE + slot * H2_WT_BYTES + (wave * 2 + p2) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment addresses (LDS byte offsets).  A: pixel (wave*4 + 2i + frow/16, frow%16) of the patch at tap (ty, tx) is slab row
    // (that pixel row + ty) * PITCH + column + tx; its 16-byte piece c sits at position c ^ key with the key taken from the slab
    // COLUMN.  Only tx changes the key: 3 x 2 lane constants; i, ty, the slab buffer and the ring slot are instruction immediates.
    const int frow = lane & 31, fk = lane >> 5;
    const int acol = frow & 15;
    int preA[3][2], preB[2];
#pragma unroll
    for (int tx2 = 0; tx2 < 3; ++tx2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            preA[tx2][kk] = ((wave * 4 + (frow >> 4)) * H2_PITCH + acol + tx2) * 64 + (((kk * 2 + fk) ^ (((acol + tx2) >> 2) & 3)) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) preB[kk] = H2_WT_BASE + frow * 64 + (((kk * 2 + fk) ^ ((frow >> 2) & 3)) << 4);

    f32x4 hiA[2][2], hiB[2][4], loA[2], loB[4];                                 // hi fragments of two consecutive taps, lo fragments of the current one
    auto load_frags = [&](int buf, int t, int kk, f32x4* pa, f32x4* pb) {      // buf, t, kk are compile-time after unrolling
        // issued as opaque asm so that the compiler's own s_waitcnt (always lgkmcnt(0) here) stays out of the way: the waits are
        // the counted ones written next to the MFMAs below
        const int ty2 = t / 3, tx2 = t - ty2 * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[i]) : "v"(preA[tx2][kk]), "n"(buf * H2_SLAB_BUF + (i * 2 + ty2) * H2_PITCH * 64) : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pb[j]) : "v"(preB[kk]), "n"((t % 3) * H2_WT_BYTES + j * 32 * 64) : "memory");
    };

    // Software pipeline: conv3x3_h2_f16_kernel's (one rendezvous per tap, three-slot weight ring, counted vmcnt), with the fragment traffic of
    // three products: sub-step a runs the 8 hi x hi MFMAs while the tap's lo fragments load; sub-step b (after the rendezvous and the DMA issue)
    // runs hi x lo and lo x hi — 16 MFMAs — while the NEXT tap's hi fragments load into the other hi register set.
    stage_slab(0, 0);
    stage_w(0, 0, 0);
    stage_w(0, 1, 1);
    stage_w(0, 2, 2);
    wait_vmcnt<2>();                                                            // slab 0, tiles 0 and 1 (tile 2 stays in flight)
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0, hiA[0], hiB[0]);
    for (int cp = 0; cp < kpairs; ++cp) {
        const bool more = cp + 1 < kpairs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = cp * 2 + h;
            const bool next_chunk = (h == 0) || more;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cur = (h * 9 + t) & 1, nxt = cur ^ 1;                 // (compile-time after unrolling: 18 taps per iteration of cp)
                // ---- sub-step a
                load_frags(h, t, 1, loA, loB);
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, hiB[cur][j]), __builtin_bit_cast(bf8, hiA[cur][i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, hiA[cur][i]), __builtin_bit_cast(bf8, hiB[cur][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- sub-step b
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                {
                    const bool t2 = (t < 7) || next_chunk;                      // tile ks+2 exists
                    if (t == 1 && next_chunk) { if (nslab == 6) wait_vmcnt<8>(); else wait_vmcnt<7>(); }
                    else if (t2) wait_vmcnt<2>();
                    else wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                if (t < 6) stage_w(cc, t + 3, t % 3);                           // tile ks+3 -> the slot tile ks just left
                else if (next_chunk) stage_w(cc + 1, t - 6, t % 3);
                if (t == 0 && next_chunk) stage_slab(cc + 1, h ^ 1);
                if (t < 8) load_frags(h, t + 1, 0, hiA[nxt], hiB[nxt]);
                else if (next_chunk) load_frags(h ^ 1, 0, 0, hiA[nxt], hiB[nxt]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, loB[j]), __builtin_bit_cast(bf8, hiA[cur][i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, hiA[cur][i]), __builtin_bit_cast(bf8, loB[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, hiB[cur][j]), __builtin_bit_cast(bf8, loA[i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, loA[i]), __builtin_bit_cast(bf8, hiB[cur][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();                                                            // every wave is done with the slabs and tiles

    if constexpr (TR) {
        // acc[i][j][r] = y^T: channel 32 j + (r & 3) + 8 (r >> 2) + 4 fk of patch pixel p = wave * 64 + i * 32 + frow.  (host: Co == 128, one channel block, rgb_co in {32, 64, 96})
        // Nothing in the loops below waits on global memory: noise, both biases and the ToRGB weights are staged into LDS once, the predecessor image's 10 x 10 pixel patch
        // arrives by LDS-DMA one 32-channel block ahead of its use (out-of-image taps read the zeros page: fma(0, w, up) as torgb_wide_split_kernel has it).  The first form
        // of this epilogue loaded biases and taps where it used them, each under its null / bounds branch and therefore waited for inside it: 29 dependent memory round trips
        // per work-group, +43 us per launch.
        typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
        constexpr int TR_BS = 1024, TR_RBS = 1536, TR_WL = 2048, TR_PV = TR_WL + 3 * 8 * 2 * 64 * 16, TR_PVBUF = 13 * 1024;       // 51 200; two predecessor buffers of 13 DMA pieces
        static_assert(TR_PV + 2 * TR_PVBUF <= H2_LDS, "the TR epilogue's LDS image");
        float* const nz = (float*)lds_b;                                        // [256] the tile's noise * strength
        float* const bs = (float*)(lds_b + TR_BS);                              // [128] the layer's bias
        float* const rbs = (float*)(lds_b + TR_RBS);                            // [96]  the ToRGB's bias
        f32x4* const wl = (f32x4*)(lds_b + TR_WL);                              // ToRGB weights in fragment order: [((ot * 8 + sidx) * 2 + hl) * 64 + lane], 48 KB at rgb_co = 96
        const int not_ = a.rgb_co >> 5;
        const int PH2 = a.H >> 1, PW2 = a.W >> 1;
        const bool have_prev = tail.prev != nullptr;
        const float* const pn = have_prev ? tail.prev + (int64_t)n * PH2 * PW2 * a.rgb_co : nullptr;
        const int py0 = (oy0 >> 1) - 1, px0 = (ox0 >> 1) - 1;                   // the patch's taps: predecessor rows py0 .. py0 + 9, columns px0 .. px0 + 9
        auto stage_prev = [&](int ot, int buf) {                                // 100 pixels x 128 bytes: slot (pixel, 16-byte chunk) = 8 pixel + chunk
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int piece = wave + 4 * g;
                if (piece < 13) {
                    const int slot = piece * 64 + lane, pp = slot >> 3, py = pp / 10, pxx = pp - py * 10;
                    const int iy = py0 + py, ix = px0 + pxx;
                    const bool ok = (pp < 100) & (iy >= 0) & (iy < PH2) & (ix >= 0) & (ix < PW2);
                    const float* src = ok ? pn + ((int64_t)iy * PW2 + ix) * a.rgb_co + ot * 32 + (slot & 7) * 4 : (const float*)a.zeros;
                    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds_b + TR_PV + buf * TR_PVBUF + piece * 1024), 16, 0, 0);
                }
            }
        };
        if (have_prev) stage_prev(0, 0);
        {
            const float ns = a.noise ? a.noise_strength[0] : 0.f;
            const int oy = oy0 + (tid >> 4), ox = ox0 + (tid & 15);
            nz[tid] = (a.noise && oy < a.H && ox < a.W) ? a.noise[(int64_t)oy * a.W + ox] * ns : 0.f;
            if (tid < 128) bs[tid] = a.bias ? a.bias[tid] : 0.f;
            else if (tid < 128 + 96) rbs[tid - 128] = (a.rgb_bias && tid - 128 < a.rgb_co) ? a.rgb_bias[tid - 128] : 0.f;
            // A fragment of k-step sidx = (j, half), piece hl, lane (row o = 32 ot + frow, k group fk): element e = w[o][32 j + 16 half + (e & 3) + 8 (e >> 2) + 4 fk] — the
            // channels the pixel lanes' registers 8 half + e carry; in the weights' split rows ([32 hi | 32 lo] bf16 per 32 channels) that is two 8-byte runs 16 bytes apart
            const char* const wn = (const char*)(a.rgb_w + (int64_t)n * a.rgb_co * 128);
            const int nfr = not_ * 8 * 2 * 64;                                   // 1 024 fragment slots per 32 image channels: 4 per thread
            u32x2w p0[12], p1[12];
#pragma unroll
            for (int it = 0; it < 12; ++it) {                                   // (all 24 loads in flight; slots past the image's channel count re-read the last one and are not written)
                const int e = min(tid + 256 * it, nfr - 1);
                const int ln = e & 63, frag = e >> 6, hl = frag & 1, sidx = (frag >> 1) & 7, ot = frag >> 4;
                const char* src = wn + (int64_t)(ot * 32 + (ln & 31)) * 512 + (sidx >> 1) * 128 + hl * 64 + (sidx & 1) * 32 + (ln >> 5) * 8;
                p0[it] = *(const u32x2w*)src; p1[it] = *(const u32x2w*)(src + 16);
            }
#pragma unroll
            for (int it = 0; it < 12; ++it)
                if (it < not_ * 4) wl[tid + 256 * it] = __builtin_bit_cast(f32x4, (u32x4_t){p0[it][0], p0[it][1], p1[it][0], p1[it][1]});
        }
        wait_vmcnt<0>();                                                        // (the DMA pieces of predecessor block 0)
        __syncthreads();
        bf8 yh[2][4][2], yl[2][4][2];                                           // the finished activations as the stored split form would hold them: B fragments of k-steps (j, half)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4*)(bs + j * 32 + 8 * q + 4 * fk);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float nzv = nz[wave * 64 + i * 32 + frow];
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = acc[i][j][r];
                    if (a.noise) t += nzv;
                    t += b4[r >> 2][r & 3];
                    if (a.act == 1) t = fmaxf(t, 0.2f * t);
                    t *= a.gain;
                    if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
                    v[r] = t;
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    split_bf16x8(f32x4{v[8 * hf], v[8 * hf + 1], v[8 * hf + 2], v[8 * hf + 3]}, f32x4{v[8 * hf + 4], v[8 * hf + 5], v[8 * hf + 6], v[8 * hf + 7]}, yh[i][j][hf], yl[i][j][hf]);
            }
        }
        float* const yimg = a.rgb_out + (int64_t)n * a.H * a.W * a.rgb_co;
        // per tile i: this lane's pixel and its four predecessor taps (torgb_wide_split_kernel::store_tile: taps (jy, jx) in {0, 1}^2 of prev rows iyb + jy, columns ixb + jx with
        // weights fr[2 jy + ky0][2 jx + kx0], ky0 = (oy - 2) & 1, iyb = (oy - 2 + ky0) >> 1, x likewise; summed in that order)
        int toff[2]; float fw[2][2][2]; int64_t yoff[2]; bool inside[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave * 64 + i * 32 + frow, oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
            inside[i] = oy < a.H && ox < a.W;
            yoff[i] = ((int64_t)oy * a.W + ox) * a.rgb_co + 4 * fk;
            const int by = oy - 2, bx = ox - 2, ky0 = by & 1, kx0 = bx & 1, iyb = (by + ky0) >> 1, ixb = (bx + kx0) >> 1;
            toff[i] = ((iyb - py0) * 10 + (ixb - px0)) * 128 + fk * 16;         // LDS byte offset of tap (0, 0), channel 4 fk, inside a predecessor buffer; tap (jy, jx) adds (10 jy + jx) * 128
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx)
                    fw[i][jy][jx] = ky0 ? (kx0 ? tail.fr[2 * jy + 1][2 * jx + 1] : tail.fr[2 * jy + 1][2 * jx]) : (kx0 ? tail.fr[2 * jy][2 * jx + 1] : tail.fr[2 * jy][2 * jx]);
        }
#pragma unroll 1
        for (int ot = 0; ot < not_; ++ot) {
            if (have_prev && ot + 1 < not_) stage_prev(ot + 1, (ot + 1) & 1);   // lands under this block's 48 MFMAs per wave (its buffer's readers left at the last rendezvous)
            const char* const pvb = lds_b + TR_PV + (ot & 1) * TR_PVBUF;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 rr;
#pragma unroll
                for (int e = 0; e < 16; ++e) rr[e] = 0.f;
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx) {                          // torgb_wide_split_kernel's term order: x_hi w_hi, x_hi w_lo, x_lo w_hi
                    const bf8 wh = __builtin_bit_cast(bf8, wl[((ot * 8 + sidx) * 2 + 0) * 64 + lane]);
                    const bf8 wlo = __builtin_bit_cast(bf8, wl[((ot * 8 + sidx) * 2 + 1) * 64 + lane]);
                    rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yh[i][sidx >> 1][sidx & 1], rr, 0, 0, 0);
                    rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, yh[i][sidx >> 1][sidx & 1], rr, 0, 0, 0);
                    rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yl[i][sidx >> 1][sidx & 1], rr, 0, 0, 0);
                }
                // rr[r]: image channel 32 ot + (r & 3) + 8 (r >> 2) + 4 fk of this lane's pixel
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 rb = *(const f32x4*)(rbs + ot * 32 + 8 * q + 4 * fk);
                    f32x4 out;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = rr[4 * q + e] + rb[e];
                        if (a.rgb_clamp >= 0.f) t = fminf(fmaxf(t, -a.rgb_clamp), a.rgb_clamp);
                        out[e] = t;
                    }
                    if (have_prev) {
                        f32x4 up = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                            for (int jx = 0; jx < 2; ++jx) {
                                const f32x4 pv = *(const f32x4*)(pvb + toff[i] + (10 * jy + jx) * 128 + q * 32);
#pragma unroll
                                for (int e = 0; e < 4; ++e) up[e] = fmaf(pv[e], fw[i][jy][jx], up[e]);
                            }
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[e] = up[e] + out[e];
                    }
                    if (inside[i]) *(f32x4*)(yimg + yoff[i] + ot * 32 + 8 * q) = out;
                }
            }
            if (have_prev && ot + 1 < not_) {                                   // the next predecessor block has landed; every wave is done reading this one
                wait_vmcnt<0>();
                __syncthreads();
            }
        }
        return;
    }

    // Epilogue straight from the accumulators (an fp32 tile would not fit the LDS next to a second block): a lane holds ONE channel of 64 pixels;
    // the 32 lanes of a half-wave are the 32 consecutive channels of one pixel — 128 bytes per store instruction, or, for a split result
    // (ConvArgs::y_split), the pixel's [32 hi | 32 lo] K row as two 64-byte runs.
    float* const nz = (float*)lds_b;
    const float ns = a.noise ? a.noise_strength[0] : 0.f;
    if (a.noise) {
        const int oy = oy0 + (tid >> 4), ox = ox0 + (tid & 15);
        nz[tid] = (oy < a.H && ox < a.W) ? a.noise[(int64_t)oy * a.W + ox] * ns : 0.f;
        __syncthreads();
    }
    float* const yimg = (float*)a.y + (int64_t)n * a.H * a.W * a.Co;
    float bj[4];                                                                // the four column blocks' biases in one batch (see conv3x3_h2_f16_kernel's epilogue)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + j * 32 + frow;
        const bool bok = a.bias && co < a.Co;
        bj[j] = (bok ? a.bias : (const float*)a.zeros)[bok ? co : 0];
    }
    if (a.y_split == 1) {
        // Split result in 16-byte stores.  A lane holds ONE channel of four consecutive pixels in registers 4 q .. 4 q + 3; the four lanes of a quad hold four
        // consecutive channels.  Each finished value becomes one dword (bf16 hi | bf16 lo << 16), the quad transposes its 4 x 4 dwords in registers (two DPP
        // stages), so lane 4 m + t then holds pixel t's channels 4 m .. 4 m + 3; lanes 4 m + t and 4 (m ^ 1) + t trade halves (ds_swizzle, no memory) and the even
        // one stores the hi halves of eight channels, the odd one the lo halves: ONE 16-byte store per lane and pixel quad where the lane = channel layout needed
        // eight 2-byte ones (256 store instructions per lane and tile, 64 bytes each: ~8 % of a work-group's life on the address path its LDS-DMA shares).
        const int t = lane & 3, m = (lane & 31) >> 2;
        const bool odd1 = lane & 1, odd2 = lane & 2, oddm = m & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (co0 + j * 32 >= a.Co) continue;                                  // (whole 32-channel rows: uniform)
            const float b = bj[j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pbase = wave * 64 + i * 32 + 8 * q + 4 * fk;      // pixels pbase .. pbase + 3 of the 16 x 16 patch: one patch row
                    unsigned w[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e];
                        if (a.noise) v += nz[pbase + e];
                        v += b;
                        if (a.act == 1) v = fmaxf(v, 0.2f * v);
                        v *= a.gain;
                        if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                        const __bf16 hv = (__bf16)v;
                        const __bf16 lv = (__bf16)(v - (float)hv);
                        w[e] = (unsigned)__builtin_bit_cast(unsigned short, hv) | ((unsigned)__builtin_bit_cast(unsigned short, lv) << 16);
                    }
                    quad_transpose4(w, odd1, odd2);
                    // w[c] = (hi | lo << 16) of channel 4 m + c at pixel pbase + t
                    const unsigned h01 = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u), h23 = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
                    const unsigned l01 = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u), l23 = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
                    const unsigned s0 = oddm ? h01 : l01, s1 = oddm ? h23 : l23;
                    const unsigned r0 = (unsigned)__builtin_amdgcn_ds_swizzle((int)s0, (4 << 10) | 0x1f);               // lane ^ 4: the neighbouring quad (same pixel)
                    const unsigned r1 = (unsigned)__builtin_amdgcn_ds_swizzle((int)s1, (4 << 10) | 0x1f);
                    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
                    const u32x4q out = oddm ? u32x4q{r0, r1, l01, l23} : u32x4q{h01, h23, r0, r1};
                    const int pm = pbase + t;
                    const int oy = oy0 + (pm >> 4), ox = ox0 + (pm & 15);
                    if (oy < a.H && ox < a.W) {
                        __bf16* row = (__bf16*)(yimg + ((int64_t)oy * a.W + ox) * a.Co + co0 + j * 32);
                        *(u32x4q*)(row + (oddm ? 32 + 4 * (m - 1) : 4 * m)) = out;
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + j * 32 + frow;
        const float b = bj[j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
                if (oy >= a.H || ox >= a.W || co >= a.Co) continue;
                float v = acc[i][j][r];
                if (a.noise) v += nz[p];
                v += b;
                if (a.act == 1) v = fmaxf(v, 0.2f * v);
                v *= a.gain;
                if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                float* px = yimg + ((int64_t)oy * a.W + ox) * a.Co;
                if (a.y_split) {
                    __bf16* row = (__bf16*)(px + co0 + j * 32);
                    const __bf16 hv = (__bf16)v;
                    row[frow] = hv;
                    row[32 + frow] = (__bf16)(v - (float)hv);
                } else px[co] = v;
            }
    }
}

// ---- stride-2 transposed 3x3 conv on the two-blocks-per-CU structure (fp16) ------------------------------------------------
// The x2 layers' conv_transpose2d (conv2d_resample.py:114-127) is four dense sub-problems, one per output parity (4 / 2 / 2 / 1
// taps, offsets in {-1, 0}).  Each block takes one 16 x 16 patch of ONE parity class and runs conv3x3_h2_f16_kernel's pipeline over
// that class's tap list: the slab is shared by the class's taps (the generic kernel re-fetched a 16 KB activation tile per tap and
// had 4..16-step K loops), 70 KB of LDS, two blocks per CU.  Tap list and ring slot are run-time values here (1..4 taps do not
// divide the three slots), so a step pays two extra adds for its fragment bases; everything else is as above.
__global__ void __launch_bounds__(256, 2) convT_h2_f16_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds_b[H2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z / a.ncls;
    const ConvArgs::Cls& kc = a.cls[blockIdx.z - n * a.ncls];
    const int tiles_x = (kc.SW + QW - 1) / QW, tiles_y = (kc.SH + QH - 1) / QH;
    const int mt = blockIdx.x, cb = blockIdx.y;
    if (mt >= tiles_x * tiles_y) return;                                       // classes differ by one row / column of positions
    const int ty0 = mt / tiles_x, tx0 = mt - ty0 * tiles_x;
    const int oy0 = ty0 * QH, ox0 = tx0 * QW, co0 = cb * BN;                   // class-grid position of the patch = input pixel of tap (0, 0)
    const char* const xin_b = (const char*)((const __half*)a.x + (int64_t)n * a.H * a.W * a.Ci);
    const char* const wgt_b = (const char*)((const __half*)a.w + (int64_t)n * a.w_img_stride);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int pos = lane & 3, prow = lane >> 2;
    const int kchunks = a.Ci / 32, ntaps = kc.ntaps, ksteps = kchunks * ntaps;

    unsigned woff[2];
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int row = (wave * 2 + p2) * 16 + prow;
        woff[p2] = (unsigned)(((co0 + row) * 9 * a.Ci + (pos ^ ((row >> 2) & 3)) * 8) * 2);
    }
    unsigned soff[6]; bool sok[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int q = (wave + 4 * g) * 16 + prow;
        const int sy = q / H2_PITCH, sx = q - sy * H2_PITCH;
        const int iy = oy0 - 1 + sy, ix = ox0 - 1 + sx;
        sok[g] = (q < H2_SLAB_ROWS) & (sx < 18) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        soff[g] = (unsigned)(((iy * a.W + ix) * a.Ci + (pos ^ ((sx >> 2) & 3)) * 8) * 2);
    }
    const int nslab = (wave == 3) ? 5 : 6;
    auto stage_slab = [&](int cc, int buf) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g < nslab) {
                const char* src = sok[g] ? xin_b + soff[g] + cc * 64 : (const char*)a.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds_b + buf * H2_SLAB_BUF + (wave + 4 * g) * 1024), 16, 0, 0);
            }
        }
    };
    // the class's tap table goes into registers once: a scalar load per step would sit on lgkmcnt and drain the fragment pipeline
    int tap_w[4], tap_a[4], tap_x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tt = t < ntaps ? t : 0;
        tap_w[t] = kc.taps[tt].widx * a.Ci * 2;                                // byte offset of the tap inside a weight row
        tap_a[t] = (1 + kc.taps[tt].dy) * (H2_PITCH * 64);                     // slab row offset (dy in {-1, 0})
        tap_x[t] = 1 + kc.taps[tt].dx;                                         // slab column offset in {0, 1}
    }
    auto sel4 = [](const int (&v)[4], int t) { return t == 0 ? v[0] : (t == 1 ? v[1] : (t == 2 ? v[2] : v[3])); };
    auto stage_w = [&](int cc, int t, int slot) {                              // weight tile of (chunk cc, tap t of the class)
        const char* base = wgt_b + sel4(tap_w, t) + cc * 64;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
            __builtin_amdgcn_global_load_lds((glb_ptr)(base + woff[p2]), (lds_ptr)(lds_b + H2_WT_BASE + slot * H2_WT_BYTES + (wave * 2 + p2) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;
    const int acol = frow & 15;
    int preA[2][2], preB[2];                                                   // slab column offset 1 + dx in {0, 1}
#pragma unroll
    for (int tx2 = 0; tx2 < 2; ++tx2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            preA[tx2][kk] = ((wave * 4 + (frow >> 4)) * H2_PITCH + acol + tx2) * 64 + (((kk * 2 + fk) ^ (((acol + tx2) >> 2) & 3)) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) preB[kk] = H2_WT_BASE + frow * 64 + (((kk * 2 + fk) ^ ((frow >> 2) & 3)) << 4);

    f32x4 fa[2][2], fb[2][4];
    auto load_frags = [&](int cc, int t, int slot, int kk, f32x4* pa, f32x4* pb) {   // kk is compile-time; the rest is not
        const int abase = (sel4(tap_x, t) ? preA[1][kk] : preA[0][kk]) + (cc & 1) * H2_SLAB_BUF + sel4(tap_a, t);
        const int bbase = preB[kk] + slot * H2_WT_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pa[i]) : "v"(abase), "n"(i * 2 * H2_PITCH * 64) : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pb[j]) : "v"(bbase), "n"(j * 32 * 64) : "memory");
    };
    auto wait_group = [&](int nloads) {                                        // counted wait: let the newest `nloads` DMA instructions fly
        if (nloads >= 8) wait_vmcnt<8>(); else if (nloads == 7) wait_vmcnt<7>(); else if (nloads == 6) wait_vmcnt<6>();
        else if (nloads == 5) wait_vmcnt<5>(); else if (nloads >= 2) wait_vmcnt<2>(); else wait_vmcnt<0>();
    };

    // Slabs: chunk c lives in buffer c & 1.  Buffer c & 1 is free once the last step of chunk c-2 has its fragments in registers, so
    // slab c is issued right there — `ntaps` steps before its first read (the prefetch at the end of chunk c-1).  For ntaps >= 2
    // the counted waits in between force it to land; a one-tap class has only one step of lead and waits for everything.
    stage_slab(0, 0);
    if (kchunks > 1) stage_slab(1, 1);
    int issued = 0;                                                            // DMA instructions of the newest issue group
    {
        int c1 = 0, t1 = 0;
        for (int k = 0; k < 3 && k < ksteps; ++k) {
            stage_w(c1, t1, k);
            if (k == 2) issued = 2;
            if (++t1 == ntaps) { t1 = 0; ++c1; }
        }
    }
    wait_group(issued);                                                        // both slabs, tiles 0 and 1 (tile 2 may stay in flight)
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0, 0, fa[0], fb[0]);
    int slot = 0;                                                              // ring slot of the current step's weight tile (= s % 3)
    int c3 = 0, t3 = 0;                                                        // (chunk, tap) of step s + 3
    for (int k = 0; k < 3; ++k) if (++t3 == ntaps) { t3 = 0; ++c3; }
    int s = 0;
    for (int cc = 0; cc < kchunks; ++cc)
        for (int t = 0; t < ntaps; ++t, ++s) {
            int cn = cc, tn = t + 1;                                           // (chunk, tap) of step s + 1
            if (tn == ntaps) { tn = 0; ++cn; }
            const int slot_n = slot == 2 ? 0 : slot + 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int cur = kk, nxt = kk ^ 1;
                if (kk == 0) { load_frags(cc, t, slot, 1, fa[nxt], fb[nxt]); asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); }
                else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wait_group(ntaps == 1 ? 0 : issued);                       // tile s+1 (and the next chunk's slab) have landed
                    __builtin_amdgcn_s_barrier();
                    issued = 0;
                    if (s + 3 < ksteps) { stage_w(c3, t3, slot); issued = 2; }   // tile s+3 -> the slot tile s just left
                    if (t == ntaps - 1 && cc + 2 < kchunks) { stage_slab(cc + 2, cc & 1); issued += nslab; }
                    {   // unconditional (the last step re-reads its own tile): a conditional asm read makes every fragment register a phi
                        const bool last = s + 1 >= ksteps;
                        load_frags(last ? cc : cn, last ? t : tn, last ? slot : slot_n, 0, fa[nxt], fb[nxt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[cur][i]), __builtin_bit_cast(h8, fb[cur][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = slot_n;
            if (++t3 == ntaps) { t3 = 0; ++c3; }
        }
    __syncthreads();

    constexpr int OP = 136;
    __half* const ot = (__half*)lds_b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = j * 32 + frow;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                ot[p * OP + cl] = __float2half(acc[i][j][r] * a.gain);
            }
    }
    __syncthreads();
    __half* const yout = (__half*)a.y + (int64_t)n * a.OH * a.OW * a.Co;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + tid, p = idx >> 4, ch = idx & 15;
        const int si = oy0 + (p >> 4), sj = ox0 + (p & 15), co = co0 + ch * 8;
        const int oy = si * a.osy + kc.ooy, ox = sj * a.osx + kc.oox;
        if (si < kc.SH && sj < kc.SW && oy < a.OH && ox < a.OW && co < a.Co)
            *(f32x4*)(yout + ((int64_t)oy * a.OW + ox) * a.Co + co) = *(const f32x4*)(ot + p * OP + ch * 8);
    }
}

// ---- per-sample weight modulation + demodulation -> fp16, tap-major -------------------------------------------------
// one block per (co, n): wm[i, t] = w[co, i, t] * s[n, i]; d = rsqrt(sum wm^2 + 1e-8) (if demodulate); out[n][co][t][i]
// One block per (output channel, image).  The channel's weight row [Ci][KT] (18 KB for a 512-channel 3x3) is read once, coalesced, into
// LDS; the demodulation sum and the modulated copy — written in the tap-major order the convolution kernels stage, coalesced — both read
// it from there (the [i][t] -> [t][i] transposition is a stride-9 LDS read: conflict-free).  A variant with one block per channel serving
// every image from one staged row measured 1.8x SLOWER (512 blocks cannot keep enough stores in flight).
constexpr int MW_MAX_ROW = 512 * 9;                              // floats of LDS for the row; larger rows are read from global memory
template <class T>
__global__ void __launch_bounds__(256) modulate_weights_kernel(const float* __restrict__ w, const float* __restrict__ styles, T* __restrict__ out,
                                                               int Co, int Ci, int KT, int demodulate, float pre_scale, int oihw, int split)
{
    __shared__ float row[MW_MAX_ROW];
    __shared__ float red[4];
    const int co = blockIdx.x, n = blockIdx.y;
    const int total = Ci * KT;
    const float* wr = w + (int64_t)co * total;
    const float* s = styles + (int64_t)n * Ci;
    const bool staged = total <= MW_MAX_ROW;
    float sq = 0.f;
    // (unrolled by six: rolled, every iteration was its own memory round trip — eighteen in a row for a 512-channel 3x3 layer)
    if (staged) {                                                    // row[e] = w * pre_scale * s: the value both passes need
#pragma unroll 6
        for (int e = threadIdx.x; e < total; e += 256) {
            const float v = wr[e] * pre_scale * s[e / KT];
            row[e] = v;
            sq = fmaf(v, v, sq);
        }
    } else {
#pragma unroll 6
        for (int e = threadIdx.x; e < total; e += 256) {
            const float v = wr[e] * pre_scale * s[e / KT];
            sq = fmaf(v, v, sq);
        }
    }
    float d = 1.f;
    if (demodulate) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        d = rsqrtf(red[0] + red[1] + red[2] + red[3] + 1e-8f);
    } else
        __syncthreads();
    auto val = [&](int i, int t) { return staged ? row[i * KT + t] * d : wr[i * KT + t] * pre_scale * s[i] * d; };
    if (split) {                                                     // bf16x3: K rows of [32 x hi | 32 x lo] bf16 (as many bytes as 32 floats)
        __bf16* ob = (__bf16*)out + ((int64_t)n * Co + co) * total * 2;
        for (int e = threadIdx.x * 8; e < total; e += 256 * 8) {     // eight channels of one tap per thread: two 16-byte stores (Ci % 32 == 0)
            const int t = e / Ci, i = e - t * Ci;
            bf8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = val(i + k, t);
                const __bf16 h = (__bf16)v;
                hi[k] = h; lo[k] = (__bf16)(v - (float)h);
            }
            const int64_t r = ((int64_t)t * Ci + (i & ~31)) * 2 + (i & 31);
            *(bf8*)(ob + r) = hi;
            *(bf8*)(ob + r + 32) = lo;
        }
        return;
    }
    T* o = out + ((int64_t)n * Co + co) * total;
    if (oihw) {                                                      // keep the source order [i][t] (GEMM route of the small layers)
        for (int e = threadIdx.x; e < total; e += 256) st(o + e, val(e / KT, e % KT));
        return;
    }
    constexpr int VEC = 16 / (int)sizeof(T);                         // the OUTPUT order [t][i] in 16-byte pieces when the row allows it
    if (Ci % VEC == 0 && ((uintptr_t)o & 15u) == 0) {
        for (int e = threadIdx.x * VEC; e < total; e += 256 * VEC) {
            const int t = e / Ci, i = e - t * Ci;
            T pk[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) st(pk + k, val(i + k, t));
            *(f32x4*)(o + e) = *(const f32x4*)pk;
        }
        return;
    }
    for (int e = threadIdx.x; e < total; e += 256) {
        const int t = e / Ci, i = e - t * Ci;
        st(o + e, val(i, t));
    }
}

// ---- 1x1 ToRGB on fp16 NHWC activations -> fp32 NCHW image (networks_stylegan2.py:355-359) ----------------------------
// y[n, o, p] = clamp( sum_i x[n, p, i] * w[o, i] * s[n, i] + b[o] ): a skinny GEMM (Co <= 8) whose cost is reading x once.
// One wave = 32 pixels per step on v_mfma_f32_32x32x16_f16: A = the pixels' channel vectors loaded straight from
// global memory in fragment layout (16 B per lane per K step, all K steps of a tile in flight together), B = the
// per-image modulated weights padded to 32 columns and held in registers for the whole kernel.  Output channel o is
// column o of the accumulator: lanes o < Co store four float4 (4 consecutive pixels each) into the NCHW image.
template <int KSTEPS>
__global__ void __launch_bounds__(256) torgb_nhwc_kernel(const __half* __restrict__ x, const float* __restrict__ w, const float* __restrict__ styles,
                                                         const float* __restrict__ bias, float* __restrict__ y, int HW, int Co, float clamp, int accumulate)
{
    constexpr int Ci = KSTEPS * 16;
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, col = lane & 31, kg = lane >> 5;
    h8 bw[KSTEPS];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = k * 16 + kg * 8 + e;
            bw[k][e] = (col < Co) ? (_Float16)(w[col * Ci + ci] * styles[(int64_t)n * Ci + ci]) : (_Float16)0.f;
        }
    const float b = (bias && col < Co) ? bias[col] : 0.f;
    const __half* xi = x + (int64_t)n * HW * Ci;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // A wave takes several 32-pixel tiles (the launch is sized to the chip: two waves per SIMD), and the NEXT tile's 16 loads are in flight while
    // this one is multiplied and stored — with one tile per wave, as first written, a wave's life was the weight set-up above plus one exposed
    // round trip to memory (68 us for 134 MB on the 256-channel 256^2 layer of the SR heads).
    const int stride = nwaves * 32;
    auto load_tile = [&](h8 (&fa)[KSTEPS], int p0) {
        const int p = min(p0 + col, HW - 1);                           // fragment row = pixel
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) fa[k] = *(const h8*)(xi + (int64_t)p * Ci + k * 16 + kg * 8);
    };
    auto finish_tile = [&](const h8 (&fa)[KSTEPS], int p0) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k], bw[k], acc, 0, 0, 0);
        if (col < Co) {
            float* dst = y + ((int64_t)n * Co + col) * HW + p0 + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                              // rows 8q + 4kg + {0..3}
                const int pp = p0 + 8 * q + 4 * kg;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[q * 4 + e] + b;
                    if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
                    v[e] = t;
                }
                if (pp + 3 < HW) {
                    f32x4* d4 = (f32x4*)(dst + 8 * q);
                    if (accumulate) { const f32x4 o = *d4; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                    *d4 = v;
                } else {
                    for (int e = 0; e < 4; ++e) if (pp + e < HW) dst[8 * q + e] = accumulate ? dst[8 * q + e] + v[e] : v[e];
                }
            }
        }
    };
    h8 fa0[KSTEPS], fa1[KSTEPS];
    int p0 = wave_global * 32;
    if (p0 < HW) load_tile(fa0, p0);
    while (p0 < HW) {
        if (p0 + stride < HW) load_tile(fa1, p0 + stride);
        finish_tile(fa0, p0);
        p0 += stride;
        if (p0 >= HW) break;
        if (p0 + stride < HW) load_tile(fa0, p0 + stride);
        finish_tile(fa1, p0);
        p0 += stride;
    }
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_modulate_weights(const float* weight, const float* styles, void* out, int dtype, int32_t n_img, int32_t co, int32_t ci,
                                    int32_t taps, int32_t demodulate, float pre_scale, int32_t oihw_order, p3d_stream_t stream)
{
    P3D_REQUIRE(weight && styles && out, "modulate_weights: null pointer");
    P3D_REQUIRE(n_img >= 1 && co >= 1 && ci >= 1 && taps >= 1, "modulate_weights: bad sizes");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32 || dtype == P3D_F32_BF16X3, "modulate_weights: dtype must be fp16, fp32 or fp32-as-bf16x3");
    P3D_REQUIRE(dtype != P3D_F32_BF16X3 || (ci % 32 == 0 && !oihw_order), "modulate_weights: the bf16x3 layout needs Ci % 32 == 0 and tap-major order");
    if (dtype == P3D_F16) hipLaunchKernelGGL(modulate_weights_kernel<__half>, dim3(co, n_img), dim3(256), 0, (hipStream_t)stream, weight, styles, (__half*)out, co, ci, taps, demodulate, pre_scale, oihw_order, 0);
    else                  hipLaunchKernelGGL(modulate_weights_kernel<float>,  dim3(co, n_img), dim3(256), 0, (hipStream_t)stream, weight, styles, (float*)out, co, ci, taps, demodulate, pre_scale, oihw_order, dtype == P3D_F32_BF16X3);
    count_launch(FAM_CONV);
    return check_launch("modulate_weights");
}

// Fold the batch into the GEMM rows when the weights are shared and one image cannot fill a 128-row tile (training-mode and
// plain Conv2dLayer calls on the low-resolution blocks): N x H x W rows instead of N launches' worth of mostly empty tiles.
static int fold_batch(const ConvArgs& a, int dtype)
{
    const int64_t mi = (int64_t)a.cls[0].SH * a.cls[0].SW;
    const int64_t in_bytes = (int64_t)a.N * a.H * a.W * a.Ci * (dtype == P3D_F16 ? 2 : 4);
    return a.ncls == 1 && a.w_img_stride == 0 && a.N > 1 && mi < 4 * BM && in_bytes < (1ll << 31) && mi * a.N < (1ll << 24);
}

// Split-K plan of the generic kernel for a launch of `blocks` work-groups whose longest K loop has `nsteps` steps: 1 when the
// launch already fills the chip or the loop is short; otherwise enough splits for ~2 work-groups per CU with >= 4 steps each.
static int splitk_plan(int64_t blocks, int nsteps)
{
    // tuning switches of tests/gpu_probe_splitk.sh (defaults = the values the sweep kept): work-groups per CU aimed at, fewest K steps per split
    static const int per_cu = [] { const char* e = getenv("P3D_SPLITK_PER_CU"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 8 ? v : 2; }();
    static const int min_steps = [] { const char* e = getenv("P3D_SPLITK_MIN_STEPS"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
    if (blocks >= kNumCU || nsteps < 8) return 1;
    int64_t want = (per_cu * kNumCU + blocks - 1) / blocks;
    if (want > nsteps / min_steps) want = nsteps / min_steps;
    if (want < 2) return 1;
    const int per = (int)((nsteps + want - 1) / want);
    return (nsteps + per - 1) / per;
}

static void conv_launch_shape(const ConvArgs& a, int* M, int* z, int* nsteps)
{
    int m = 0, taps = 0;
    for (int c = 0; c < a.ncls; ++c) {
        m = a.cls[c].SH * a.cls[c].SW > m ? a.cls[c].SH * a.cls[c].SW : m;
        taps = a.cls[c].ntaps > taps ? a.cls[c].ntaps : taps;
    }
    *M = a.fold ? m * a.N : m;
    *z = a.fold ? 1 : a.N * a.ncls;
    *nsteps = taps;
}

static int launch_conv(ConvArgs& a, int dtype, hipStream_t s, void* workspace, int64_t workspace_bytes, int64_t* query, bool x_split = false)
{
    int M, z, taps;
    conv_launch_shape(a, &M, &z, &taps);
    if (M <= 0) return P3D_OK;
    const int gx = (M + BM - 1) / BM, gy = (a.Co + BN - 1) / BN;
    const int nsteps = taps * (a.Ci / (dtype == P3D_F16 ? 64 : 32));
    a.ksplit = 1; a.partial = nullptr;
    const int want = splitk_plan((int64_t)gx * gy * z, nsteps);
    const int64_t need = want > 1 ? (int64_t)want * z * gx * BM * gy * BN * 4 : 0;
    if (query) { *query = need; return P3D_OK; }
    if (want > 1 && workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15u) == 0) { a.ksplit = want; a.partial = (float*)workspace; }
    dim3 grid(gx, gy, z * a.ksplit);
    static const bool no_co64 = getenv("P3D_CONV_NO_CO64") != nullptr;          // (A/B switch of the measurement scripts)
    const bool co64 = !no_co64 && a.Co <= 64;
    static const bool no_cls_major = [] { const char* e = getenv("P3D_CONV_CLASS_MAJOR"); return e && atoi(e) == 0; }();      // (A/B switch)
    a.cls_major = (a.ncls > 1 && !a.fold && !no_cls_major) ? 1 : 0;
    if (a.iscale) {                                                             // the table of conv2d_nhwc_kernel<.., ISC>: the scale rows of every image a 128-row tile touches
        if (dtype != P3D_F32_BF16X3 || x_split) return fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: the input scale is implemented for bf16x3 on plain fp32 activations");
        int worst = 1;
        if (a.fold) {
            const int MI = a.cls[0].SH * a.cls[0].SW;
            for (int tile = 0; tile < gx; ++tile) {
                const int m0 = tile * BM, m1 = (m0 + BM < M ? m0 + BM : M) - 1;
                int i1 = m1 / MI; if (i1 > a.N - 1) i1 = a.N - 1;
                const int cnt = i1 - m0 / MI + 1;
                if (cnt > worst) worst = cnt;
            }
        }
        if ((int64_t)worst * a.Ci > kIscaleFloats) return fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: %d images x %d channels of input scales exceed the kernel's table", worst, a.Ci);
    }
    if (dtype == P3D_F16 && co64)     hipLaunchKernelGGL((conv2d_nhwc_kernel<__half, false, false, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32 && co64) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, false, false, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32_BF16X6 && co64) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, false, false, true, false, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32_BF16X6) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, false, false, false, false, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F16)        hipLaunchKernelGGL(conv2d_nhwc_kernel<__half>, grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32_BF16X3 && x_split) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, true, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32_BF16X3 && a.iscale) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, true, false, false, true>), grid, dim3(256), 0, s, a);
    else if (dtype == P3D_F32_BF16X3) hipLaunchKernelGGL((conv2d_nhwc_kernel<float, true>), grid, dim3(256), 0, s, a);
    else                              hipLaunchKernelGGL(conv2d_nhwc_kernel<float>, grid, dim3(256), 0, s, a);
    count_launch(FAM_CONV);
    int rc = check_launch("conv2d_nhwc");
    if (rc != P3D_OK || a.ksplit == 1) return rc;
    const int64_t total = (int64_t)z * gx * BM * ((a.Co + 3) / 4);
    const int blocks = (int)((total + 255) / 256 < 8 * kNumCU ? (total + 255) / 256 : 8 * kNumCU);
    if (dtype == P3D_F16) hipLaunchKernelGGL(splitk_epilogue_kernel<__half>, dim3(blocks), dim3(256), 0, s, a, gx * BM, gy * BN, z);   // (P3D_F32_BF16X3: fp32 tensors)
    else                  hipLaunchKernelGGL(splitk_epilogue_kernel<float>, dim3(blocks), dim3(256), 0, s, a, gx * BM, gy * BN, z);
    count_launch(FAM_CONV);
    return check_launch("conv2d_nhwc split-K epilogue");
}

extern "C" int p3d_conv2d_nhwc(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                               const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                               int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, p3d_stream_t stream)
{
    return p3d::conv2d_nhwc_run(x, w, y, dtype, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain, clamp,
                                0, 0, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int p3d_conv2d_nhwc_ws(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                                  const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                                  int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, void* workspace, int64_t workspace_bytes,
                                  p3d_stream_t stream)
{
    return p3d::conv2d_nhwc_run(x, w, y, dtype, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain, clamp,
                                0, 0, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int p3d_conv3x3_torgb_f16(const void* x, const void* w, void* y, const float* bias, const void* zeros128, const float* rgb_w, const float* rgb_bias,
                                     float* rgb_out, int32_t rgb_co, float rgb_clamp, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                                     int64_t w_img_stride, int32_t act, float gain, float clamp, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && w && zeros128 && rgb_w && rgb_out, "conv3x3_torgb_f16: null pointer");      // y may be null: the layer's activations have no other consumer
    P3D_REQUIRE(rgb_co >= 1 && rgb_co <= 8, "conv3x3_torgb_f16: 1 .. 8 image channels");
    P3D_REQUIRE(act == 0 || act == 1, "conv3x3_torgb_f16: act must be 0 (linear) or 1 (lrelu)");
    if ((co != BN && co != 2 * BN) || ci % 64 != 0 || h < 32 || wdt < 32 || ((uintptr_t)y & 15u) || ((uintptr_t)rgb_w & 15u) || (bias && ((uintptr_t)bias & 15u)))
        return fail(P3D_ERR_UNSUPPORTED, "conv3x3_torgb_f16: needs Co = 128 or 256, Ci %% 64 = 0, an image of 32 x 32 or more (got %d, %d, %d x %d)", co, ci, h, wdt);
    P3D_REQUIRE((((uintptr_t)x) & 15u) == 0 && (((uintptr_t)w) & 15u) == 0 && (((uintptr_t)zeros128) & 15u) == 0, "conv3x3_torgb_f16: x, w and zeros128 must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.zeros = zeros128;
    a.N = n_img; a.H = h; a.W = wdt; a.Ci = ci; a.Co = co; a.KT = 9; a.w_img_stride = w_img_stride;
    a.act = act; a.gain = gain; a.clamp = clamp; a.isy = a.isx = 1; a.OH = h; a.OW = wdt; a.osy = a.osx = 1; a.ncls = 1; a.ksplit = 1;
    a.cls[0].SH = h; a.cls[0].SW = wdt; a.cls[0].ntaps = 9;
    for (int t = 0; t < 9; ++t) a.cls[0].taps[t] = ConvTap{t / 3 - 1, t % 3 - 1, t};
    a.rgb_w = rgb_w; a.rgb_bias = rgb_bias; a.rgb_out = rgb_out; a.rgb_co = rgb_co; a.rgb_clamp = rgb_clamp;
    dim3 grid(((h + QH - 1) / QH) * ((wdt + QW - 1) / QW), 1, n_img);
    static const bool no_tr = [] { const char* d = getenv("P3D_TORGB_NO_TR"); return d && atoi(d) != 0; }();      // A/B switch: the LDS-image epilogue without the store
    a.cb_loop = co / BN;                                                        // Co = 256: each work-group walks both channel blocks of its patch (the contraction runs over all of them)
    if (a.cb_loop > 1)     hipLaunchKernelGGL((conv3x3_h2_f16_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (!y && !no_tr) hipLaunchKernelGGL(conv3x3_h2_f16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else                   hipLaunchKernelGGL(conv3x3_h2_f16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    count_launch(FAM_CONV);
    return check_launch("conv3x3_torgb_f16");
}

extern "C" int p3d_conv3x3_torgb_split(const void* x_split, const void* w_split, const float* bias, const float* noise, const float* noise_strength, const void* zeros128,
                                       const void* rgb_wmod_split, const float* rgb_bias, float* img_nhwc, const float* prev_nhwc, const float* f4x4_host,
                                       int32_t rgb_co, float rgb_clamp, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                                       int32_t act, float gain, float clamp, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x_split && w_split && zeros128 && rgb_wmod_split && img_nhwc, "conv3x3_torgb_split: null pointer");
    P3D_REQUIRE(!prev_nhwc || f4x4_host, "conv3x3_torgb_split: the skip image needs its upsampling filter");
    P3D_REQUIRE(!noise || noise_strength, "conv3x3_torgb_split: noise without its strength");
    P3D_REQUIRE(act == 0 || act == 1, "conv3x3_torgb_split: act must be 0 (linear) or 1 (lrelu)");
    P3D_REQUIRE(n_img >= 1 && n_img < 65536 && h >= 1 && wdt >= 1, "conv3x3_torgb_split: bad sizes");
    const int64_t blocks = (int64_t)((h + QH - 1) / QH) * ((wdt + QW - 1) / QW) * n_img;
    if (co != BN || ci % 32 != 0 || h < 32 || wdt < 32 || rgb_co % 32 != 0 || rgb_co < 32 || rgb_co > 96 || (prev_nhwc && ((h | wdt) & 1)) || blocks < 192)
        return fail(P3D_ERR_UNSUPPORTED, "conv3x3_torgb_split: needs Co = 128, Ci %% 32 = 0, rgb_co in {32, 64, 96}, an even image of 32 x 32 or more and >= 192 patches (got %d, %d, %d, %d x %d x %d)",
                    co, ci, rgb_co, n_img, h, wdt);
    P3D_REQUIRE(((((uintptr_t)x_split) | ((uintptr_t)w_split) | ((uintptr_t)zeros128) | ((uintptr_t)rgb_wmod_split) | ((uintptr_t)img_nhwc) | ((uintptr_t)prev_nhwc) | ((uintptr_t)bias)
                  | ((uintptr_t)rgb_bias)) & 15u) == 0, "conv3x3_torgb_split: pointers must be 16-byte aligned");
    ConvArgs a{};
    a.x = x_split; a.w = w_split; a.bias = bias; a.noise = noise; a.noise_strength = noise_strength; a.zeros = zeros128;
    a.N = n_img; a.H = h; a.W = wdt; a.Ci = ci; a.Co = co; a.KT = 9; a.w_img_stride = w_img_stride;
    a.act = act; a.gain = gain; a.clamp = clamp; a.isy = a.isx = 1; a.OH = h; a.OW = wdt; a.osy = a.osx = 1; a.ncls = 1; a.ksplit = 1;
    a.cls[0].SH = h; a.cls[0].SW = wdt; a.cls[0].ntaps = 9;
    for (int t = 0; t < 9; ++t) a.cls[0].taps[t] = ConvTap{t / 3 - 1, t % 3 - 1, t};
    a.rgb_w = (const float*)rgb_wmod_split; a.rgb_bias = rgb_bias; a.rgb_out = img_nhwc; a.rgb_co = rgb_co; a.rgb_clamp = rgb_clamp;
    WideRgbTail tail{};
    tail.prev = prev_nhwc;
    if (prev_nhwc)
        for (int ky = 0; ky < 4; ++ky)                                            // (f4x4_host: sixteen floats in HOST memory, row-major; as p3d_torgb_wide_split builds them)
            for (int kx = 0; kx < 4; ++kx) tail.fr[ky][kx] = f4x4_host[(3 - kx) + (3 - ky) * 4] * 4.f;
    dim3 grid(((h + QH - 1) / QH) * ((wdt + QW - 1) / QW), 1, n_img);
    hipLaunchKernelGGL(conv3x3_r2_bf16x3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, tail);
    count_launch(FAM_CONV);
    return check_launch("conv3x3_torgb_split");
}

extern "C" int p3d_conv2d_nhwc_bf16x3_io(const void* x, const void* w, void* y, const float* bias, const float* noise, const float* noise_strength,
                                         const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                                         int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t x_split, int32_t y_split,
                                         void* workspace, int64_t workspace_bytes, p3d_stream_t stream)
{
    return p3d::conv2d_nhwc_run_io(x, w, y, P3D_F32_BF16X3, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain,
                                   clamp, 0, 0, workspace, workspace_bytes, nullptr, nullptr, x_split, y_split, stream);
}

// The route p3d_conv2d_nhwc_bf16x3_io would take, without launching: returns 1 when a split result (y_split) would be granted for these sizes,
// 0 when the caller has to ask for a plain tensor, < 0 on unusable arguments; *workspace_bytes = the split-K scratch of THAT route (0: none).
extern "C" int p3d_conv2d_nhwc_bf16x3_io_plan(int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride, int32_t kernel_size,
                                              int32_t resample, int32_t x_split, int32_t want_y_split, int64_t* workspace_bytes)
{
    P3D_REQUIRE(workspace_bytes, "conv2d_nhwc_bf16x3_io_plan: null pointer");
    *workspace_bytes = 0;
    for (int ys = (want_y_split && co % 32 == 0) ? 1 : 0; ys >= 0; --ys) {
        int64_t bytes = 0;
        const int rc = p3d::conv2d_nhwc_run_io(nullptr, nullptr, nullptr, P3D_F32_BF16X3, nullptr, nullptr, nullptr, nullptr, n_img, h, wdt, ci, co, w_img_stride, kernel_size,
                                               resample, 0, 1.f, -1.f, 0, 0, nullptr, 0, &bytes, nullptr, x_split, ys, nullptr);
        if (rc == P3D_OK) { *workspace_bytes = bytes; return ys; }
        if (rc != P3D_ERR_UNSUPPORTED || ys == 0) return rc < 0 ? rc : -rc;
    }
    return 0;
}

extern "C" int p3d_conv2d_nhwc_scaled(const void* x, const void* w, void* y, int dtype, const float* out_scale, const float* bias, const float* noise,
                                      const float* noise_strength, const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                                      int64_t w_img_stride, int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, void* workspace,
                                      int64_t workspace_bytes, p3d_stream_t stream)
{
    return p3d::conv2d_nhwc_run(x, w, y, dtype, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain, clamp,
                                0, 0, workspace, workspace_bytes, nullptr, out_scale, stream);
}

extern "C" int p3d_conv2d_nhwc_scaled_in(const void* x, const void* w, void* y, int dtype, const float* in_scale, const float* out_scale, const float* bias,
                                         const float* noise, const float* noise_strength, const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci,
                                         int32_t co, int64_t w_img_stride, int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp,
                                         void* workspace, int64_t workspace_bytes, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(in_scale && out_scale, "conv2d_nhwc_scaled_in: both scales are required (the shared-weight form of the modulated convolution)");
    if (dtype != P3D_F32_BF16X3) return fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc_scaled_in: implemented for dtype P3D_F32_BF16X3");
    P3D_REQUIRE((((uintptr_t)in_scale) & 15u) == 0, "conv2d_nhwc_scaled_in: in_scale must be 16-byte aligned");
    tl_in_scale = in_scale;
    const int rc = conv2d_nhwc_run(x, w, y, dtype, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain, clamp,
                                   0, 0, workspace, workspace_bytes, nullptr, out_scale, stream);
    tl_in_scale = nullptr;
    return rc;
}

extern "C" int64_t p3d_conv2d_nhwc_workspace(int dtype, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride, int32_t kernel_size,
                                             int32_t resample)
{
    int64_t bytes = 0;
    const int rc = p3d::conv2d_nhwc_run(nullptr, nullptr, nullptr, dtype, nullptr, nullptr, nullptr, nullptr, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, 0, 1.f, -1.f,
                                        0, 0, nullptr, 0, &bytes, nullptr, nullptr);
    return rc == P3D_OK ? bytes : 0;
}

// out_h / out_w (transposed form only, 0 = 2h+1 / 2w+1): the output size conv_transpose2d's output_padding asks for (2h+1 or 2h+2);
// the extra row / column only sees taps that fall outside the input, i.e. comes out as zeros, as in the reference's op.
int p3d::conv2d_nhwc_run(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                         const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                         int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t out_h, int32_t out_w,
                         void* workspace, int64_t workspace_bytes, int64_t* query, const float* out_scale, p3d_stream_t stream)
{
    return conv2d_nhwc_run_io(x, w, y, dtype, bias, noise, noise_strength, zeros128, n_img, h, wdt, ci, co, w_img_stride, kernel_size, resample, act, gain, clamp,
                              out_h, out_w, workspace, workspace_bytes, query, out_scale, 0, 0, stream);
}

// x_split / y_split (dtype P3D_F32_BF16X3 only): the activations are / the result is to be in the bf16x3 K-row layout — per pixel and 32 channels
// [32 x bf16 hi | 32 x bf16 lo] in the 128 bytes of 32 floats.  Every route takes x_split; y_split needs the halo-slab 3x3 kernel (anything else:
// P3D_ERR_UNSUPPORTED before a launch, the caller asks again for a plain result).
int p3d::conv2d_nhwc_run_io(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                            const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                            int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t out_h, int32_t out_w,
                            void* workspace, int64_t workspace_bytes, int64_t* query, const float* out_scale, int32_t x_split, int32_t y_split,
                            p3d_stream_t stream)
{
    P3D_REQUIRE(!(x_split || y_split) || dtype == P3D_F32_BF16X3, "conv2d_nhwc: pre-split activations are a bf16x3 format");
    P3D_REQUIRE(!y_split || co % 32 == 0, "conv2d_nhwc: a split result needs whole 32-channel rows");
    // query != null: dry run — *query = bytes of split-K scratch this call would like (0: none); nothing is launched
    const bool dry = query != nullptr;
    if (dry) { *query = 0; x = w = zeros128 = (const void*)(uintptr_t)16; y = (void*)(uintptr_t)16; }
    const bool transposed_stride2 = (resample == 1), down2 = (resample == 2);
    P3D_REQUIRE(resample >= 0 && resample <= 2, "conv2d_nhwc: resample must be 0 (same), 1 (transposed x2) or 2 (valid, stride 2)");
    P3D_REQUIRE(x && w && y && zeros128, "conv2d_nhwc: null pointer");
    P3D_REQUIRE(n_img >= 1 && h >= 1 && wdt >= 1 && co >= 1, "conv2d_nhwc: bad sizes");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32 || dtype == P3D_F32_BF16X3 || dtype == P3D_F32_BF16X6, "conv2d_nhwc: dtype must be fp16, fp32, fp32-as-bf16x3 or fp32-as-bf16x6");
    P3D_REQUIRE(kernel_size == 3 || (kernel_size == 1 && !transposed_stride2), "conv2d_nhwc: kernel 3x3, or 1x1 without upsampling");
    P3D_REQUIRE(!out_scale || dtype != P3D_F16, "conv2d_nhwc: the per-image output scale is implemented for fp32 tensors");
    static const bool no_h2t = getenv("P3D_CONV_NO_H2") != nullptr;
    const bool h2t = !no_h2t && resample == 1 && dtype == P3D_F16 && h >= 32 && wdt >= 32 && ci % 32 == 0 && co % BN == 0 && (((uintptr_t)y) & 15u) == 0;   // convT_h2_f16_kernel: 64-byte K rows
    const int bk = dtype == P3D_F16 ? 64 : 32;
    if (ci % bk != 0 && !h2t) return fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: Ci=%d must be a multiple of %d", ci, bk);
    P3D_REQUIRE((((uintptr_t)x) & 15u) == 0 && (((uintptr_t)w) & 15u) == 0 && (((uintptr_t)zeros128) & 15u) == 0, "conv2d_nhwc: x, w and zeros128 must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.noise = noise; a.noise_strength = noise_strength; a.zeros = zeros128;
    a.N = n_img; a.H = h; a.W = wdt; a.Ci = ci; a.Co = co; a.KT = kernel_size * kernel_size; a.w_img_stride = w_img_stride;
    a.act = act; a.gain = gain; a.clamp = clamp; a.isy = a.isx = 1; a.oscale = out_scale;
    a.iscale = tl_in_scale;                                          // (set by p3d_conv2d_nhwc_scaled_in for the duration of its call)
    { static const bool narrow = [] { const char* d = getenv("P3D_CONV_STORE4"); return d && atoi(d) != 0; }(); a.store_narrow = narrow; }
    hipStream_t s = (hipStream_t)stream;
    if (down2) {                                                     // valid (unpadded) correlation at stride 2: conv2d_resample.py:108-111 after its FIR
        P3D_REQUIRE(h >= kernel_size && wdt >= kernel_size, "conv2d_nhwc: image smaller than the kernel");
        a.OH = (h - kernel_size) / 2 + 1; a.OW = (wdt - kernel_size) / 2 + 1; a.osy = a.osx = 1; a.isy = a.isx = 2; a.ncls = 1;
        a.cls[0].SH = a.OH; a.cls[0].SW = a.OW; a.cls[0].ooy = a.cls[0].oox = 0; a.cls[0].ntaps = a.KT;
        for (int t = 0; t < a.KT; ++t) a.cls[0].taps[t] = ConvTap{t / kernel_size, t % kernel_size, t};
        if (y_split) return dry ? (int)P3D_ERR_UNSUPPORTED : fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: a split result is produced by the 3x3 halo kernel only");
        a.fold = fold_batch(a, dtype);
        return launch_conv(a, dtype, s, workspace, workspace_bytes, query, x_split != 0);
    }
    if (!transposed_stride2) {                                       // correlation, "same" padding: input offset = tap - k/2
        a.OH = h; a.OW = wdt; a.osy = a.osx = 1; a.ncls = 1;
        a.cls[0].SH = h; a.cls[0].SW = wdt; a.cls[0].ooy = a.cls[0].oox = 0; a.cls[0].ntaps = a.KT;
        for (int t = 0; t < a.KT; ++t) a.cls[0].taps[t] = ConvTap{t / kernel_size - kernel_size / 2, t % kernel_size - kernel_size / 2, t};
        static const bool no_halo = getenv("P3D_CONV_NO_HALO") != nullptr;
        static const bool no_h2 = getenv("P3D_CONV_NO_H2") != nullptr;
        // the halo kernels have no split-K: they take the layers whose own grid fills the chip (or every layer when the caller brought
        // no scratch); a 512-channel layer at 16^2 / 32^2 is 32 / 128 work-groups with a 144-step K loop — that goes to the generic
        // kernel with its K steps dealt out
        const bool h2_ok = !no_h2 && !no_halo && kernel_size == 3 && dtype == P3D_F16 && h >= 32 && wdt >= 32 && ci % 64 == 0 && co % BN == 0 && (((uintptr_t)y) & 15u) == 0;
        const bool halo_ok = kernel_size == 3 && h >= PH && wdt >= PW && !no_halo && !out_scale && !a.iscale;      // (the per-image scales live in the generic kernel)
        const int64_t own_blocks = h2_ok ? (int64_t)((h + QH - 1) / QH) * ((wdt + QW - 1) / QW) * (co / BN) * n_img
                                         : (int64_t)((h + PH - 1) / PH) * ((wdt + PW - 1) / PW) * ((co + BN - 1) / BN) * n_img;
        const bool have_ws = dry || (workspace != nullptr && workspace_bytes > 0);
        const bool prefer_split = have_ws && (h2_ok || halo_ok) && own_blocks < 192 && ci / bk * a.KT >= 16;
        if (h2_ok && !prefer_split) {
            if (dry) return P3D_OK;
            dim3 grid(((h + QH - 1) / QH) * ((wdt + QW - 1) / QW), co / BN, n_img);
            hipLaunchKernelGGL(conv3x3_h2_f16_kernel<false>, grid, dim3(256), 0, s, a);
            count_launch(FAM_CONV);
            return check_launch("conv3x3_h2_f16");
        }
        static const bool no_r2 = getenv("P3D_CONV_NO_R2") != nullptr;
        const bool r2_ok = !no_r2 && !no_halo && kernel_size == 3 && dtype == P3D_F32_BF16X3 && x_split && h >= 32 && wdt >= 32 && ci % 32 == 0 && co % BN == 0 && !out_scale;
        if (r2_ok && (int64_t)((h + QH - 1) / QH) * ((wdt + QW - 1) / QW) * (co / BN) * n_img >= 192) {      // (smaller grids: the 8 x 16 halo kernel, then split-K)
            if (dry) return P3D_OK;                                       // split activations in: the ring pipeline on 16-channel half rows
            static const bool store2 = [] { const char* d = getenv("P3D_R2_STORE2"); return d && atoi(d) != 0; }();      // (A/B only: the 2-byte stores of the lane = channel layout)
            a.y_split = (y_split && store2) ? 2 : y_split;
            dim3 grid(((h + QH - 1) / QH) * ((wdt + QW - 1) / QW), co / BN, n_img);
            hipLaunchKernelGGL(conv3x3_r2_bf16x3_kernel<false>, grid, dim3(256), 0, s, a, WideRgbTail{});
            count_launch(FAM_CONV);
            return check_launch("conv3x3_r2_bf16x3");
        }
        if (halo_ok && !prefer_split) {                                   // halo-reuse kernel for the plain 3x3 layers
            if (dry) return P3D_OK;
            dim3 grid(((h + PH - 1) / PH) * ((wdt + PW - 1) / PW), (co + BN - 1) / BN, n_img);
            a.y_split = y_split;
            // bf16x6: operands split once per work-group (conv3x3_halo_x6p_kernel) where the grid gives every CU its two work-groups — one alone has nobody to hide its
            // per-tap rendezvous behind (512 -> 512 at 32^2 x 4, 128 work-groups: 0.149 ms against 0.135 for the in-register kernel, profiles/round6_m_*).  P3D_X6_PRESPLIT=0: never
            // (2: always — read at every call, so that one test process can put small geometries through either kernel)
            const char* const x6p_env = dtype == P3D_F32_BF16X6 ? getenv("P3D_X6_PRESPLIT") : nullptr;
            const int x6p_mode = x6p_env ? atoi(x6p_env) : 1;
            const bool x6p = x6p_mode == 2 || (x6p_mode == 1 && own_blocks >= 256);
            if (dtype == P3D_F16)             hipLaunchKernelGGL(conv3x3_halo_kernel<__half>, grid, dim3(256), 0, s, a);
            else if (dtype == P3D_F32_BF16X3 && x_split) hipLaunchKernelGGL((conv3x3_halo_kernel<float, true, true>), grid, dim3(256), 0, s, a);
            else if (dtype == P3D_F32_BF16X3) hipLaunchKernelGGL((conv3x3_halo_kernel<float, true>), grid, dim3(256), 0, s, a);
            else if (dtype == P3D_F32_BF16X6 && x6p && ci % 32 == 0 && co <= 64) hipLaunchKernelGGL(conv3x3_halo_x6p_kernel<true>, grid, dim3(256), 0, s, a);
            else if (dtype == P3D_F32_BF16X6 && x6p && ci % 32 == 0) hipLaunchKernelGGL(conv3x3_halo_x6p_kernel<false>, grid, dim3(256), 0, s, a);
            else if (dtype == P3D_F32_BF16X6) hipLaunchKernelGGL((conv3x3_halo_kernel<float, false, false, true>), grid, dim3(256), 0, s, a);
            else                              hipLaunchKernelGGL(conv3x3_halo_kernel<float>, grid, dim3(256), 0, s, a);
            count_launch(FAM_CONV);
            return check_launch("conv3x3_halo");
        }
        if (y_split) return dry ? (int)P3D_ERR_UNSUPPORTED : fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: a split result is produced by the 3x3 halo kernel only");
        a.fold = fold_batch(a, dtype);
        return launch_conv(a, dtype, s, workspace, workspace_bytes, query, x_split != 0);
    }
    // conv_transpose2d(stride 2, no padding): out[(2i+py), (2j+px)] = sum_{ky = py (mod 2), kx = px (mod 2)} x[i - (ky-py)/2, j - (kx-px)/2] w[ky, kx]
    // -> four dense sub-problems (4 / 2 / 2 / 1 taps), all in ONE launch so the grid fills the chip
    P3D_REQUIRE(!noise && !bias && act == 0, "conv2d_nhwc: the transposed form has no epilogue (the FIR runs next)");
    a.OH = out_h > 0 ? out_h : 2 * h + 1; a.OW = out_w > 0 ? out_w : 2 * wdt + 1; a.osy = a.osx = 2; a.ncls = 4;
    P3D_REQUIRE(a.OH >= 2 * h + 1 && a.OH <= 2 * h + 2 && a.OW >= 2 * wdt + 1 && a.OW <= 2 * wdt + 2, "conv2d_nhwc: transposed output must be 2h+1 or 2h+2");
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvArgs::Cls& c = a.cls[py * 2 + px];
            c.ooy = py; c.oox = px;
            c.SH = (a.OH - py + 1) / 2; c.SW = (a.OW - px + 1) / 2;                // output rows py, py + 2, ... < OH
            c.ntaps = 0;
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2)
                    c.taps[c.ntaps++] = ConvTap{-(ky - py) / 2, -(kx - px) / 2, ky * 3 + kx};
        }
    {
        if (h2t) {
            if (dry) return P3D_OK;
            const int tiles = ((h + 1 + QH - 1) / QH) * ((wdt + 1 + QW - 1) / QW);          // class (0, 0) is the largest: (H + 1) x (W + 1) positions
            hipLaunchKernelGGL(convT_h2_f16_kernel, dim3(tiles, co / BN, n_img * 4), dim3(256), 0, s, a);
            count_launch(FAM_CONV);
            return check_launch("convT_h2_f16");
        }
    }
    if (y_split) return dry ? (int)P3D_ERR_UNSUPPORTED : fail(P3D_ERR_UNSUPPORTED, "conv2d_nhwc: a split result is produced by the 3x3 halo kernel only");
    return launch_conv(a, dtype, s, workspace, workspace_bytes, query, x_split != 0);
}

extern "C" int p3d_torgb_nhwc_f16(const void* x, const float* weight, const float* styles, const float* bias, float* y_nchw,
                                  int32_t n_img, int32_t hw, int32_t ci, int32_t co, float clamp, int32_t accumulate, p3d_stream_t stream)
{
    P3D_REQUIRE(x && weight && styles && y_nchw, "torgb_nhwc_f16: null pointer");
    if (!(ci == 64 || ci == 128 || ci == 256) || co < 1 || co > 32 || (hw & 3))
        return fail(P3D_ERR_UNSUPPORTED, "torgb_nhwc_f16: needs Ci in {64,128,256}, Co <= 32, H*W a multiple of 4 (got %d, %d, %d)", ci, co, hw);
    hipStream_t s = (hipStream_t)stream;
    int blocks = (hw / 32 + 3) / 4;
    const int cap = kNumCU * 2 / (n_img > 0 ? n_img : 1);            // 2 blocks x 4 waves per CU over all images: two waves per SIMD (two tiles of fragments each)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
#define P3D_RGB(K) case K * 16: hipLaunchKernelGGL(torgb_nhwc_kernel<K>, dim3(blocks, n_img), dim3(256), 0, s, (const __half*)x, weight, styles, bias, y_nchw, hw, co, clamp, accumulate); break;
    switch (ci) { P3D_RGB(4) P3D_RGB(8) P3D_RGB(16) }
#undef P3D_RGB
    count_launch(FAM_CONV);
    return check_launch("torgb_nhwc_f16");
}
