// Training-mode convolutions of conv2d_gradfix on the gfx950 matrix cores: the forward op with shared weights, its data gradient
// and its weight gradient (torch_utils/ops/conv2d_gradfix.py:37-45, 107-194), channels-last, fp16 / fp32 with fp32 accumulation.
//
// Every derivative of a convolution is another member of the same family (conv2d_gradfix.py:95-104, 139-143), so two primitives and
// their weight gradient cover all orders:
//   conv2d(x, w[O][I][k][k])            stride 1 / padding k/2  ("same", k in {1, 3})   or   stride 2 / padding 0 (k = 3)
//   conv_transpose2d(x, w[I][O][k][k])  stride 1 / padding k/2                          or   stride 2 / padding 0 (+ output_padding)
// The dense arithmetic of both runs on the implicit-GEMM kernels of conv2d.hip (p3d_conv2d_nhwc) after a one-launch weight re-layout
// to tap-major [O][k*k][I] — the transposed stride-1 form is the plain correlation with the weights mirrored and their channel axes
// swapped, the transposed stride-2 form the polyphase kernel.  The weight gradient is its own kernel (below): a GEMM whose
// contraction runs over PIXELS, so both operands are needed pixel-major while channels-last memory is channel-major.
//
// Skinny 1x1 layers (ToRGB: hundreds of channels -> 3; fromrgb: 6 -> 64) have no GEMM worth the name: they are memory-bound and run
// on a strided VALU kernel.
#include "p3d_common.h"

namespace p3d {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "bf16_split.h"
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- weight re-layout: torch [A][B][taps] -> tap-major [O][taps][I] in the same dtype ------------------------------------------------
// swap == 0: O = A, I = B (conv2d weights).  swap != 0: O = B, I = A (conv_transpose2d weights [in][out]).  flip mirrors the taps.
// split != 0 (T = float, I % 32 == 0): the bf16x3 layout of csrc/conv2d.hip — K rows of [32 x bf16 hi | 32 x bf16 lo] per 32 input channels.
template <class T>
__global__ void __launch_bounds__(256) weight_relayout_kernel(const T* __restrict__ src, T* __restrict__ dst, int A, int B, int taps, int swap, int flip, int split)
{
    const int O = swap ? B : A, I = swap ? A : B;
    const int64_t total = (int64_t)O * taps * I;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % I);
        const int t = (int)((e / I) % taps);
        const int o = (int)(e / ((int64_t)I * taps));
        const int ts = flip ? taps - 1 - t : t;
        const int64_t s = swap ? ((int64_t)i * B + o) * taps + ts : ((int64_t)o * B + i) * taps + ts;
        if constexpr (sizeof(T) == 4) {
            if (split) {
                const float v = src[s];
                const __bf16 hi = (__bf16)v;
                __bf16* row = (__bf16*)dst + (e - i + (i & ~31)) * 2 + (i & 31);
                row[0] = hi;
                row[32] = (__bf16)(v - (float)hi);
                continue;
            }
        }
        dst[e] = src[s];
    }
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------------
// G[cs][cb][ky][kx] = sum over (n, i, j) of S[n, i, j, cs] * B[n, i*stride + ky - pad, j*stride + kx - pad, cb]
// with S the SMALL image of the pair (gy for conv2d, x for conv_transpose2d) and B the big one: for conv2d that is gw[O][I], for
// conv_transpose2d gw[I][O] — each in torch's own layout for that op.
//
// One block = one 128 x 128 tile of (cs, cb) for ONE tap over a contiguous range of pixel chunks (split-K); its fp32 partial tile
// goes to the workspace with plain coalesced stores and a second launch sums the splits and casts (two short launches cost less
// than 16 K atomics per block).  Work-groups of the nine taps of one (tile, split) are neighbours on one XCD: they read the same S
// rows and B rows that differ by a pixel, so the second to ninth read hit that XCD's L2.
//
// Staging is through registers (global -> VGPR -> LDS, double-buffered, one barrier per chunk) because the operands must be
// re-shaped on the way:
//   fp16: v_mfma_f32_32x32x16_f16 wants 8 consecutive K (= pixels) per lane; memory has 8 consecutive CHANNELS per 16 bytes.  A
//         thread loads the same 8 channels of 4 consecutive pixels, transposes the 4 x 8 block in registers (one byte-permute per word) and stores
//         eight 8-byte runs into an LDS image [channel][pixel] (pitch 72 halfs: conflict-free ds_write_b64 and ds_read_b128);
//   fp32: v_mfma_f32_32x32x2_f32 takes ONE float per lane, so the channel-major image [pixel][channel] is already what the
//         fragment reads want (ds_read_b32, lanes = consecutive channels): no transposition, 16-byte stores.
struct WgradArgs {
    const void* s;          // [N][HS][WS][Cs]
    const void* b;          // [N][HB][WB][Cb]
    float* ws;              // partial tiles [ksplit][taps][CsP][CbP]
    int N, HS, WS, Cs, HB, WB, Cb;
    int k, stride, pad;
    int ksplit, chunks, chunks_per_split;
    int tiles_s, tiles_b, CsP, CbP;
    int xcd_pad;            // the grid holds ceil(groups / 8) * 8 groups: group g runs on XCD g % 8 whatever the group count, the surplus work-groups leave at once
    int psplit;             // 1, 2 or 4: how many ways the waves split the chunk's PIXELS instead of the 128 x 128 tile's quadrants.  A side of <= 64 channels leaves half
                            // of the tile empty: the waves that would multiply zeros take a share of the pixels instead and write their own partial tile
                            // (workspace [ksplit * psplit][taps][CsP][CbP]).  4: both sides <= 64 (one 64 x 64 tile is all there is); 2: one side
    int narrow_b;           // psplit == 2: the narrow side is B (the waves pair up along S) — else S
};

template <class T> struct WgradTraits;
template <> struct WgradTraits<__half> { static constexpr int EPC = 8, PITCH = 72;  };   // elements per 16 B, LDS row pitch (elements); pixels per chunk KP: 64
template <> struct WgradTraits<float>  { static constexpr int EPC = 4, PITCH = 128; };   // KP: 16 (32 pixels a chunk, 64 KB, two resident work-groups: 96 vs 104 TFLOP/s)

struct PixPos { int n, i, j; };
__device__ __forceinline__ void pix_advance(PixPos& p, int d, int HS, int WS)
{
    p.j += d;
    if (p.j >= WS) {
        const int q = p.j / WS;
        p.j -= q * WS; p.i += q;
        if (p.i >= HS) { const int q2 = p.i / HS; p.i -= q2 * HS; p.n += q2; }
    }
}

// FAST: both images hold whole, 16-byte aligned channel groups and fewer than 2^31 bytes each (32-bit offsets on a uniform base, no element-wise tail code in
// the loop); fp16 also wants rows of a multiple of four pixels (a lane's four pixels then share one row: one position, one row test).  The general loop carried
// the tail path's branches through every chunk: ~1000 instructions for 16 MFMAs in the fp16 kernel (the ISA of round 4's library).
// SMALL (with FAST, psplit == 4): both sides hold at most 64 channels.  The 128-channel loader would leave half of its threads without a load and a wave with a
// quarter of a chunk's MFMAs against a whole chunk's staging and barrier (61 TFLOP/s on the fp32 64 x 64 layers at 512^2); here every thread loads, the chunk
// is twice as long (KP = two chunks of the split plan) and the LDS image is 64 channels wide.
// X6 (fp32, psplit == 1): the products as bf16x6 (bf16_split.h; P3D_F32_BF16X6) — a chunk's 16 pixels are ONE k-step of v_mfma_f32_32x32x16_bf16: lane (frow, fk) gathers
// pixels 8 fk .. 8 fk + 7 of its channel from the pixel-major LDS image (the same 32 ds_read_b32 per chunk as the exact loop's), splits them in registers and issues 24
// MFMAs (768 cycles) where the f32-input MFMA needs 32 (2048).
// (LDS row pitch of the fp32 image: 128 / 64 floats.  Rows padded so that lanes l and l + 32 of a fragment read land 32 banks apart — 160 / 96, and 132 for the bf16x6
// loop's rows 8 apart — measured no faster, the 64-channel form 20 % slower: profiles/round5_q_wgrad_variants.txt.)
template <class T, int KP, bool FAST, bool SMALL = false, bool X6 = false>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs a)
{
    typedef WgradTraits<T> TR;
    static_assert(FAST || !SMALL, "the 64-channel form exists for aligned geometries only");
    static_assert(!X6 || (sizeof(T) == 4 && !SMALL && KP == 16), "bf16x6 is an arithmetic of the plain fp32 loop (whole 128 x 128 tiles)");
    constexpr int EPC = TR::EPC, PITCH = !SMALL ? TR::PITCH : (sizeof(T) == 2 ? KP + 8 : 64);
    constexpr int KPB = SMALL ? KP / 2 : KP;                                   // pixels of one chunk of the split plan
    constexpr int OPER = sizeof(T) == 2 ? (SMALL ? 64 : 128) * PITCH : KP * PITCH;   // elements of one operand image
    __shared__ __attribute__((aligned(16))) T lds[2][2][OPER];                  // [buffer][S | B][...]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (wave-uniform roles) quadrant (wm, wn) of the tile and the share `part` of the chunk's pixels
    const int wm = a.psplit == 1 ? wave >> 1 : (a.psplit == 2 && a.narrow_b ? wave & 1 : 0);
    const int wn = a.psplit == 1 ? wave & 1 : (a.psplit == 2 && !a.narrow_b ? wave & 1 : 0);
    const int part = a.psplit == 1 ? 0 : (a.psplit == 2 ? wave >> 1 : wave), pmask = a.psplit - 1;
    const int taps = a.k * a.k;
    // work item: (group = (split, tile), tap); the taps of a group sit on one XCD (work-group L runs on XCD L % 8)
    const int groups = a.ksplit * a.tiles_s * a.tiles_b;
    int L = blockIdx.x, g, tap;
    if (a.xcd_pad)              { g = (L & 7) + 8 * ((L >> 3) / taps); tap = (L >> 3) % taps; if (g >= groups) return; }    // (grid padded to 8 groups a round)
    else if ((groups & 7) == 0) { g = (L & 7) + 8 * ((L >> 3) / taps); tap = (L >> 3) % taps; }
    else                        { g = L / taps; tap = L - g * taps; }
    const int tile = g % (a.tiles_s * a.tiles_b), split = g / (a.tiles_s * a.tiles_b);
    const int cs0 = (tile / a.tiles_b) * 128, cb0 = (tile % a.tiles_b) * 128;
    const int ky = tap / a.k, kx = tap - ky * a.k;
    const int dy = ky - a.pad, dx = kx - a.pad;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.chunks) c_end = a.chunks;
    const int64_t Mtot = (int64_t)a.N * a.HS * a.WS;
    const T* const S = (const T*)a.s;
    const T* const B = (const T*)a.b;
    const bool vec_s = (a.Cs % EPC) == 0 && ((uintptr_t)a.s & 15u) == 0;        // 16-byte loads need whole, aligned channel groups
    const bool vec_b = (a.Cb % EPC) == 0 && ((uintptr_t)a.b & 15u) == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;

    // ---- loader role of this thread -------------------------------------------------------------------------------------
    // fp16: pixel quad pq = lane & 15 (pixels 4pq .. 4pq+3 of the chunk), channel group cg = (lane >> 4) + 4 * wave (8 channels)
    // fp32: KP / 8 16-byte pieces r: pixel (tid >> 5) + 8 r of the chunk, channels 4 * (tid & 31) ..
    // SMALL: fp16 pixel quad (lane & 15) + 16 * (wave >> 1), channel group (lane >> 4) + 4 * (wave & 1); fp32 pixel (tid >> 4) + 16 r, channels 4 * (tid & 15) ..
    constexpr int NLD = sizeof(T) == 2 ? 4 : (SMALL ? KP / 16 : KP / 8);       // 16-byte loads per operand per chunk
    int lp[NLD];                                                               // chunk-local pixel of each load
    int lc;                                                                    // first tile-local channel of this thread's loads
    const int pq = SMALL ? (lane & 15) + 16 * (wave >> 1) : (lane & 15);       // (fp16) the thread's pixel quad
    if constexpr (sizeof(T) == 2) {
        lc = ((lane >> 4) + 4 * (SMALL ? (wave & 1) : wave)) * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) lp[q] = pq * 4 + q;
    } else {
        lc = SMALL ? (tid & 15) * 4 : (tid & 31) * 4;
#pragma unroll
        for (int q = 0; q < NLD; ++q) lp[q] = SMALL ? (tid >> 4) + 16 * q : (tid >> 5) + 8 * q;
    }
    const int64_t m_end64 = (int64_t)c_end * KPB < Mtot ? (int64_t)c_end * KPB : Mtot;     // this work-group's pixels end here
    PixPos pos[NLD];                                                           // (n, i, j) of each load's pixel in the current chunk
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int64_t m = (int64_t)c_begin * KPB + lp[q];
        const int64_t per = (int64_t)a.HS * a.WS;
        pos[q].n = (int)(m / per);
        const int rem = (int)(m - (int64_t)pos[q].n * per);
        pos[q].i = rem / a.WS; pos[q].j = rem - pos[q].i * a.WS;
    }
    f32x4 rs[NLD], rb[NLD];                                                    // staged operands of the NEXT chunk

    auto load16 = [&](const T* base, int64_t elem_off, int c_first, int C, bool vec) -> f32x4 {
        // 16 bytes = EPC channels starting at channel c_first of one pixel (elem_off = the pixel's first element); channels >= C read 0
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c_first >= C) return v;
        if (vec) return *(const f32x4*)(base + elem_off + c_first);
        T tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = (c_first + e < C) ? base[elem_off + c_first + e] : (T)0.f;
        return *(const f32x4*)tmp;
    };
    const bool s_ok = cs0 + lc < a.Cs, b_ok = cb0 + lc < a.Cb;                  // (FAST) this thread's channel group exists on either side
    const char* const Sb = (const char*)a.s;
    const char* const Bb = (const char*)a.b;
    auto fetch = [&](int chunk) {
        if constexpr (FAST && sizeof(T) == 2) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const int m0 = chunk * KPB + lp[0];
            const bool live = m0 < (int)m_end64;                                    // (rows of 4 k pixels: the lane's four pixels are live together)
            const int bi = pos[0].i * a.stride + dy, bj0 = pos[0].j * a.stride + dx;
            const bool row_ok = live && b_ok && bi >= 0 && bi < a.HB;
            const unsigned so = (unsigned)(m0 * a.Cs + cs0 + lc) * 2u;
            const unsigned bo = (unsigned)(((pos[0].n * a.HB + bi) * a.WB + bj0) * a.Cb + cb0 + lc) * 2u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int bj = bj0 + q * a.stride;
                rs[q] = (live && s_ok) ? *(const f32x4*)(Sb + (so + (unsigned)(q * a.Cs) * 2u)) : zero;
                rb[q] = (row_ok && bj >= 0 && bj < a.WB) ? *(const f32x4*)(Bb + (bo + (unsigned)(q * a.stride * a.Cb) * 2u)) : zero;
            }
            pix_advance(pos[0], KP, a.HS, a.WS);
            return;
        } else if constexpr (FAST) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int m = chunk * KPB + lp[q];
                const bool live = m < (int)m_end64;
                const int bi = pos[q].i * a.stride + dy, bj = pos[q].j * a.stride + dx;
                const unsigned so = (unsigned)(m * a.Cs + cs0 + lc) * 4u;
                const unsigned bo = (unsigned)(((pos[q].n * a.HB + bi) * a.WB + bj) * a.Cb + cb0 + lc) * 4u;
                rs[q] = (live && s_ok) ? *(const f32x4*)(Sb + so) : zero;
                rb[q] = (live && b_ok && bi >= 0 && bi < a.HB && bj >= 0 && bj < a.WB) ? *(const f32x4*)(Bb + bo) : zero;
                pix_advance(pos[q], KP, a.HS, a.WS);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int64_t m = (int64_t)chunk * KP + lp[q];
            const bool live = m < Mtot;
            f32x4 vs = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                vs = load16(S, m * a.Cs, cs0 + lc, a.Cs, vec_s);
                const int bi = pos[q].i * a.stride + dy, bj = pos[q].j * a.stride + dx;
                if (bi >= 0 && bi < a.HB && bj >= 0 && bj < a.WB)
                    vb = load16(B, (((int64_t)pos[q].n * a.HB + bi) * a.WB + bj) * a.Cb, cb0 + lc, a.Cb, vec_b);
            }
            rs[q] = vs; rb[q] = vb;
            pix_advance(pos[q], KP, a.HS, a.WS);
        }
    };
    auto deposit = [&](int buf) {
        if constexpr (sizeof(T) == 2) {
            // registers hold [pixel q][8 channels]; LDS wants [channel][4 pixels] as one 8-byte run per channel
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                const f32x4* r = op ? rb : rs;
                T* img = lds[buf][op];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                   // register e of each pixel = channels 2e, 2e+1
                    // (element first, THEN reinterpret: __builtin_bit_cast applied to a vector-element expression reads element 0)
                    const float f0 = r[0][e], f1 = r[1][e], f2 = r[2][e], f3 = r[3][e];
                    const unsigned v0 = __float_as_uint(f0), v1 = __float_as_uint(f1), v2 = __float_as_uint(f2), v3 = __float_as_uint(f3);
                    // (the compiler turns these into one v_perm_b32 / v_pack each)
                    const unsigned lo01 = (v0 & 0xffffu) | (v1 << 16), lo23 = (v2 & 0xffffu) | (v3 << 16);
                    const unsigned hi01 = (v0 >> 16) | (v1 & 0xffff0000u), hi23 = (v2 >> 16) | (v3 & 0xffff0000u);
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    *(u32x2*)(img + (lc + 2 * e) * PITCH + pq * 4) = u32x2{lo01, lo23};
                    *(u32x2*)(img + (lc + 2 * e + 1) * PITCH + pq * 4) = u32x2{hi01, hi23};
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                *(f32x4*)(lds[buf][0] + lp[q] * PITCH + lc) = rs[q];
                *(f32x4*)(lds[buf][1] + lp[q] * PITCH + lc) = rb[q];
            }
        }
    };

    constexpr int STEP = KP / KPB;                                             // plan chunks per iteration
    if (c_begin < c_end) fetch(c_begin);
    int buf = 0;
    for (int chunk = c_begin; chunk < c_end; chunk += STEP) {
        deposit(buf);
        __syncthreads();
        if (chunk + STEP < c_end) fetch(chunk + STEP);                         // flies under this chunk's MFMAs
        const T* ls = lds[buf][0];
        const T* lb = lds[buf][1];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < KP / 16; ++kk) {
                if ((kk & pmask) != part) continue;                             // (wave-uniform)
                h8 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = *(const h8*)(ls + (wm * 64 + i * 32 + frow) * PITCH + kk * 16 + fk * 8);
                    fb[i] = *(const h8*)(lb + (wn * 64 + i * 32 + frow) * PITCH + kk * 16 + fk * 8);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (X6) {
            bf8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 a0, a1, b0, b1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = ls[(8 * fk + e) * PITCH + wm * 64 + i * 32 + frow]; a1[e] = ls[(8 * fk + 4 + e) * PITCH + wm * 64 + i * 32 + frow];
                    b0[e] = lb[(8 * fk + e) * PITCH + wn * 64 + i * 32 + frow]; b1[e] = lb[(8 * fk + 4 + e) * PITCH + wn * 64 + i * 32 + frow];
                }
                split3_bf16x8(a0, a1, ah[i], am[i], al[i]);
                split3_bf16x8(b0, b1, bh[i], bm[i], bl[i]);
            }
#pragma unroll
            for (int term = 0; term < 6; ++term)                                // term outermost: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, ah[i], am[i], al[i]), P3D_X6_B(term, bh[j], bm[j], bl[j]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < KP / 2; ++kk) {
                if ((kk & pmask) != part) continue;
                float fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = ls[(kk * 2 + fk) * PITCH + wm * 64 + i * 32 + frow];
                    fb[i] = lb[(kk * 2 + fk) * PITCH + wn * 64 + i * 32 + frow];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        buf ^= 1;
    }

    // partial tile -> workspace [split][tap][CsP][CbP]: accumulator element r of tile (i, j) is row (r&3) + 8(r>>2) + 4 fk, column frow
    float* out = a.ws + ((int64_t)(split * a.psplit + part) * taps + tap) * a.CsP * a.CbP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cb0 + wn * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cs0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                out[(int64_t)row * a.CbP + col] = acc[i][j][r];
            }
        }
}

// ---- fp16 weight gradient with the operands left as they arrive: pixel-major rows in LDS, fragments by the transposing LDS read -------------------------------
// The contraction runs over PIXELS and both images are stored pixel-major, so an MFMA fragment (8 consecutive k for one channel per lane) is a strided gather.
// conv_wgrad_kernel<__half> transposes 4 x 8 blocks in registers on the way into LDS (byte permutes: VALU 0.25-0.38 busy against 0.12-0.19 for the matrix pipe).
// gfx950's ds_read_b64_tr_b16 does that transposition in the LDS read: within a 16-lane group, lane 4 r + q passes the address of row r, columns 4 q .. 4 q + 3 of a
// 4 x 16 matrix of 16-bit elements and lane i receives column i (tests/probes/tr_b16_probe.hip establishes exactly this on the device).  So the 16-byte global
// loads go to LDS as they are — row p of an operand image is pixel p's 128 channels, 256 bytes, rotated by 64 bytes x (p & 3) so that the four rows one read
// touches sit in four different quarters of the 64 banks — and a fragment is two such reads (k 0..3 and 4..7 of the lane's k-group).  Same work decomposition,
// roles and partial-tile layout as conv_wgrad_kernel; FAST geometries, psplit 1 or 2.
typedef __fp16 tr_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// ROWK: image rows are a multiple of the chunk's pixels wide, so a chunk lies inside ONE row and a thread's four pixels share one position, one row test.
// SMALL64 (psplit == 4: both sides <= 64 channels): rows of 64 channels (128 bytes, rotated by 64 bytes x ((p >> 1) & 1)), 128 pixels a chunk (two chunks of the
// split plan), every thread loads, each wave takes two of the chunk's eight 16-pixel steps — the transposing-read form of conv_wgrad_kernel's SMALL loader.
template <bool ROWK, bool SMALL64>
__global__ void __launch_bounds__(256, 2) conv_wgrad_tr_f16_kernel(WgradArgs a)
{
    constexpr int KPB = 64, KP = SMALL64 ? 128 : 64, STEP = KP / KPB;          // pixels per chunk of the split plan / per iteration
    constexpr int ROW = SMALL64 ? 128 : 256, IMG = KP * ROW;                   // bytes per pixel row, bytes per operand image
    constexpr int TPR = ROW / 16, PASS = 256 / TPR;                            // threads per pixel row, pixels per loading pass (four passes per iteration)
    __shared__ __attribute__((aligned(16))) char lds[2][2][IMG];               // [buffer][S | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = SMALL64 ? 0 : (a.psplit == 1 ? wave >> 1 : (a.narrow_b ? wave & 1 : 0));
    const int wn = SMALL64 ? 0 : (a.psplit == 1 ? wave & 1 : (!a.narrow_b ? wave & 1 : 0));
    const int part = SMALL64 ? wave : (a.psplit == 1 ? 0 : wave >> 1), pmask = a.psplit - 1;
    const int taps = a.k * a.k;
    const int groups = a.ksplit * a.tiles_s * a.tiles_b;
    int L = blockIdx.x, g, tap;
    if (a.xcd_pad)              { g = (L & 7) + 8 * ((L >> 3) / taps); tap = (L >> 3) % taps; if (g >= groups) return; }
    else if ((groups & 7) == 0) { g = (L & 7) + 8 * ((L >> 3) / taps); tap = (L >> 3) % taps; }
    else                        { g = L / taps; tap = L - g * taps; }
    const int tile = g % (a.tiles_s * a.tiles_b), split = g / (a.tiles_s * a.tiles_b);
    const int cs0 = (tile / a.tiles_b) * 128, cb0 = (tile % a.tiles_b) * 128;
    const int ky = tap / a.k, kx = tap - ky * a.k;
    const int dy = ky - a.pad, dx = kx - a.pad;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.chunks) c_end = a.chunks;
    const int64_t Mtot = (int64_t)a.N * a.HS * a.WS;
    const int m_end = (int)((int64_t)c_end * KPB < Mtot ? (int64_t)c_end * KPB : Mtot);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // loader: TPR threads cover one pixel's channels (contiguous bytes of global memory), PASS pixels per pass, four passes per iteration
    const int lc = (tid % TPR) * 8;
    int lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) lp[q] = tid / TPR + PASS * q;
    auto rot = [](int p) { return SMALL64 ? 64 * ((p >> 1) & 1) : 64 * (p & 3); };      // bytes an LDS row is rotated by (PASS is a multiple of 4: the same for a thread's four pixels)
    const int wrow = (lc * 2 + rot(tid / TPR)) & (ROW - 1);                     // byte position of this thread's 16 bytes inside its LDS rows
    PixPos pos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t m = (int64_t)c_begin * KPB + lp[q];
        const int64_t per = (int64_t)a.HS * a.WS;
        pos[q].n = (int)(m / per);
        const int rem = (int)(m - (int64_t)pos[q].n * per);
        pos[q].i = rem / a.WS; pos[q].j = rem - pos[q].i * a.WS;
    }
    const bool s_ok = cs0 + lc < a.Cs, b_ok = cb0 + lc < a.Cb;
    const char* const Sb = (const char*)a.s;
    const char* const Bb = (const char*)a.b;
    f32x4 rs[4], rb[4];
    auto fetch = [&](int chunk) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ROWK) {
            const int m0 = chunk * KPB + lp[0];
            const int bi = pos[0].i * a.stride + dy, bj0 = pos[0].j * a.stride + dx;
            const bool row_b = b_ok && bi >= 0 && bi < a.HB;
            const unsigned so = (unsigned)(m0 * a.Cs + cs0 + lc) * 2u;
            const unsigned bo = (unsigned)(((pos[0].n * a.HB + bi) * a.WB + bj0) * a.Cb + cb0 + lc) * 2u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool live = m0 + PASS * q < m_end;                            // (SMALL64: the second half of a double chunk may lie past this work-group's pixels)
                const int bj = bj0 + PASS * q * a.stride;
                rs[q] = (live && s_ok) ? *(const f32x4*)(Sb + (so + (unsigned)(PASS * q * a.Cs) * 2u)) : zero;
                rb[q] = (live && row_b && bj >= 0 && bj < a.WB) ? *(const f32x4*)(Bb + (bo + (unsigned)(PASS * q * a.stride * a.Cb) * 2u)) : zero;
            }
            pix_advance(pos[0], KP, a.HS, a.WS);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = chunk * KPB + lp[q];
            const bool live = m < m_end;
            const int bi = pos[q].i * a.stride + dy, bj = pos[q].j * a.stride + dx;
            const unsigned so = (unsigned)(m * a.Cs + cs0 + lc) * 2u;
            const unsigned bo = (unsigned)(((pos[q].n * a.HB + bi) * a.WB + bj) * a.Cb + cb0 + lc) * 2u;
            rs[q] = (live && s_ok) ? *(const f32x4*)(Sb + so) : zero;
            rb[q] = (live && b_ok && bi >= 0 && bi < a.HB && bj >= 0 && bj < a.WB) ? *(const f32x4*)(Bb + bo) : zero;
            pix_advance(pos[q], KP, a.HS, a.WS);
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f32x4*)(lds[buf][0] + lp[q] * ROW + wrow) = rs[q];
            *(f32x4*)(lds[buf][1] + lp[q] * ROW + wrow) = rb[q];
        }
    };
    // fragment reads: lane = (group gq = lane >> 4, i16 = lane & 15): channel (gq & 1) * 16 + i16 = lane & 31 of its 32-channel tile, k-group gq >> 1 = lane >> 5 —
    // the MFMA operand layout; the lane's ADDRESS is row (i16 >> 2) of the group's 4 x 16 matrix (= pixel), columns 4 (i16 & 3) .. (= channels)
    const int gq = lane >> 4, i16 = lane & 15;
    int preA[2], preB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int prow = (gq >> 1) * 8 + (i16 >> 2);
        preA[i] = prow * ROW + ((((wm * 64 + i * 32 + (gq & 1) * 16) * 2) + (i16 & 3) * 8 + rot(prow)) & (ROW - 1));
        preB[i] = prow * ROW + ((((wn * 64 + i * 32 + (gq & 1) * 16) * 2) + (i16 & 3) * 8 + rot(prow)) & (ROW - 1));
    }
    typedef __attribute__((address_space(3))) tr_f16x4* lds_tr;
    auto frag = [&](const char* img, int pre, int kk) -> h8 {                  // (+ 16 kk and + 4 pixels leave the rotation of the lane's row unchanged)
        const tr_f16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr)(img + pre + kk * 16 * ROW));
        const tr_f16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr)(img + pre + kk * 16 * ROW + 4 * ROW));
        h8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (_Float16)lo[e]; v[4 + e] = (_Float16)hi[e]; }
        return v;
    };
    const int frow = lane & 31, fk = lane >> 5;

    if (c_begin < c_end) fetch(c_begin);
    int buf = 0;
    for (int chunk = c_begin; chunk < c_end; chunk += STEP) {
        deposit(buf);
        __syncthreads();
        if (chunk + STEP < c_end) fetch(chunk + STEP);
#pragma unroll
        for (int kk = 0; kk < KP / 16; ++kk) {
            if ((kk & pmask) != part) continue;                                 // (wave-uniform)
            h8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[i] = frag(lds[buf][0], preA[i], kk); fb[i] = frag(lds[buf][1], preB[i], kk); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }

    float* out = a.ws + ((int64_t)(split * a.psplit + part) * taps + tap) * a.CsP * a.CbP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cb0 + wn * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cs0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                out[(int64_t)row * a.CbP + col] = acc[i][j][r];
            }
        }
}

// sum the splits, cast, and write torch's layout gw[cs][cb][taps].  One thread = four consecutive cb of one (tap, cs): 16-byte loads
// along the workspace's fastest axis, four splits in flight at a time (the first version walked taps x splits with scalar loads from
// 16 K threads and took 100-200 us for 50 MB).
template <class T>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, T* __restrict__ gw, int Cs, int Cb, int taps, int ksplit, int CsP, int CbP, float scale)
{
    const int cb4 = CbP / 4;
    const int64_t total = (int64_t)taps * Cs * cb4;
    const int64_t split_stride = (int64_t)taps * CsP * CbP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % cb4);
        const int cs = (int)((e / cb4) % Cs);
        const int t = (int)(e / ((int64_t)cb4 * Cs));
        const float* src = ws + ((int64_t)t * CsP + cs) * CbP + q * 4;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        int k = 0;
        for (; k + 4 <= ksplit; k += 4) {
            const f32x4 a0 = *(const f32x4*)(src + (k + 0) * split_stride), a1 = *(const f32x4*)(src + (k + 1) * split_stride);
            const f32x4 a2 = *(const f32x4*)(src + (k + 2) * split_stride), a3 = *(const f32x4*)(src + (k + 3) * split_stride);
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; k < ksplit; ++k) s0 += *(const f32x4*)(src + k * split_stride);
        const f32x4 sum = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cb = q * 4 + c;
            if (cb < Cb) st(gw + ((int64_t)cs * Cb + cb) * taps + t, sum[c] * scale);          // (scale 1: the sum itself, bit for bit)
        }
    }
}

// ---- weight gradient of the skinny 1x1 layers (fromrgb: 6 / 18 -> 64, ToRGB: 128 / 256 -> 3): part[split][f][w] = sum_m few[m][f] * many[m][w] --------------------
// The MFMA kernel above pads the few-channel side to a 64-row tile and walks a million pixels for it (460-510 us per call at 512^2, batch 4:
// profiles/round4_o_train_conv_geometries.txt); the contraction is a memory pass over `many`.  A thread owns one 16-byte channel vector of `many` and up to
// eight `few` channels; row lanes split the block's pixels; the sums are folded across the lanes of a wave with xor-shuffles, across the four waves through
// LDS, and one partial tile per block goes to the workspace in the layout wgrad_reduce_kernel sums ([split][CsP][CbP], `few_is_rows` says which side is cs).
template <class T>
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const T* __restrict__ few, const T* __restrict__ many, float* __restrict__ part, int64_t M, int F, int W,
                                                           int rows_per_block, int few_is_rows, int CsP, int CbP)
{
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[4][32][8 * EPC];
    const int wv = W / EPC, cpb = wv < 32 ? wv : 32, rl = 256 / cpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cv = tid % cpb, r0 = tid / cpb, col = blockIdx.y * cpb + cv;
    const int64_t a_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t a_end = a_begin + rows_per_block < M ? a_begin + rows_per_block : M;
    const bool live_col = col < wv;
    float* const out = part + (int64_t)blockIdx.x * CsP * CbP;
    for (int f0 = 0; f0 < F; f0 += 8) {
        float acc[8][EPC];
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[f][e] = 0.f;
        if (live_col) {
            for (int64_t m = a_begin + r0; m < a_end; m += 2 * rl) {            // two pixels in flight per lane
                const bool two = m + rl < a_end;
                T mv[2][EPC];
                *(f32x4*)mv[0] = *(const f32x4*)(many + m * W + col * EPC);
                *(f32x4*)mv[1] = two ? *(const f32x4*)(many + (m + rl) * W + col * EPC) : f32x4{0.f, 0.f, 0.f, 0.f};
                float fv[2][8];                                                 // (unconditional loads of clamped addresses, discarded by a select: as `cond ? load : 0` each of the sixteen
                const int64_t m1 = two ? m + rl : m;                            // was a branch with its own wait)
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const int fc = f0 + f < F ? f0 + f : F - 1;
                    const float v0 = (float)ld(few + m * F + fc), v1 = (float)ld(few + m1 * F + fc);
                    fv[0][f] = (f0 + f < F) ? v0 : 0.f;
                    fv[1][f] = (two && f0 + f < F) ? v1 : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int f = 0; f < 8; ++f)
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[f][e] = fmaf(fv[u][f], (float)ld(&mv[u][e]), acc[f][e]);
            }
        }
        // fold the row lanes of this wave (lanes that share cv differ by multiples of cpb), then the four waves
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = acc[f][e];
                for (int msk = cpb; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
                acc[f][e] = v;
            }
        __syncthreads();                                                            // (red is reused by the next group of few-channels)
        if (lane < cpb) {
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int e = 0; e < EPC; ++e) red[wave][lane][f * EPC + e] = acc[f][e];
        }
        __syncthreads();
        if (wave == 0 && lane < cpb && live_col) {
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                if (f0 + f >= F) continue;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float v = (red[0][lane][f * EPC + e] + red[1][lane][f * EPC + e]) + (red[2][lane][f * EPC + e] + red[3][lane][f * EPC + e]);
                    const int w = col * EPC + e;
                    if (few_is_rows) out[(int64_t)(f0 + f) * CbP + w] = v;
                    else             out[(int64_t)w * CbP + f0 + f] = v;
                }
            }
        }
    }
}

// ---- skinny 1x1: y[n, p, o] = sum_i x[n, p, i] * w[o, i] with a handful of channels on one side, channels-last ---------------------
// Memory-bound by construction (ToRGB reads Ci * sizeof(T) per pixel for 3 outputs; fromrgb writes Co * sizeof(T) per pixel from 6
// inputs).  w is addressed by (w_so, w_si) so the data gradient passes the same tensor transposed.  The weights sit in LDS as fp32.
//
// contract (few outputs): LPP = min(64, Ci / EPC) lanes share a pixel, each owning EPC-channel groups lane, lane + LPP, ...; a lane
// accumulates its partial dot products for up to 8 outputs and the group folds them with xor-shuffles.  Every load is a full 16-byte
// vector and a wave instruction covers whole contiguous pixel rows.
template <class T>
__global__ void __launch_bounds__(256) skinny_contract_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int64_t npix, int Ci, int Co,
                                                              int64_t w_so, int64_t w_si, int lpp)
{
    constexpr int EPC = 16 / sizeof(T);
    extern __shared__ float wl[];                                                // [Co][Ci]
    for (int e = threadIdx.x; e < Co * Ci; e += blockDim.x) wl[e] = ld(w + (e / Ci) * w_so + (e % Ci) * w_si);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpp - 1), slot = lane / lpp, ppw = 64 / lpp;         // lane within its pixel group, group within the wave
    const int groups = Ci / EPC;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    // Fast path (ToRGB and its relatives: every lane of a pixel group owns exactly ONE 16-byte channel group, at most 8 outputs): four pixel groups per
    // iteration, their loads issued together — the loop below it keeps one 16-byte load per lane in flight and spent its life waiting for it
    // (207 us for a 268 MB fp16 image = 1.3 TB/s in a training iteration, profiles/round4_a_train_kernel_stats.csv).
    if (groups == lpp && Co <= 8) {
        constexpr int U = 4;
        for (int64_t p0 = wave * ppw; p0 < npix; p0 += nwaves * ppw * U) {
            f32x4 xr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * nwaves * ppw + slot;
                xr[u] = p < npix ? *(const f32x4*)(x + p * Ci + sub * EPC) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * nwaves * ppw + slot;
                T xv[EPC];
                *(f32x4*)xv = xr[u];
                float acc[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    acc[o] = 0.f;
                    if (o < Co) {
                        const float* wr = wl + o * Ci + sub * EPC;
#pragma unroll
                        for (int q = 0; q < EPC; ++q) acc[o] = fmaf(ld(xv + q), wr[q], acc[o]);
                        for (int m = lpp >> 1; m > 0; m >>= 1) acc[o] += __shfl_xor(acc[o], m, 64);
                    }
                }
                if (p < npix && sub == 0)
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < Co) st(y + p * Co + o, acc[o]);
            }
        }
        return;
    }
    for (int64_t p0 = wave * ppw; p0 < npix; p0 += nwaves * ppw) {
        const int64_t p = p0 + slot;
        const bool live = p < npix;
        for (int o0 = 0; o0 < Co; o0 += 8) {
            float acc[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[o] = 0.f;
            for (int g = sub; g < groups; g += lpp) {
                T xv[EPC];
                *(f32x4*)xv = live ? *(const f32x4*)(x + p * Ci + g * EPC) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o0 + o < Co) {
                        const float* wr = wl + (o0 + o) * Ci + g * EPC;
#pragma unroll
                        for (int q = 0; q < EPC; ++q) acc[o] = fmaf(ld(xv + q), wr[q], acc[o]);
                    }
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                for (int m = lpp >> 1; m > 0; m >>= 1) acc[o] += __shfl_xor(acc[o], m, 64);
            if (live && sub == 0)
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o0 + o < Co) st(y + p * Co + o0 + o, acc[o]);
        }
    }
}

// expand (few inputs): one thread = one pixel x 8 consecutive outputs (one 16-byte store; the Co / 8 lanes of a pixel write its whole
// channel run); the pixel's inputs are read once per thread (same addresses across the lanes of a pixel: one fetch), the weights come
// from LDS (lanes of one output group read the same words: broadcast).
template <class T>
__global__ void __launch_bounds__(256) skinny_expand_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int64_t npix, int Ci, int Co,
                                                            int64_t w_so, int64_t w_si)
{
    extern __shared__ float wl[];                                                // [Ci][CoP], CoP = Co rounded up to 8
    const int CoP = (Co + 7) & ~7;
    for (int e = threadIdx.x; e < Ci * CoP; e += blockDim.x) {
        const int i = e / CoP, o = e - i * CoP;
        wl[e] = o < Co ? ld(w + o * w_so + i * w_si) : 0.f;
    }
    __syncthreads();
    const int ogroups = CoP / 8;
    const int64_t total = npix * ogroups;
    const bool vec = (Co & 7) == 0 && sizeof(T) == 2 && ((uintptr_t)y & 15u) == 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int og = (int)(e % ogroups);
        const int64_t p = e / ogroups;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.f;
        for (int i = 0; i < Ci; ++i) {
            const float xv = ld(x + p * Ci + i);
            const f32x4 w0 = *(const f32x4*)(wl + i * CoP + og * 8), w1 = *(const f32x4*)(wl + i * CoP + og * 8 + 4);
#pragma unroll
            for (int o = 0; o < 4; ++o) { acc[o] = fmaf(xv, w0[o], acc[o]); acc[4 + o] = fmaf(xv, w1[o], acc[4 + o]); }
        }
        T* yp = y + p * Co + og * 8;
        if (vec) {
            h8 hv;
#pragma unroll
            for (int o = 0; o < 8; ++o) hv[o] = (_Float16)acc[o];
            *(h8*)yp = hv;
        } else if ((Co & 7) == 0 && sizeof(T) == 4 && ((uintptr_t)y & 15u) == 0) {
            *(f32x4*)yp = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *(f32x4*)(yp + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (og * 8 + o < Co) st(yp + o, acc[o]);
        }
    }
}

static int wgrad_pixel_split(int cs, int cb)
{
    static const bool no_half = getenv("P3D_WGRAD_NO_HALF") != nullptr;         // (A/B switch of the measurement scripts)
    if (cs <= 64 && cb <= 64) return 4;
    return (cs <= 64 || cb <= 64) && !no_half ? 2 : 1;
}

static bool wgrad_plan_old()
{
    static const bool v = getenv("P3D_WGRAD_PLAN_OLD") != nullptr;              // (A/B switch of the measurement scripts)
    return v;
}

// Split of the pixel axis.  The kernel is MFMA-bound and its work-groups cost the same, so the launch ends when the fullest CU does: every work-group should be
// resident at once, and the same number on every CU.  Work-group L runs on XCD L % 8 (32 CUs, r resident work-groups each) and a group's taps share an XCD, so
// the unit is groups per XCD: floor(32 r / taps) of them, 8 times that in the launch, floor(that / tiles) splits.  Round 4's plan aimed at 3 work-groups per
// CU and rounded UP: 22 splits of a 256 x 256 x 9 problem = 99 work-groups on an XCD's 96 places, 88 TFLOP/s; 28 splits at r = 4 (126 on 128) run 105
// (profiles/round4_t_wgrad_variants.txt: every variant that overfills an XCD is the slow one of its row).
static int wgrad_plan(int dtype, int64_t pixels, int cs, int cb, int k, int* ksplit, int* chunks, int* cps)
{
    const int KP = dtype == P3D_F16 ? 64 : 16;
    const int taps = k * k;
    const int tiles = ceil_div(cs, 128) * ceil_div(cb, 128);
    const int64_t nchunks = (pixels + KP - 1) / KP;
    static const int env_per_cu = getenv("P3D_WGRAD_WG_PER_CU") ? atoi(getenv("P3D_WGRAD_WG_PER_CU")) : 0;
    int64_t per, split;
    if (wgrad_plan_old()) {
        const int per_cu = env_per_cu > 0 ? env_per_cu : 3;
        int64_t want = ((int64_t)per_cu * kNumCU + tiles * taps - 1) / (tiles * taps);
        if (want < 1) want = 1;
        if (want > nchunks) want = nchunks;
        if (want > 1024) want = 1024;
        per = (nchunks + want - 1) / want;
        split = (nchunks + per - 1) / per;
        if (tiles * split >= 8) {                                               // a multiple of 8 groups keeps a group's taps on one XCD
            const int64_t up = ((tiles * split + 7) / 8) * 8;
            if (up % tiles == 0 && up / tiles <= nchunks) { split = up / tiles; per = (nchunks + split - 1) / split; }
        }
    } else {
        // resident work-groups per CU: fp16 holds 72 KB of LDS (2); fp32 32 KB and 120 VGPRs (4)
        const int per_cu = env_per_cu > 0 ? env_per_cu : (dtype == P3D_F16 ? 2 : 4);
        int gpx = (kNumCU / 8) * per_cu / taps;                                 // groups per XCD
        if (gpx < 1) gpx = 1;
        int64_t want = 8 * gpx / tiles;
        if (want < 1) want = 1;
        if (want > nchunks) want = nchunks;
        if (want > 1024) want = 1024;
        per = (nchunks + want - 1) / want;
        if (cs <= 64 && cb <= 64 && per > 1) per = (per + 1) & ~(int64_t)1;       // the 64-channel kernels take two chunks an iteration: whole iterations per split, each
        split = (nchunks + per - 1) / per;                                      // starting on a multiple of its 2 x KP pixels (what lets a row-aligned image share one position)
    }
    *ksplit = (int)split; *chunks = (int)nchunks; *cps = (int)per;
    return taps;
}

} // namespace p3d

using namespace p3d;

enum { MODE_SAME = 0, MODE_STRIDE2 = 1 };

static int relayout(const void* w, void* dst, int dtype, int A, int B, int taps, int swap, int flip, hipStream_t s)
{
    const int64_t total = (int64_t)A * B * taps;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dtype == P3D_F16) hipLaunchKernelGGL(weight_relayout_kernel<__half>, dim3(blocks), dim3(256), 0, s, (const __half*)w, (__half*)dst, A, B, taps, swap, flip, 0);
    else                  hipLaunchKernelGGL(weight_relayout_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)w, (float*)dst, A, B, taps, swap, flip, dtype == P3D_F32_BF16X3);
    count_launch(FAM_CONV);
    return check_launch("conv weight relayout");
}

static bool mfma_channels_ok(int dtype, int ci) { return ci % (dtype == P3D_F16 ? 64 : 32) == 0; }      // (P3D_F32_BF16X3: fp32 tensors, 32)

static int conv_forward_impl(const void* x, const void* weight, void* y, void* w_scratch, const void* zeros128, int dtype,
                             int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                             int32_t transposed, int32_t out_h, int32_t out_w, void* workspace, int64_t workspace_bytes, int64_t* query, p3d_stream_t stream)
{
    const bool dry = query != nullptr;
    if (dry) { *query = 0; x = weight = zeros128 = (const void*)(uintptr_t)16; y = w_scratch = (void*)(uintptr_t)16; }
    P3D_REQUIRE(x && weight && y, "conv2d_forward: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32 || dtype == P3D_F32_BF16X3 || dtype == P3D_F32_BF16X6, "conv2d_forward: dtype must be fp16, fp32, fp32-as-bf16x3 or fp32-as-bf16x6");
    P3D_REQUIRE(n_img >= 1 && h >= 1 && wdt >= 1 && ci >= 1 && co >= 1, "conv2d_forward: bad sizes");
    P3D_REQUIRE((kernel_size == 3 && (stride == 1 || stride == 2)) || (kernel_size == 1 && stride == 1), "conv2d_forward: 3x3 at stride 1 / 2 or 1x1 at stride 1");
    hipStream_t s = (hipStream_t)stream;
    const int taps = kernel_size * kernel_size;
    if (kernel_size == 1 && (!mfma_channels_ok(dtype, ci) || co < 32)) {
        if (dtype == P3D_F32_BF16X3 || dtype == P3D_F32_BF16X6) dtype = P3D_F32;      // the skinny kernels are plain fp32 VALU
        // skinny 1x1 (either direction: a transposed 1x1 is the same product with the weight read transposed), channels-last
        if (dry) return P3D_OK;
        const int64_t npix = (int64_t)n_img * h * wdt;
        const int64_t w_so = transposed ? 1 : ci, w_si = transposed ? co : 1;   // conv2d: w[o][i]; conv_transpose2d: w[i][o]
        const int epc = dtype == P3D_F16 ? 8 : 4;
        P3D_REQUIRE((int64_t)ci * ((co + 7) & ~7) * 4 <= 64 * 1024, "conv2d_forward: skinny 1x1 weights must fit 64 KB of LDS");
        if (co <= ci && ci % epc == 0 && (((uintptr_t)x) & 15u) == 0) {                            // many -> few
            int lpp = 1;
            while (lpp < 64 && lpp * 2 <= ci / epc) lpp *= 2;
            const int ppw = 64 / lpp;
            int64_t blocks = (npix + 4 * ppw - 1) / (4 * ppw);
            if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
            const size_t lds = (size_t)ci * co * 4;
            if (dtype == P3D_F16) hipLaunchKernelGGL(skinny_contract_kernel<__half>, dim3((int)blocks), dim3(256), lds, s, (const __half*)x, (const __half*)weight, (__half*)y, npix, ci, co, w_so, w_si, lpp);
            else                  hipLaunchKernelGGL(skinny_contract_kernel<float>, dim3((int)blocks), dim3(256), lds, s, (const float*)x, (const float*)weight, (float*)y, npix, ci, co, w_so, w_si, lpp);
            count_launch(FAM_CONV);
            return check_launch("skinny_contract");
        }
        const int64_t total = npix * ((co + 7) / 8);
        int64_t blocks = (total + 255) / 256;
        if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
        const size_t lds = (size_t)ci * ((co + 7) & ~7) * 4;
        if (dtype == P3D_F16) hipLaunchKernelGGL(skinny_expand_kernel<__half>, dim3((int)blocks), dim3(256), lds, s, (const __half*)x, (const __half*)weight, (__half*)y, npix, ci, co, w_so, w_si);
        else                  hipLaunchKernelGGL(skinny_expand_kernel<float>, dim3((int)blocks), dim3(256), lds, s, (const float*)x, (const float*)weight, (float*)y, npix, ci, co, w_so, w_si);
        count_launch(FAM_CONV);
        return check_launch("skinny_expand");
    }
    P3D_REQUIRE(w_scratch && zeros128, "conv2d_forward: the MFMA route needs w_scratch (Co*Ci*k*k elements) and zeros128");
    if (!mfma_channels_ok(dtype, ci) && !(dtype == P3D_F16 && transposed && stride == 2 && ci % 32 == 0 && co % 128 == 0 && h >= 32 && wdt >= 32))
        return fail(P3D_ERR_UNSUPPORTED, "conv2d_forward: Ci=%d must be a multiple of %d (pad the channels)", ci, dtype == P3D_F16 ? 64 : 32);
    int rc;
    if (!transposed) {
        rc = dry ? P3D_OK : relayout(weight, w_scratch, dtype, co, ci, taps, 0, 0, s);           // wm[o][t][i] = w[o][i][t]
        if (rc != P3D_OK) return rc;
        return conv2d_nhwc_run(x, w_scratch, y, dtype, nullptr, nullptr, nullptr, zeros128, n_img, h, wdt, ci, co, 0, kernel_size, stride == 2 ? 2 : 0, 0, 1.f, -1.f, 0, 0, workspace, workspace_bytes, query, nullptr, stream);
    }
    if (stride == 1) {                                                                            // = correlation with mirrored taps and swapped channel axes
        rc = dry ? P3D_OK : relayout(weight, w_scratch, dtype, ci, co, taps, 1, 1, s);           // wm[o][t][i] = w[i][o][taps-1-t]
        if (rc != P3D_OK) return rc;
        return conv2d_nhwc_run(x, w_scratch, y, dtype, nullptr, nullptr, nullptr, zeros128, n_img, h, wdt, ci, co, 0, kernel_size, 0, 0, 1.f, -1.f, 0, 0, workspace, workspace_bytes, query, nullptr, stream);
    }
    rc = dry ? P3D_OK : relayout(weight, w_scratch, dtype, ci, co, taps, 1, 0, s);               // wm[o][t][i] = w[i][o][t]
    if (rc != P3D_OK) return rc;
    return conv2d_nhwc_run(x, w_scratch, y, dtype, nullptr, nullptr, nullptr, zeros128, n_img, h, wdt, ci, co, 0, kernel_size, 1, 0, 1.f, -1.f, out_h, out_w, workspace, workspace_bytes, query, nullptr, stream);
}

extern "C" int p3d_conv2d_forward(const void* x, const void* weight, void* y, void* w_scratch, const void* zeros128, int dtype,
                                  int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                                  int32_t transposed, int32_t out_h, int32_t out_w, void* workspace, int64_t workspace_bytes, p3d_stream_t stream)
{
    return conv_forward_impl(x, weight, y, w_scratch, zeros128, dtype, n_img, h, wdt, ci, co, kernel_size, stride, transposed, out_h, out_w, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int64_t p3d_conv2d_forward_workspace(int dtype, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                                                int32_t transposed)
{
    int64_t bytes = 0;
    const int rc = conv_forward_impl(nullptr, nullptr, nullptr, nullptr, nullptr, dtype, n_img, h, wdt, ci, co, kernel_size, stride, transposed, 0, 0, nullptr, 0, &bytes, nullptr);
    return rc == P3D_OK ? bytes : 0;
}

extern "C" int p3d_conv2d_bwd_data(const void* gy, const void* weight, void* gx, void* w_scratch, const void* zeros128, int dtype,
                                   int32_t n_img, int32_t gy_h, int32_t gy_w, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                                   int32_t transposed, int32_t x_h, int32_t x_w, void* workspace, int64_t workspace_bytes, p3d_stream_t stream)
{
    // d(input) of op(transposed) is op(!transposed) over the same weight tensor with input / output channels trading places
    // (conv2d_gradfix.py:139-143); x_h / x_w fix the output_padding when that op is the transposed one
    return conv_forward_impl(gy, weight, gx, w_scratch, zeros128, dtype, n_img, gy_h, gy_w, co, ci, kernel_size, stride, !transposed, x_h, x_w, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int64_t p3d_conv2d_bwd_weight_workspace(int dtype, int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t c_big, int32_t kernel_size)
{
    int ksplit, chunks, cps;
    const int taps = wgrad_plan(dtype, (int64_t)n_img * small_h * small_w, c_small, c_big, kernel_size, &ksplit, &chunks, &cps);
    const int per_wave = wgrad_pixel_split(c_small, c_big);                       // WgradArgs::psplit partial tiles per split
    return (int64_t)ksplit * per_wave * taps * (ceil_div(c_small, 128) * 128) * (ceil_div(c_big, 128) * 128) * 4;
}

static int bwd_weight_impl(const void* small_img, const void* big_img, void* gw, void* workspace, int64_t workspace_bytes, int dtype,
                           int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t big_h, int32_t big_w, int32_t c_big,
                           int32_t kernel_size, int32_t stride, int32_t pad, bool out_f32, float scale, p3d_stream_t stream)
{
    P3D_REQUIRE(small_img && big_img && gw && workspace, "conv2d_bwd_weight: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32 || dtype == P3D_F32_BF16X6, "conv2d_bwd_weight: dtype must be fp16, fp32 or fp32-as-bf16x6");
    const bool x6 = dtype == P3D_F32_BF16X6;                                     // fp32 tensors; the arithmetic of the whole-tile kernel only (anything else: the exact kernels)
    if (x6) dtype = P3D_F32;
    P3D_REQUIRE(n_img >= 1 && small_h >= 1 && small_w >= 1 && big_h >= 1 && big_w >= 1 && c_small >= 1 && c_big >= 1, "conv2d_bwd_weight: bad sizes");
    P3D_REQUIRE((kernel_size == 1 || kernel_size == 3) && (stride == 1 || stride == 2) && pad >= 0 && pad <= 1, "conv2d_bwd_weight: k in {1,3}, stride in {1,2}, pad in {0,1}");
    P3D_REQUIRE(workspace_bytes >= p3d_conv2d_bwd_weight_workspace(dtype, n_img, small_h, small_w, c_small, c_big, kernel_size), "conv2d_bwd_weight: workspace too small");
    P3D_REQUIRE((((uintptr_t)workspace) & 15u) == 0, "conv2d_bwd_weight: workspace must be 16-byte aligned");
    WgradArgs a{};
    a.s = small_img; a.b = big_img; a.ws = (float*)workspace;
    a.N = n_img; a.HS = small_h; a.WS = small_w; a.Cs = c_small; a.HB = big_h; a.WB = big_w; a.Cb = c_big;
    a.k = kernel_size; a.stride = stride; a.pad = pad;
    const int taps = wgrad_plan(dtype, (int64_t)n_img * small_h * small_w, c_small, c_big, kernel_size, &a.ksplit, &a.chunks, &a.chunks_per_split);
    a.tiles_s = ceil_div(c_small, 128); a.tiles_b = ceil_div(c_big, 128);
    a.CsP = a.tiles_s * 128; a.CbP = a.tiles_b * 128;
    a.psplit = wgrad_pixel_split(c_small, c_big);
    a.narrow_b = c_big <= 64 ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    {   // skinny 1x1 (one side with a handful of channels, the other a whole number of 16-byte vectors): the memory-pass kernel
        const int epc = dtype == P3D_F16 ? 8 : 4;
        const bool few_small = c_small <= 32 && c_big % epc == 0, few_big = c_big <= 32 && c_small % epc == 0;
        static const bool no_skinny = getenv("P3D_WGRAD_NO_SKINNY") != nullptr;
        if (!no_skinny && kernel_size == 1 && stride == 1 && pad == 0 && small_h == big_h && small_w == big_w && (few_small || few_big)) {
            const bool few_is_rows = few_small && !(few_big && c_big < c_small);            // which side is the `few` one (the smaller if both qualify)
            const void* few = few_is_rows ? small_img : big_img;
            const void* many = few_is_rows ? big_img : small_img;
            const int F = few_is_rows ? c_small : c_big, W = few_is_rows ? c_big : c_small;
            const int CsP = c_small, CbP = (c_big + 3) & ~3;
            const int64_t M = (int64_t)n_img * small_h * small_w;
            const int wv = W / epc, cpb = wv < 32 ? wv : 32, rl = 256 / cpb;
            int64_t rows = (M + 1023) / 1024;
            if (rows < (int64_t)rl * 16) rows = (int64_t)rl * 16;
            const int64_t nsplit = (M + rows - 1) / rows;
            const int64_t need = nsplit * CsP * CbP * 4;
            if ((cpb & (cpb - 1)) == 0 && need <= workspace_bytes && (((uintptr_t)many) & 15u) == 0 && rows < (1ll << 31)) {
                dim3 grid((unsigned)nsplit, (unsigned)((wv + cpb - 1) / cpb));
                if (dtype == P3D_F16) hipLaunchKernelGGL(skinny_wgrad_kernel<__half>, grid, dim3(256), 0, s, (const __half*)few, (const __half*)many, a.ws, M, F, W, (int)rows, (int)few_is_rows, CsP, CbP);
                else                  hipLaunchKernelGGL(skinny_wgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)few, (const float*)many, a.ws, M, F, W, (int)rows, (int)few_is_rows, CsP, CbP);
                count_launch(FAM_CONV);
                int rc2 = check_launch("skinny_wgrad");
                if (rc2 != P3D_OK) return rc2;
                const int64_t total2 = (int64_t)c_small * (CbP / 4);
                const int rb2 = (int)((total2 + 255) / 256 < 8192 ? (total2 + 255) / 256 : 8192);
                if (dtype == P3D_F16 && !out_f32) hipLaunchKernelGGL(wgrad_reduce_kernel<__half>, dim3(rb2), dim3(256), 0, s, a.ws, (__half*)gw, c_small, c_big, 1, (int)nsplit, CsP, CbP, scale);
                else                              hipLaunchKernelGGL(wgrad_reduce_kernel<float>, dim3(rb2), dim3(256), 0, s, a.ws, (float*)gw, c_small, c_big, 1, (int)nsplit, CsP, CbP, scale);
                count_launch(FAM_CONV);
                return check_launch("skinny_wgrad reduce");
            }
        }
    }
    a.xcd_pad = wgrad_plan_old() ? 0 : 1;
    const int ngroups = a.ksplit * a.tiles_s * a.tiles_b;
    const int blocks = (a.xcd_pad ? (ngroups + 7) / 8 * 8 : ngroups) * taps;
    const int esz = dtype == P3D_F16 ? 2 : 4, epc16 = 16 / esz;
    static const bool no_fast = getenv("P3D_WGRAD_NO_FAST") != nullptr;          // (A/B switches of the measurement scripts)
    static const bool no_small = getenv("P3D_WGRAD_NO_SMALL") != nullptr;
    const bool fast = !no_fast && c_small % epc16 == 0 && c_big % epc16 == 0 && ((((uintptr_t)small_img) | ((uintptr_t)big_img)) & 15u) == 0
                      && (int64_t)n_img * small_h * small_w * c_small * esz < (1ll << 31) && (int64_t)n_img * big_h * big_w * c_big * esz < (1ll << 31)
                      && (dtype != P3D_F16 || small_w % 4 == 0);
    const bool small64 = fast && !no_small && a.psplit == 4;
    static const bool no_tr = getenv("P3D_WGRAD_NO_TR") != nullptr;
    // (the transposing-read kernel handles its pixels one by one: no rows-of-four requirement)
    const bool fast_tr = !no_fast && !no_tr && dtype == P3D_F16 && c_small % 8 == 0 && c_big % 8 == 0 && ((((uintptr_t)small_img) | ((uintptr_t)big_img)) & 15u) == 0
                         && (int64_t)n_img * small_h * small_w * c_small * 2 < (1ll << 31) && (int64_t)n_img * big_h * big_w * c_big * 2 < (1ll << 31);
    static const bool no_tr_small = getenv("P3D_WGRAD_NO_TR_SMALL") != nullptr;
    if (dtype == P3D_F16) {
        if (fast_tr && a.psplit == 4 && !no_small && !no_tr_small) {
            // (one position per thread needs every iteration to start on a multiple of its 128 pixels: whole double chunks per split — what wgrad_plan hands out)
            if (small_w % 128 == 0 && (a.chunks_per_split % 2 == 0 || a.ksplit == 1)) hipLaunchKernelGGL((conv_wgrad_tr_f16_kernel<true, true>), dim3(blocks), dim3(256), 0, s, a);
            else                    hipLaunchKernelGGL((conv_wgrad_tr_f16_kernel<false, true>), dim3(blocks), dim3(256), 0, s, a);
        }
        else if (small64) hipLaunchKernelGGL((conv_wgrad_kernel<__half, 128, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (fast_tr && a.psplit != 4 && small_w % 64 == 0) hipLaunchKernelGGL((conv_wgrad_tr_f16_kernel<true, false>), dim3(blocks), dim3(256), 0, s, a);
        else if (fast_tr && a.psplit != 4) hipLaunchKernelGGL((conv_wgrad_tr_f16_kernel<false, false>), dim3(blocks), dim3(256), 0, s, a);
        else if (fast) hipLaunchKernelGGL((conv_wgrad_kernel<__half, 64, true>), dim3(blocks), dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv_wgrad_kernel<__half, 64, false>), dim3(blocks), dim3(256), 0, s, a);
    } else {
        if (small64)   hipLaunchKernelGGL((conv_wgrad_kernel<float, 32, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (fast && x6 && a.psplit == 1) hipLaunchKernelGGL((conv_wgrad_kernel<float, 16, true, false, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (fast) hipLaunchKernelGGL((conv_wgrad_kernel<float, 16, true>), dim3(blocks), dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv_wgrad_kernel<float, 16, false>), dim3(blocks), dim3(256), 0, s, a);
    }
    count_launch(FAM_CONV);
    int rc = check_launch("conv_wgrad");
    if (rc != P3D_OK) return rc;
    const int64_t total = (int64_t)taps * c_small * (a.CbP / 4);
    const int rblocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == P3D_F16 && !out_f32) hipLaunchKernelGGL(wgrad_reduce_kernel<__half>, dim3(rblocks), dim3(256), 0, s, a.ws, (__half*)gw, c_small, c_big, taps, a.ksplit * a.psplit, a.CsP, a.CbP, scale);
    else                              hipLaunchKernelGGL(wgrad_reduce_kernel<float>, dim3(rblocks), dim3(256), 0, s, a.ws, (float*)gw, c_small, c_big, taps, a.ksplit * a.psplit, a.CsP, a.CbP, scale);
    count_launch(FAM_CONV);
    return check_launch("conv_wgrad reduce");
}

extern "C" int p3d_conv2d_bwd_weight(const void* small_img, const void* big_img, void* gw, void* workspace, int64_t workspace_bytes, int dtype,
                                     int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t big_h, int32_t big_w, int32_t c_big,
                                     int32_t kernel_size, int32_t stride, int32_t pad, p3d_stream_t stream)
{
    return bwd_weight_impl(small_img, big_img, gw, workspace, workspace_bytes, dtype, n_img, small_h, small_w, c_small, big_h, big_w, c_big, kernel_size, stride, pad, false, 1.f, stream);
}

extern "C" int p3d_conv2d_bwd_weight_scaled(const void* small_img, const void* big_img, float* gw_f32, void* workspace, int64_t workspace_bytes, int dtype,
                                            int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t big_h, int32_t big_w, int32_t c_big,
                                            int32_t kernel_size, int32_t stride, int32_t pad, float scale, p3d_stream_t stream)
{
    return bwd_weight_impl(small_img, big_img, gw_f32, workspace, workspace_bytes, dtype, n_img, small_h, small_w, c_small, big_h, big_w, c_big, kernel_size, stride, pad, true, scale, stream);
}
