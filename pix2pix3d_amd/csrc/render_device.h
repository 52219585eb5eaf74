// Device-side pieces of the fused tri-plane ray-marcher shared by render.hip (inference / forward) and render_bwd.hip
// (training: forward with tape + point-wise backward).  See render.hip for the mapping notes.
#pragma once
#include "p3d_common.h"
#include <math.h>

namespace p3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4  __attribute__((ext_vector_type(4)));
#include "bf16_split.h"

// ---- decoder stream layout (floats); identical in the packed global buffer and in LDS ------------
constexpr int kNetStride = 4096;                 // per net: 64 MFMA steps x 64 lanes
constexpr int OFF_B1   = 2 * kNetStride;         // [net][half][32]  hidden biases in accumulator order
constexpr int OFF_B2   = OFF_B1 + 128;           // [net][half][16]  colour biases in accumulator order
constexpr int OFF_W2S  = OFF_B2 + 64;            // [half][32]       density row of the density net
constexpr int OFF_B2S  = OFF_W2S + 64;           // [1]              density bias
constexpr int kDecoderFloats = 8464;             // padded to 16 floats
// Two-plane-set renderer (ImportanceSemanticRenderer, renderer.py:256-438): net 0's first layer reads 64 inputs, cat(texture features,
// semantic features); the weights of inputs 32..63 follow the common stream as one more 32-step block in the same per-lane order.
constexpr int OFF_W1X  = kDecoderFloats;
constexpr int kDecoderFloatsDual = kDecoderFloats + 2048;
// L1X6 (render_forward_kernel<.., L1X6>, p3d_pack_decoder_l1x6): layer 1 of every net as bf16x6 (csrc/bf16_split.h) — its weights sit in LDS as three bf16 images
// [block 4][lane 64][8] of 4 KB each: hi and mid in the 8 KB the net's fp32 layer-1 stream occupied, lo (both nets) in 8 KB behind the common stream.
constexpr int OFF_L1LO = kDecoderFloats;
constexpr int kDecoderFloatsL1X6 = kDecoderFloats + 2048;
constexpr int kPitch = 33;                       // LDS pitch of the per-wave [sample][ray] tile
constexpr int kMaxS = 64;                        // max coarse / fine samples per ray
constexpr int kWaveTile = kMaxS * kPitch + 128;  // + two 64-float scratch rows (unused since the importance sampler went to registers; kept: the tile bases are 16-byte aligned with it)
constexpr int kWavesPerBlock = 8;
constexpr int kFeatPitch = 36;                   // floats per ray in the cooperative gather's hand-over tile (144 B: 16 consecutive rays start on 16 different bank quads)
constexpr int kFeatTile = 32 * kFeatPitch;       // per wave

struct RenderArgs {
    const float* planes;      // [N][3][H][W][32]
    const float* planes2;     // DUAL kernels: the semantic plane set (same sizes and strides); `planes` is then the texture set
    const float* decoder;     // kDecoderFloats (kDecoderFloatsDual for the DUAL kernels), see p3d_pack_decoder
    const float* ray_o;       // [N*M][3]
    const float* ray_d;       // [N*M][3]
    const float* u_coarse;    // [N*M][Sc]
    const float* u_fine;      // [N*M][Sf]
    const float* t_start;     // optional [N*M] per-ray limits ('auto' ray range), else null
    const float* t_end;
    float* feat;              // [N*M][n_nets*32]
    float* depth;             // [N*M]   (unclamped; p3d_render_clamp_depth finishes it)
    float* wsum;              // [N*M]
    float* dbg_fine;          // optional [N*M][Sf] sorted fine depths
    float* dbg_wcoarse;       // optional [N*M][Sc-1] coarse weights
    int*   dbg_bins;          // optional [N*M][Sf] searchsorted index of every importance draw, in draw order (p3d_render_forward_debug: the integer test)
    // training tape (render_bwd.hip): upstream gradients in, per-interval / per-sample records out
    const float* g_feat;      // [N*M][n_nets*32] dL/dfeat
    const float* g_wsum;      // optional [N*M] dL/dwsum
    float* tape_i;            // [N*M][S-1][4]  alpha, T (before the interval), colour part of dL/dw, sigma_mid
    float* tape_s;            // [N*M][S][4]    z, 0.5 (w[k-1] + w[k]), dL/dsigma_k, unused
    unsigned* minmax;         // [2] ordered-uint encoded min / max of all sample depths
    int total_rays, rays_per_img, res, H, W, Sc, Sf;          // res: image side when the rays form a res x res raster (else 0)
    int64_t plane_stride, pix_stride, img_stride;   // texel (n, p, y, x) starts at n*img_stride + p*plane_stride + (y*W + x)*pix_stride
    unsigned plane_bytes, pix_bytes, img_bytes, planes_total_bytes;   // the same strides in bytes (everything fits 32 bits, checked on the host)
    float ray_start, ray_end, coord_scale, lin_step;
    int disparity, white_back, sem_sigmoid;
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned order_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// Hardware transcendentals (v_exp_f32 = 2^x, v_log_f32 = log2, ~1 ulp): two instructions per exp/log instead of the
// range-checked library expansions; arguments here are always in the safe range (no denormal inputs matter).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float softplus20(float x) {      // torch.nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : fast_log(1.f + fast_exp(x));
}
// The same in the log2 domain: with layer 1's weights and biases pre-scaled by log2(e) (see the LDS copy in render_forward_kernel) the
// accumulator holds xs = x log2(e) and softplus(x) / ln 2 = log2(1 + 2^xs) — v_exp, v_add, v_log with no scaling multiplies; the 1 / ln 2 is
// folded into layer 2.  x > 20 <=> xs > 20 log2(e) keeps torch's threshold semantics.
__device__ __forceinline__ float softplus20_log2(float xs) {
    return xs > 28.853900817779268f ? xs : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(xs));
}
__device__ __forceinline__ float sigmoid_clamped(float x) {  // sigmoid(x) * (1 + 2*0.001) - 0.001
    return fmaf(__builtin_amdgcn_rcpf(1.f + fast_exp(-x)), 1.002f, -0.001f);
}

// Depth of coarse sample i on ray g (renderer.py:169-192), fp32 with the same operation order.
__device__ __forceinline__ float coarse_depth(const RenderArgs& a, int g, int i, float u)
{
#pragma clang fp contract(off)
    const int S = a.Sc;
    if (a.t_start) {                                      // tensor limits: math_utils.linspace (math_utils.py:101-118)
        const float s = a.t_start[g], e = a.t_end[g];
        const float t = (float)i / (float)(S - 1);
        const float z = s + t * (e - s);
        return z + u * ((e - s) / (float)(S - 1));
    }
    if (a.disparity) {
        const float step = 1.f / (float)(S - 1);
        float t = (i < S / 2) ? 0.f + step * (float)i : 1.f - step * (float)(S - i - 1);
        t = t + u * step;
        return 1.f / (1.f / a.ray_start * (1.f - t) + 1.f / a.ray_end * t);
    }
    const float step = a.lin_step;                                          // (ray_end - ray_start) / (S - 1) in fp32, as torch.linspace
    const float lin = (i < S / 2) ? a.ray_start + step * (float)i : a.ray_end - step * (float)(S - i - 1);
    return lin + u * step;
}

// ---- tri-plane gather: mean over planes of the bilinear sample, channels [16h, 16h+16) ------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
struct PlaneTaps { unsigned o00, o10, o01, o11; float w00, w10, w01, w11; };

// Byte offsets and weights of the four bilinear taps of plane p (grid_sample, align_corners=False, zero padding).
__device__ __forceinline__ PlaneTaps plane_taps(const RenderArgs& a, unsigned img_off, int h, int p, float px, float py, float pz)
{
    const int W = a.W, H = a.H;
    // inverse plane bases (renderer.py:23-53): plane 0 -> (x, y), plane 1 -> (x, z), plane 2 -> (z, x)
    const float gx = (p == 2) ? pz : px;
    const float gy = (p == 0) ? py : (p == 1 ? pz : px);
    // pixel = ((g + 1) * size - 1) / 2
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);       // keeps the int conversion sane; all taps of a
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);       // clamped coordinate are out of range anyway
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
    const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y0 + 1 >= 0) & (y0 + 1 < H);
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
    PlaneTaps t;
    t.w00 = (vx0 & vy0) ? wx0 * wy0 : 0.f;    // nw
    t.w10 = (vx1 & vy0) ? wx1 * wy0 : 0.f;    // ne
    t.w01 = (vx0 & vy1) ? wx0 * wy1 : 0.f;    // sw
    t.w11 = (vx1 & vy1) ? wx1 * wy1 : 0.f;    // se
    // 32-bit byte offsets into one buffer resource (wave-uniform descriptor): no 64-bit address arithmetic per tap
    const unsigned pbase = img_off + (unsigned)p * a.plane_bytes + (unsigned)h * 64u;
    t.o00 = pbase + __umul24(__umul24(cy0, W) + cx0, a.pix_bytes);
    t.o10 = pbase + __umul24(__umul24(cy0, W) + cx1, a.pix_bytes);
    t.o01 = pbase + __umul24(__umul24(cy1, W) + cx0, a.pix_bytes);
    t.o11 = pbase + __umul24(__umul24(cy1, W) + cx1, a.pix_bytes);
    return t;
}
// One ROW of a plane's 2 x 2 footprint (two taps x 64 bytes per lane): 8 buffer_load_b128 into v, and its share of the blend.
__device__ __forceinline__ void issue_row(rsrc_t rsrc, unsigned o_left, unsigned o_right, f32x4 (&v)[8])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q]     = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_left + 16 * q, 0, 0));
        v[4 + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_right + 16 * q, 0, 0));
    }
}
__device__ __forceinline__ void blend_row(const f32x4 (&v)[8], float w_left, float w_right, float (&acc)[16])
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)                    // FMAs straight into the three-plane sum
            acc[q * 4 + e] = fmaf(v[4 + q][e], w_right, fmaf(v[q][e], w_left, acc[q * 4 + e]));
}

// PIPE = false: one plane at a time (16 loads in flight per lane, then its blend) — three exposed memory latencies per sample, 64 staging
// registers; what the training kernels use, their register budget is spent on gradients.  PIPE = true (inference): the six footprint rows
// of a sample go through THREE 8-load buffers, the next row's loads being issued as soon as a buffer has been blended, so roughly one
// latency is exposed per sample for 96 staging registers (two whole planes in flight need 128 and spill).
template <bool PIPE>
__device__ __forceinline__ void gather_features(const RenderArgs& a, rsrc_t rsrc, unsigned img_off, int h,
                                                float px, float py, float pz, float (&feat)[16])
{
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    if (!PIPE) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const PlaneTaps t = plane_taps(a, img_off, h, p, px, py, pz);
            f32x4 top[8], bot[8];
            issue_row(rsrc, t.o00, t.o10, top);
            issue_row(rsrc, t.o01, t.o11, bot);
            blend_row(top, t.w00, t.w10, acc);
            blend_row(bot, t.w01, t.w11, acc);
        }
    } else {
        const PlaneTaps t0 = plane_taps(a, img_off, h, 0, px, py, pz);
        const PlaneTaps t1 = plane_taps(a, img_off, h, 1, px, py, pz);
        f32x4 b0[8], b1[8], b2[8];
        issue_row(rsrc, t0.o00, t0.o10, b0);
        issue_row(rsrc, t0.o01, t0.o11, b1);
        issue_row(rsrc, t1.o00, t1.o10, b2);
        __builtin_amdgcn_sched_barrier(0);
        blend_row(b0, t0.w00, t0.w10, acc);
        __builtin_amdgcn_sched_barrier(0);
        issue_row(rsrc, t1.o01, t1.o11, b0);
        const PlaneTaps t2 = plane_taps(a, img_off, h, 2, px, py, pz);       // under the loads in flight; keeps 8 registers free until here
        __builtin_amdgcn_sched_barrier(0);
        blend_row(b1, t0.w01, t0.w11, acc);
        __builtin_amdgcn_sched_barrier(0);
        issue_row(rsrc, t2.o00, t2.o10, b1);
        __builtin_amdgcn_sched_barrier(0);
        blend_row(b2, t1.w00, t1.w10, acc);
        __builtin_amdgcn_sched_barrier(0);
        issue_row(rsrc, t2.o01, t2.o11, b2);
        __builtin_amdgcn_sched_barrier(0);
        blend_row(b0, t1.w01, t1.w11, acc);
        blend_row(b1, t2.w00, t2.w10, acc);
        blend_row(b2, t2.w01, t2.w11, acc);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) feat[c] = acc[c] * (1.f / 3.f);
}

// ---- cooperative gather (inference kernels) ---------------------------------------------------------
// gather_features has lane (j, h) fetch half a texel (64 B) of ITS ray in four 16-byte loads: every load instruction of the wave
// touches 32 texels and uses 32 bytes of each 128-byte line, and the four instructions of a tap walk the same 32 lines four times.
// The texture path pays per line, so the kernel ran at the L1's request rate, not at anything the memory behind it could give.
// Here EIGHT lanes share a texel: in group i (0..3) lane l fetches the 16-byte chunk l & 7 of the taps of wave ray 8 i + (l >> 3),
// so one load instruction reads 8 whole lines and a tap of the wave's 32 rays is 4 instructions x 8 lines instead of 4 x 32.
// The taps (offsets, weights) of the wave's 32 rays x 3 planes are computed ONCE per step by the rays' owner lanes and passed through
// a 3 KB LDS table (a first version had every lane recompute the taps of its four rays: 8x the arithmetic, and the kernel went from
// line-rate-bound to VALU-bound).  Each lane blends its 4 channels in the same order as gather_features (bit-identical sums) and
// drops them into a per-wave [32 rays][36] LDS tile, from which lane (j, h) picks up the 16 channels the decoder wants.  Groups are
// pipelined two deep: 24 loads in flight per lane, as before.
// Tap table of one sample step: [ray][plane][4 byte offsets | 4 weights] = 96 B per ray.  Rays 16..31 live in the wave's `ttile`; rays 0..15 in
// the bytes of ftile rows 16..31, which the step does not write before groups 0 and 1 (the only readers of those entries) have issued.
constexpr int kTapRow = 24;                      // floats per ray
constexpr int kTapTile = 16 * kTapRow;           // per wave, beside ftile
constexpr int kRaysB = 4;                        // rays that go through the importance sampler together (independent chains; 2 -> 4: -1.3 %, profiles/round3_q_*)
__device__ __forceinline__ float* tap_entry(float* ftile, float* ttile, int ray)
{
    return ray < 16 ? ftile + 16 * kFeatPitch + ray * kTapRow : ttile + (ray - 16) * kTapRow;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put_taps(float* e, const PlaneTaps& t)
{
    *(u32x4*)e = u32x4{t.o00, t.o10, t.o01, t.o11};
    *(f32x4*)(e + 4) = f32x4{t.w00, t.w10, t.w01, t.w11};
}
__device__ __forceinline__ void coop_issue(rsrc_t rsrc, const float* e, unsigned chunk_off, f32x4 (&buf)[12], float (&w)[12])
{
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const u32x4 o = *(const u32x4*)(e + 8 * p);
        const f32x4 wv = *(const f32x4*)(e + 8 * p + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            buf[4 * p + t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o[t] + chunk_off, 0, 0));
            w[4 * p + t] = wv[t];
        }
    }
}
// TRANSPOSED: the tile is [channel][ray] (the backward kernel's T_f, which its weight-gradient MFMAs read) instead of [ray][channel]
template <bool TRANSPOSED>
__device__ __forceinline__ void coop_blend(const f32x4 (&buf)[12], const float (&w)[12], float* dst)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 12; t += 2)                       // (left, right) of the top row, then of the bottom row, plane by plane: gather_features' order
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(buf[t + 1][e], w[t + 1], fmaf(buf[t][e], w[t], acc[e]));
    acc = acc * (1.f / 3.f);
    if (TRANSPOSED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e * kFeatPitch] = acc[e];
    } else
        *(f32x4*)dst = acc;
}
// (px, py, pz): the sample point of the lane's OWN ray (lane & 31); img: its image's byte offset.  The taps of the wave's 96 (ray, plane)
// pairs are worked out once — lane (j, h) does plane 2h, the h = 0 half plane 1 as well — and handed round through the tap table.
template <bool TRANSPOSED = false>
__device__ __forceinline__ void gather_features_coop(const RenderArgs& a, rsrc_t rsrc, unsigned img, int lane, float px, float py, float pz,
                                                     float* ftile, float* ttile, float (&feat)[16])
{
    const int sub = lane >> 3, chunk = lane & 7, j = lane & 31, h = lane >> 5;
    // Lanes exchange data through LDS here.  The hardware keeps a wave's LDS operations in order; wave_sync() is what tells the COMPILER
    // that other lanes read what this lane wrote (without it a build of the bf16x3 kernel hoisted the feature reads above the blends' stores).
    wave_sync();                                           // the previous step's feature reads are done: the table may overwrite rows 16..31
    {
        float* const e = tap_entry(ftile, ttile, j);
        put_taps(e + 16 * h, plane_taps(a, img, 0, 2 * h, px, py, pz));
        if (h == 0) put_taps(e + 8, plane_taps(a, img, 0, 1, px, py, pz));
    }
    wave_sync();
    f32x4 bufA[12], bufB[12];
    float wA[12], wB[12];
    constexpr int RS = TRANSPOSED ? 1 : kFeatPitch;        // tile stride of a ray / of a channel
    constexpr int CS = TRANSPOSED ? kFeatPitch : 1;
    float* const dst = ftile + sub * RS + chunk * 4 * CS;
    const unsigned c16 = (unsigned)chunk * 16u;
    const float* const e_lo = ftile + 16 * kFeatPitch + sub * kTapRow;     // rays sub, 8 + sub
    const float* const e_hi = ttile + sub * kTapRow;                       // rays 16 + sub, 24 + sub
    coop_issue(rsrc, e_lo, c16, bufA, wA);
    __builtin_amdgcn_sched_barrier(0);
    coop_issue(rsrc, e_lo + 8 * kTapRow, c16, bufB, wB);
    __builtin_amdgcn_sched_barrier(0);
    coop_blend<TRANSPOSED>(bufA, wA, dst);
    __builtin_amdgcn_sched_barrier(0);
    coop_issue(rsrc, e_hi, c16, bufA, wA);
    __builtin_amdgcn_sched_barrier(0);
    coop_blend<TRANSPOSED>(bufB, wB, dst + 8 * RS);
    __builtin_amdgcn_sched_barrier(0);
    coop_issue(rsrc, e_hi + 8 * kTapRow, c16, bufB, wB);
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();                                           // every lane has read its table entries of rays 0..15: rows 16..31 may take features
    coop_blend<TRANSPOSED>(bufA, wA, dst + 16 * RS);
    coop_blend<TRANSPOSED>(bufB, wB, dst + 24 * RS);
    wave_sync();
    if (TRANSPOSED) {
#pragma unroll
        for (int c = 0; c < 16; ++c) feat[c] = ftile[(h * 16 + c) * kFeatPitch + j];
    } else {
        const f32x4* src = (const f32x4*)(ftile + j * kFeatPitch + h * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = src[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) feat[q * 4 + e] = v[e];
        }
    }
}

// ---- decoder pieces -------------------------------------------------------------------------------
// Layer 1 of net `n` for this wave's 32 samples: returns the 64 hidden units (post-softplus) as two
// accumulator tiles; lane (j,h) holds hidden unit 32t + (r&3) + 8(r>>2) + 4h of sample j in tile[t][r].
template <bool LOG2 = false, bool WIDE = false>
__device__ __forceinline__ void mlp_layer1(const float* lds, int n, int lane, int h, const float (&feat)[16],
                                           f32x16& h0, f32x16& h1, const float* feat_hi = nullptr)
{
    const f32x4* b1 = (const f32x4*)(lds + OFF_B1 + (n * 2 + h) * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = b1[q], v1 = b1[4 + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[q * 4 + e] = v0[e]; h1[q * 4 + e] = v1[e]; }
    }
    const f32x4* wv = (const f32x4*)(lds + n * kNetStride) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a0 = wv[q * 64], a1 = wv[(4 + q) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], feat[q * 4 + e], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], feat[q * 4 + e], h1, 0, 0, 0);
        }
    }
    if (WIDE) {                                              // inputs 32..63 of a 64-input first layer (the colour net of the two-plane-set renderer)
        const f32x4* wx = (const f32x4*)(lds + OFF_W1X) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a0 = wx[q * 64], a1 = wx[(4 + q) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], feat_hi[q * 4 + e], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], feat_hi[q * 4 + e], h1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        h0[r] = LOG2 ? softplus20_log2(h0[r]) : softplus20(h0[r]);
        h1[r] = LOG2 ? softplus20_log2(h1[r]) : softplus20(h1[r]);
    }
}

// Layer 2 colour rows (decoder outputs 1..32) of net `n`: lane (j,h) gets channel (r&3)+8(r>>2)+4h in out[r].
__device__ __forceinline__ void mlp_layer2(const float* lds, int n, int lane, int h, const f32x16& h0, const f32x16& h1, f32x16& out)
{
    const f32x4* b2 = (const f32x4*)(lds + OFF_B2 + (n * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = b2[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[q * 4 + e] = v[e];
    }
    const f32x4* wv = (const f32x4*)(lds + n * kNetStride) + 8 * 64 + lane;      // steps 32..63
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 a = wv[q * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = q * 4 + e;
            const float b = (s < 16) ? h0[s] : h1[s - 16];
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b, out, 0, 0, 0);
        }
    }
}

// Density row (decoder output 0) of the density net from its hidden units; reduced over the two halves.
__device__ __forceinline__ float mlp_sigma(const float* lds, int h, const f32x16& h0, const f32x16& h1)
{
    const f32x4* w = (const f32x4*)(lds + OFF_W2S + h * 32);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = w[q], v1 = w[4 + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s = fmaf(v0[e], h0[q * 4 + e], s); s = fmaf(v1[e], h1[q * 4 + e], s); }
    }
    s += __shfl_xor(s, 32, 64);
    return s + lds[OFF_B2S];
}

// ---- the decoder as bf16x3 (inference) ---------------------------------------------------------------
// The f32-input MFMA runs at 1/16 of the bf16 rate and the decoder's 128 of them per sample are 38 % of this kernel's SIMD time (and its
// floor: 1.0 ms per launch).  Same split as the convolutions (csrc/conv2d.hip): x = hi + lo in bf16, x * w = xh*wh + xh*wl + xl*wh, fp32
// accumulation — 48 bf16 MFMAs of half the duration per sample.  The transposed formulation is kept: weights are the A operand (LDS,
// pre-split when the block loads its decoder copy), the lane's own registers are the B operand (features, then hidden units straight
// from the accumulator layout), split in registers.  v_mfma_f32_32x32x16_bf16: lane (j, h) supplies k = 8h + e, e = 0..7.
//   layer 1, tile t, k-step s: k (h, e) <-> input channel 16h + 8s + e            (the lane's feat[8s + e])
//   layer 2,         k-step s: k (h, e) <-> hidden unit 32(s>>1) + pi(8(s&1) + e) + 4h, pi(r) = (r&3) + 8(r>>2)   (the lane's h_{s>>1}[8(s&1) + e])
// LDS image per net (16 KB, the fp32 stream's size): [hi | lo] x [block 8][lane 64][8 bf16]; blocks 0..3 = layer 1 (t, s), 4..7 = layer 2 (s).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// HAZARD (measured on gfx950, ROCm 7.2 hipcc): an MFMA that reads, as SrcB, a register written by v_cvt_pk_bf16_f32 two wait states
// earlier — the distance the compiler's own hazard nops give — now and then sees the OLD value of one 16-lane group: 5 % of the rays of a
// render off by 1e-3, and whether a build shows it depends on how the scheduler happened to interleave conversions and MFMAs (one
// version of this kernel was clean, the next two were not, with identical decoder code).  The asm below ties every converted register to
// one point after the conversions and holds the wave there for five wait states before any MFMA can read them (a build with >= 5 wait
// states between conversion and MFMA measured clean; 2 did not; the threshold in between was not searched).
__device__ __forceinline__ void split8(const float* v, bf8& hi, bf8& lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e];
        const __bf16 hx = (__bf16)x;
        hi[e] = hx;
        lo[e] = (__bf16)(x - (float)hx);
    }
    asm volatile("s_nop 4" : "+v"(hi), "+v"(lo));
}
__device__ __forceinline__ f32x16 mfma3(const bf8& ah, const bf8& al, const bf8& bh, const bf8& bl, f32x16 c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
}
// hidden units of net n (post-softplus, log2 domain) from the split features fh / fl (k-steps 0, 1)
__device__ __forceinline__ void mlp_layer1_bf3(const float* lds, int n, int lane, int h, const bf8 (&fh)[2], const bf8 (&fl)[2], f32x16& h0, f32x16& h1)
{
    const f32x4* b1 = (const f32x4*)(lds + OFF_B1 + (n * 2 + h) * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = b1[q], v1 = b1[4 + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[q * 4 + e] = v0[e]; h1[q * 4 + e] = v1[e]; }
    }
    // ONE lane-dependent base (lane * 16 bytes); net, block and hi/lo are compile-time byte offsets that fold into ds_read_b128 immediates
    const char* wb = (const char*)lds + lane * 16 + n * (kNetStride * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                        // interleave the two tiles: consecutive MFMAs on different accumulators
        const bf8 a0h = *(const bf8*)(wb + (0 * 2 + s) * 1024), a0l = *(const bf8*)(wb + 8192 + (0 * 2 + s) * 1024);
        const bf8 a1h = *(const bf8*)(wb + (1 * 2 + s) * 1024), a1l = *(const bf8*)(wb + 8192 + (1 * 2 + s) * 1024);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, fh[s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, fh[s], h1, 0, 0, 0);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, fh[s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, fh[s], h1, 0, 0, 0);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, fl[s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, fl[s], h1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = softplus20_log2(h0[r]); h1[r] = softplus20_log2(h1[r]); }
}
// colour rows of net n from its hidden units
__device__ __forceinline__ void mlp_layer2_bf3(const float* lds, int n, int lane, int h, const f32x16& h0, const f32x16& h1, f32x16& out)
{
    const f32x4* b2 = (const f32x4*)(lds + OFF_B2 + (n * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = b2[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[q * 4 + e] = v[e];
    }
    const char* wb = (const char*)lds + lane * 16 + n * (kNetStride * 4) + 4 * 1024;
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                        // (one accumulator: a second one to break the dependence spills at 256 registers)
        float hv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (s < 2) ? h0[8 * (s & 1) + e] : h1[8 * (s & 1) + e];
        bf8 xh, xl;
        split8(hv, xh, xl);
        const bf8 ah = *(const bf8*)(wb + s * 1024), al = *(const bf8*)(wb + 8192 + s * 1024);
        out = mfma3(ah, al, xh, xl, out);
    }
}

// ---- layer 1 as bf16x6 (fp32-accurate; training forward and the exact legs) ---------------------------------------------------------------
// The exact decoder's 128 f32-input MFMAs per sample are half of the exact launch (they do not hide vector work).  Layer 1 reads the gathered features:
// split ONCE per sample into three bf16 pieces (72 vector instructions, shared by both nets), against weights split once per work-group by the block prologue, it is
// 24 bf16 MFMAs of half the duration where the f32 form issues 32 — six terms per product (hh, hm, mh, hl, lh, mm), the error of one fp32 multiply-add.  Layer 2
// stays on the f32-input MFMA: its third weight image does not fit the 160 KB of LDS next to eight waves' tiles (layer 1's fits with 960 bytes to spare).
// Fragment order as in mlp_layer1_bf3: tile t, k-step s: k (h, e) <-> input channel 16 h + 8 s + e.
template <bool LOG2>
__device__ __forceinline__ void mlp_layer1_x6(const float* lds, int n, int lane, int h, const bf8 (&fh)[2], const bf8 (&fm)[2], const bf8 (&fl)[2], f32x16& h0, f32x16& h1)
{
    const f32x4* b1 = (const f32x4*)(lds + OFF_B1 + (n * 2 + h) * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = b1[q], v1 = b1[4 + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[q * 4 + e] = v0[e]; h1[q * 4 + e] = v1[e]; }
    }
    const char* wb = (const char*)lds + lane * 16 + n * (kNetStride * 4);              // hi at + block * 1024, mid 4 KB further
    const char* wl = (const char*)(lds + OFF_L1LO) + lane * 16 + n * 4096;             // lo
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bf8 a0h = *(const bf8*)(wb + (0 * 2 + s) * 1024), a0m = *(const bf8*)(wb + 4096 + (0 * 2 + s) * 1024), a0l = *(const bf8*)(wl + (0 * 2 + s) * 1024);
        const bf8 a1h = *(const bf8*)(wb + (1 * 2 + s) * 1024), a1m = *(const bf8*)(wb + 4096 + (1 * 2 + s) * 1024), a1l = *(const bf8*)(wl + (1 * 2 + s) * 1024);
#pragma unroll
        for (int term = 0; term < 6; ++term) {                                          // the two tiles alternate: consecutive MFMAs on different accumulators
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, a0h, a0m, a0l), P3D_X6_B(term, fh[s], fm[s], fl[s]), h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3D_X6_A(term, a1h, a1m, a1l), P3D_X6_B(term, fh[s], fm[s], fl[s]), h1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        h0[r] = LOG2 ? softplus20_log2(h0[r]) : softplus20(h0[r]);
        h1[r] = LOG2 ? softplus20_log2(h1[r]) : softplus20(h1[r]);
    }
}
__device__ __forceinline__ void split3_feat(const float (&feat)[16], bf8 (&fh)[2], bf8 (&fm)[2], bf8 (&fl)[2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 a0 = {feat[8 * s], feat[8 * s + 1], feat[8 * s + 2], feat[8 * s + 3]}, a1 = {feat[8 * s + 4], feat[8 * s + 5], feat[8 * s + 6], feat[8 * s + 7]};
        split3_bf16x8(a0, a1, fh[s], fm[s], fl[s]);
    }
}

// ---- importance sampling, wave-cooperative (renderer.py:194-253), NR rays at a time ------------------------------
// Per ray: lane i holds coarse weight w_i (i < Sc-1) and coarse depth z_i (i < Sc); lane j returns fine depth j (unsorted), +inf for j >= Sf.
// Everything stays in registers: lane k keeps pdf entry k, bin midpoint k and (after the scan) cdf entry k; the two sequential fp32 sums
// (normaliser, cdf — sequential so that the searchsorted indices are reproducible bit for bit) broadcast lane k's value with v_readlane.
// The whole phase is dependent-latency work (61-step scans, a 21-stage sort, divisions) with two waves per SIMD to hide it — 11 % of the
// kernel (profiles/round3_n_render_ablation.log) — so NR rays go through it TOGETHER: NR independent chains in every loop.
__device__ __forceinline__ float lane_bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
template <int NR>
__device__ __forceinline__ void importance_depth(int Sc, int Sf, int lane, const float (&w_i)[NR], const float (&z_i)[NR], const float (&u)[NR], float (&z)[NR],
                                                 int (*bins)[NR] = nullptr)          // bins (tests): lane j's searchsorted index, as torch.searchsorted(cdf, u, right=True) returns it
{
#pragma clang fp contract(off)
    const float ninf = -INFINITY;
    const int nw = Sc - 3;                                 // pdf entries = smoothed[1:-1]
    float pnum[NR], zmid[NR], total[NR], pdf[NR], cdf[NR], mycdf[NR];
    int inds[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const float wi  = (lane < Sc - 1) ? w_i[q] : ninf;
        float wl = __shfl_up(wi, 1, 64);  if (lane == 0) wl = ninf;
        const float mp = fmaxf(wl, wi);                    // max_pool1d(k=2, s=1, pad=1): Sc values
        const float mpn = __shfl_down(mp, 1, 64);
        const float ap = (mp + mpn) / 2.f;                 // avg_pool1d(k=2, s=1): Sc-1 values
        const float wk = (ap + 0.01f) + 1e-5f;             // "+ 0.01" then sample_pdf's "+ eps"
        const float zn = __shfl_down(z_i[q], 1, 64);
        zmid[q] = 0.5f * (z_i[q] + zn);                    // bins: Sc-1 midpoints, lane k holds bin k
        pnum[q] = __shfl_down(wk, 1, 64);                  // lane k: pdf numerator k = smoothed[k + 1]
        total[q] = 0.f;
    }
    for (int k = 0; k < nw; ++k)
#pragma unroll
        for (int q = 0; q < NR; ++q) total[q] = total[q] + lane_bcast(pnum[q], k);
    // cdf_0 = 0, cdf_{k+1} = cdf_k + pdf_k (Sc-2 entries); inds = #{cdf <= u} (searchsorted right=True; the cdf is non-decreasing)
#pragma unroll
    for (int q = 0; q < NR; ++q) { pdf[q] = pnum[q] / total[q]; cdf[q] = 0.f; mycdf[q] = 0.f; inds[q] = (0.f <= u[q]) ? 1 : 0; }
    for (int k = 0; k < nw; ++k)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            cdf[q] = cdf[q] + lane_bcast(pdf[q], k);
            inds[q] += (cdf[q] <= u[q]) ? 1 : 0;
            mycdf[q] = (lane == k + 1) ? cdf[q] : mycdf[q];    // lane k keeps cdf entry k
        }
    if (bins) {
#pragma unroll
        for (int q = 0; q < NR; ++q) (*bins)[q] = inds[q];
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int below = max(inds[q] - 1, 0), above = min(inds[q], nw);
        const float cb = __shfl(mycdf[q], below, 64), ca = __shfl(mycdf[q], above, 64);
        const float zb = __shfl(zmid[q], below, 64), za = __shfl(zmid[q], above, 64);
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u[q] - cb) / denom;
        const float zz = zb + t * (za - zb);
        z[q] = (lane < Sf) ? zz : INFINITY;
    }
}

// One step of unify_samples (renderer.py:157-167: sort of cat(coarse, fine) depths, both halves already ascending): take the coarse head unless the
// fine head is strictly smaller — the order a stable sort of [coarse, fine] gives.
__device__ __forceinline__ bool merge_takes_coarse(float zc, float zf) { return zc <= zf; }

template <int NR>
__device__ __forceinline__ void bitonic_sort64(float (&v)[NR], int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & j) == 0);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float o = __shfl_xor(v[q], j, 64);
                v[q] = (lower == up) ? fminf(v[q], o) : fmaxf(v[q], o);
            }
        }
    }
}

// ---- the fused kernel -------------------------------------------------------------------------------
// TAPE = false: the inference / forward kernel.  TAPE = true: the same sweep driven by dL/dfeat instead of writing feat: it records, per
// ray, what the point-wise backward kernel needs (sample depth, colour weight 0.5 (w[k-1] + w[k]), dL/dsigma_k) — see render_bwd.hip.
// DUAL (NNETS == 2, inference only): two plane sets.  The label net (density + labels) reads the SEMANTIC planes' features, the colour
// net reads cat(texture features, semantic features) through a 64-input first layer (renderer.py:324-333); everything else —
// sampling, merge, compositing over cat(colour, label) — is the same sweep.
constexpr int kWavesPerBlockDual = 4;            // the DUAL kernel keeps two feature vectors live: one wave per SIMD (512 registers) instead of spilling
template <int NNETS, bool TAPE, bool DUAL = false, bool BF3 = false, bool L1X6 = false>
__global__ void __launch_bounds__((DUAL ? kWavesPerBlockDual : kWavesPerBlock) * 64, DUAL ? 1 : 2)
render_forward_kernel(RenderArgs a)
{
    static_assert(!DUAL || (NNETS == 2 && !TAPE), "the two-plane-set variant is the two-net inference kernel");
    static_assert(!BF3 || (!TAPE && !DUAL), "the bf16x3 decoder is the one-plane-set inference kernel");
    static_assert(!L1X6 || (!BF3 && !DUAL && !TAPE), "layer 1 as bf16x6 is a form of the exact one-plane-set forward kernel");
    constexpr int kDecFloats = DUAL ? kDecoderFloatsDual : (L1X6 ? kDecoderFloatsL1X6 : kDecoderFloats);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // Inference works in the log2 domain (softplus20_log2): its LDS copy of the decoder has layer 1 (weights, biases) scaled by log2(e) and
    // everything that consumes the hidden units (layer-2 colour rows, the density row) by ln 2.  The packed stream in global memory is
    // the same for every consumer; the tape sweep (TAPE) and the backward kernel keep natural units, their gradients use the hidden values.
    constexpr bool LOG2 = !TAPE;
    {   // decoder stream -> LDS (16 B per lane)
        const f32x4* src = (const f32x4*)a.decoder;
        f32x4* dst = (f32x4*)lds;
        constexpr int kSrcFloats = L1X6 ? kDecoderFloats : kDecFloats;         // (the lo images of L1X6 are made here, not streamed)
        for (int i = tid; i < kSrcFloats / 4; i += blockDim.x) {
            f32x4 v = src[i];
            if (LOG2) {
                const int f = i * 4;
                const bool first  = f >= OFF_W1X ? true : (f < OFF_B1 ? (f & (kNetStride - 1)) < kNetStride / 2 : f < OFF_B2);      // layer-1 weights / biases
                const bool second = f < OFF_B1 ? !first : (f >= OFF_W2S && f < OFF_B2S);                    // layer-2 colour rows / density row
                const float sc = first ? 1.4426950408889634f : (second ? 0.6931471805599453f : 1.f);
                v = v * sc;
            }
            if (BF3 && i * 4 < 2 * kNetStride) {                  // weights: the stream is [net][block][lane][8] floats (p3d_pack_decoder_bf16x3) -> [hi | lo] bf16
                const int f = i * 4, n = f / kNetStride, e = f - n * kNetStride;          // e: float index inside the net, a multiple of 4
                __bf16* hi = (__bf16*)(lds + n * kNetStride) + e;                       // same [block][lane][8] position, 2-byte elements
                __bf16* lo = hi + kNetStride;                                           // lo image: 8 KB further
#pragma unroll
                for (int c = 0; c < 4; ++c) { const __bf16 hx = (__bf16)v[c]; hi[c] = hx; lo[c] = (__bf16)(v[c] - (float)hx); }
            } else if (L1X6 && i * 4 < 2 * kNetStride && ((i * 4) & (kNetStride - 1)) < kNetStride / 2) {      // layer-1 weights ([net][block 4][lane][8] floats): three bf16 images
                const int f = i * 4, n = f / kNetStride, e = f - n * kNetStride;
                __bf16* hi = (__bf16*)(lds + n * kNetStride) + e;                       // [block][lane][8], 2-byte elements: 4 KB
                __bf16* mid = hi + kNetStride / 2;                                      // 4 KB further (still inside the net's layer-1 half)
                __bf16* lo = (__bf16*)(lds + OFF_L1LO + n * 1024) + e;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const __bf16 hx = (__bf16)v[c];
                    const float r1 = v[c] - (float)hx;
                    const __bf16 mx = (__bf16)r1;
                    hi[c] = hx; mid[c] = mx; lo[c] = (__bf16)(r1 - (float)mx);
                }
            } else
            dst[i] = v;
        }
    }
    __syncthreads();

    float* tile = lds + kDecFloats + wave * kWaveTile;          // [sample][kPitch] coarse weights, then fine depths
    const int SN = NNETS - 1;                                    // density comes from the last net (triplane_cond.py:958)
    const int Sc = a.Sc, Sf = a.Sf;

    // ---- ray -> (workgroup, wave, lane) assignment -------------------------------------------------------------
    // Rays of one pixel COLUMN project onto the same texels of the (x,z) plane, rays of one pixel ROW onto the same
    // texels of the (z,y) plane.  Workgroup b runs on XCD b % 8 (private L2), so when the image is R x R with
    // R = 16 * ns and ns | 8, XCD k is given the 16-pixel-wide column strip k % ns of every image: the (x,z) texels a
    // strip needs (~1 MB per image) stay in that XCD's L2 for all its rows instead of being re-fetched by 8 L2s.
    // A workgroup is a 16 x 16 pixel block, a wave two 16-pixel rows of it (same row => same (z,y) texels: one fetch
    // serves 16 lanes).  Any other shape falls back to consecutive rays.
    const int wpb = (int)blockDim.x >> 6;                        // waves per block: 8, or fewer for small launches (host: p3d_render_forward)
    int ray0 = (blockIdx.x * wpb + wave) * 32;                   // linear assignment (and the bound for 'live')
    int g_lane = ray0 + j;
    {
        const int R = a.res;
        const int ns = R >> 4, rpb = 2 * wpb;                     // 16-pixel column strips; pixel rows per block
        if (R > 0 && (R & 15) == 0 && (R % rpb) == 0 && ns <= 8 && (8 % ns) == 0 && a.rays_per_img == R * R && ((a.total_rays / (wpb * 32)) & 7) == 0) {
            const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
            const int strip = xcd % ns, sub = xcd / ns, per = 8 / ns;         // `per` XCDs share a strip
            const int blk = slot * per + sub;                                 // (image, row block) index within the strip
            const int nrb = R / rpb;                                          // row blocks per image
            const int n_i = blk / nrb, rb = blk - n_i * nrb;
            // a full block (8 waves, 16 x 16 pixels): wave w = 8 rows x 4 columns, ray j = (row j & 7, column j >> 3), so the eight rays of a gather
            // group (8 i .. 8 i + 7) are ONE pixel column: their (x,z)- and (z,x)-plane taps are the same one or two texel lines per load
            // instruction instead of eight (round 3, same box: 1.302 vs 1.338 ms).  Smaller blocks keep two 16-pixel rows per wave.
            const int row = (wpb == 8) ? rb * rpb + (wave & 1) * 8 + (j & 7) : rb * rpb + wave * 2 + (j >> 4);
            const int col = (wpb == 8) ? strip * 16 + (wave >> 1) * 4 + (j >> 3) : strip * 16 + (j & 15);
            g_lane = n_i * a.rays_per_img + row * R + col;
            ray0 = 0;                                                         // every lane is a real ray in this mode
        }
    }
    if (ray0 >= a.total_rays) return;
    const int g = min(g_lane, a.total_rays - 1);                 // tail lanes shadow the last ray, never store
    const bool live = g_lane < a.total_rays;
    const int n_img = g / a.rays_per_img;
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, 0, a.planes_total_bytes, 0x00020000);
    const rsrc_t rsrc_sem = DUAL ? __builtin_amdgcn_make_buffer_rsrc((void*)a.planes2, 0, a.planes_total_bytes, 0x00020000) : rsrc;   // density / labels read these
    const unsigned img = (unsigned)n_img * a.img_bytes;
    const float ox = a.ray_o[g * 3 + 0], oy = a.ray_o[g * 3 + 1], oz = a.ray_o[g * 3 + 2];
    const float dx = a.ray_d[g * 3 + 0], dy = a.ray_d[g * 3 + 1], dz = a.ray_d[g * 3 + 2];
    const float cs = a.coord_scale;
    const float* uc = a.u_coarse + (size_t)g * Sc;
    constexpr bool COOP = !DUAL;                                  // one plane set: eight lanes to a texel
    float* const ftile = lds + kDecFloats + wpb * kWaveTile + wave * kFeatTile;
    float* const ttile = lds + kDecFloats + wpb * (kWaveTile + kFeatTile) + wave * kTapTile;

    // ------------------------------ phase A: coarse densities -> weights ------------------------------
    {
        float T = 1.f, z_prev = 0.f, s_prev = 0.f;
        for (int i = 0; i < Sc; ++i) {
            const float z = coarse_depth(a, g, i, uc[i]);
            float feat[16];
            if constexpr (COOP) gather_features_coop(a, rsrc_sem, img, lane, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), ftile, ttile, feat);
            else gather_features<true>(a, rsrc_sem, img, h, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), feat);
            f32x16 h0, h1;
            if constexpr (BF3) {
                bf8 fh[2], fl[2];
                split8(feat, fh[0], fl[0]); split8(feat + 8, fh[1], fl[1]);
                mlp_layer1_bf3(lds, SN, lane, h, fh, fl, h0, h1);
            } else if constexpr (L1X6) {
                bf8 fh[2], fm[2], fl[2];
                split3_feat(feat, fh, fm, fl);
                mlp_layer1_x6<LOG2>(lds, SN, lane, h, fh, fm, fl, h0, h1);
            } else
            mlp_layer1<LOG2>(lds, SN, lane, h, feat, h0, h1);
            const float sigma = mlp_sigma(lds, h, h0, h1);
            if (i > 0) {
                const float dens = softplus20(0.5f * (s_prev + sigma) - 1.f);
                const float alpha = 1.f - fast_exp(-dens * (z - z_prev));
                const float w = alpha * T;
                T *= (1.f - alpha + 1e-10f);
                if (h == 0) tile[(i - 1) * kPitch + j] = w;
            }
            z_prev = z; s_prev = sigma;
        }
    }
    wave_sync();

    // ------------------------------ phase B: importance depths, sorted (kRaysB rays at a time) ---------
    for (int r = 0; r < 32; r += kRaysB) {
        int gr[kRaysB]; bool r_live[kRaysB];
        float w_i[kRaysB], z_i[kRaysB], u[kRaysB], zf[kRaysB];
#pragma unroll
        for (int q = 0; q < kRaysB; ++q) {
            gr[q] = __shfl(g, r + q, 64);                                        // global index of the wave's ray (wave-uniform)
            r_live[q] = __shfl((int)live, r + q, 64) != 0;
            w_i[q] = (lane < Sc - 1) ? tile[lane * kPitch + r + q] : 0.f;
            z_i[q] = (lane < Sc) ? coarse_depth(a, gr[q], lane, a.u_coarse[(size_t)gr[q] * Sc + lane]) : 0.f;
            u[q]   = (lane < Sf) ? a.u_fine[(size_t)gr[q] * Sf + lane] : 2.f;
            if (a.dbg_wcoarse && lane < Sc - 1 && r_live[q]) a.dbg_wcoarse[(size_t)gr[q] * (Sc - 1) + lane] = w_i[q];
        }
        int bins[kRaysB];
        importance_depth<kRaysB>(Sc, Sf, lane, w_i, z_i, u, zf, &bins);
        if (a.dbg_bins) {
#pragma unroll
            for (int q = 0; q < kRaysB; ++q)
                if (lane < Sf && r_live[q]) a.dbg_bins[(size_t)gr[q] * Sf + lane] = bins[q];
        }
        bitonic_sort64<kRaysB>(zf, lane);
        wave_sync();
#pragma unroll
        for (int q = 0; q < kRaysB; ++q) {
            if (lane < Sf) tile[lane * kPitch + r + q] = zf[q];
            if (a.dbg_fine && lane < Sf && r_live[q]) a.dbg_fine[(size_t)gr[q] * Sf + lane] = zf[q];
        }
    }
    wave_sync();

    // ------------------------------ phase C: merged decode + composite --------------------------------
    float acc[NNETS][16], prev[NNETS][16];       // TAPE: acc holds dL/dC (= 2 dL/dfeat) of this lane's 16 channels per net
#pragma unroll
    for (int n = 0; n < NNETS; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[n][r] = TAPE ? 2.f * a.g_feat[(size_t)g * (NNETS * 32) + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
            prev[n][r] = 0.f;
        }
    float4* const tape_i = TAPE ? (float4*)a.tape_i + (size_t)g * (Sc + Sf - 1) : nullptr;
    float4* const tape_s = TAPE ? (float4*)a.tape_s + (size_t)g * (Sc + Sf) : nullptr;
    float T = 1.f, z_prev = 0.f, s_prev = 0.f, w_sum = 0.f, wz_sum = 0.f, z_first = 0.f;
    int ic = 0, jf = 0;
    float zc = coarse_depth(a, g, 0, uc[0]);
    float zf = (Sf > 0) ? tile[j] : INFINITY;
    const int S = Sc + Sf;
    for (int k = 0; k < S; ++k) {
        const bool take_c = merge_takes_coarse(zc, zf);
        const float z = take_c ? zc : zf;
        if (take_c) { ++ic; zc = (ic < Sc) ? coarse_depth(a, g, ic, uc[ic]) : INFINITY; }
        else        { ++jf; zf = (jf < Sf) ? tile[jf * kPitch + j] : INFINITY; }

        float feat[16], feat_tex[16];
        if constexpr (COOP) gather_features_coop(a, rsrc_sem, img, lane, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), ftile, ttile, feat);
        else gather_features<false>(a, rsrc_sem, img, h, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), feat);
        if (DUAL) gather_features<false>(a, rsrc, img, h, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), feat_tex);
        // The density net goes first: its sigma closes interval k-1 (weight w), after which every net's
        // colours are folded into the accumulators as soon as its layer 2 retires — only `prev` (the
        // other end of the midpoint rule) stays live across samples.
        float sigma = 0.f, hw = 0.f;
        float t_alpha = 0.f, t_T = 0.f, t_sm = 0.f, t_A = 0.f;               // TAPE: record of interval k-1
        bf8 fh[2], fl[2];                                                    // BF3: the sample's features, split once for both nets
        [[maybe_unused]] bf8 fm[2];                                          // L1X6: three pieces
        if constexpr (BF3) { split8(feat, fh[0], fl[0]); split8(feat + 8, fh[1], fl[1]); }
        if constexpr (L1X6) split3_feat(feat, fh, fm, fl);
#pragma unroll
        for (int idx = 0; idx < NNETS; ++idx) {
            const int n = (idx == 0) ? SN : idx - 1;
            f32x16 h0, h1, o;
            if constexpr (BF3)  mlp_layer1_bf3(lds, n, lane, h, fh, fl, h0, h1);
            else if constexpr (L1X6) mlp_layer1_x6<LOG2>(lds, n, lane, h, fh, fm, fl, h0, h1);
            else if (DUAL && n == 0) mlp_layer1<LOG2, true>(lds, n, lane, h, feat_tex, h0, h1, feat);     // colour net: cat(texture, semantic)
            else                mlp_layer1<LOG2>(lds, n, lane, h, feat, h0, h1);
            if (idx == 0) {
                sigma = mlp_sigma(lds, h, h0, h1);
                if (k == 0) z_first = z;
                else {
                    const float dens = softplus20(0.5f * (s_prev + sigma) - 1.f);
                    const float alpha = 1.f - fast_exp(-dens * (z - z_prev));
                    const float w = alpha * T;
                    if (TAPE) { t_alpha = alpha; t_T = T; t_sm = 0.5f * (s_prev + sigma); }
                    T *= (1.f - alpha + 1e-10f);
                    hw = 0.5f * w;
                    w_sum += w;
                    wz_sum = fmaf(w, 0.5f * (z_prev + z), wz_sum);
                }
            }
            if constexpr (BF3) mlp_layer2_bf3(lds, n, lane, h, h0, h1, o);
            else               mlp_layer2(lds, n, lane, h, h0, h1, o);
            const bool squash = (n == 0) || (NNETS == 1) || a.sem_sigmoid;     // raw logits for the label net (triplane_cond.py:960-964)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float c = squash ? sigmoid_clamped(o[r]) : o[r];
                if (TAPE) t_A = fmaf(acc[n][r], prev[n][r] + c, t_A);          // dL/dw of interval k-1, colour part: sum_ch dC (c[k-1] + c[k]) / 2
                else      acc[n][r] = fmaf(hw, prev[n][r] + c, acc[n][r]);     // hw == 0 for the first sample
                prev[n][r] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAPE) {
            t_A += __shfl_xor(t_A, 32, 64);
            if (h == 0 && live) {
                tape_s[k] = make_float4(z, 0.f, 0.f, 0.f);
                if (k > 0) tape_i[k - 1] = make_float4(t_alpha, t_T, 0.5f * t_A, t_sm);
            }
        }
        z_prev = z; s_prev = sigma;
    }
    if (TAPE) {
        // ---- backward of the compositing (ray_marcher.py:25-57), one ray per lane, walking the intervals back to front:
        //   w_i = alpha_i T_i,  T_i = prod_{j<i} (1 - alpha_j + 1e-10)  =>  dL/dalpha_i = dw_i T_i - (sum_{j>i} dw_j w_j) / (1 - alpha_i + 1e-10)
        //   alpha = 1 - exp(-dens delta), dens = softplus(sigma_mid - 1)  =>  dL/dsigma_mid = dL/dalpha * delta (1 - alpha) sigmoid(sigma_mid - 1)
        // and hands every SAMPLE its share: dL/dsigma_k = (dsm[k-1] + dsm[k]) / 2, colour weight (w[k-1] + w[k]) / 2.
        float sum_dc = 0.f;
#pragma unroll
        for (int n = 0; n < NNETS; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum_dc += acc[n][r];
        sum_dc += __shfl_xor(sum_dc, 32, 64);
        if (h == 0 && live) {
            const float base = (a.g_wsum ? a.g_wsum[g] : 0.f) - (a.white_back ? sum_dc : 0.f);
            float suffix = 0.f, dsm_next = 0.f, w_next = 0.f;
            float z_hi = tape_s[S - 1].x;
            for (int i = S - 2; i >= 0; --i) {
                const float4 rec = tape_i[i];
                const float z_lo = tape_s[i].x;
                const float alpha = rec.x, Ti = rec.y, w = alpha * Ti;
                const float dw = rec.z + base;
                const float dalpha = dw * Ti - suffix / (1.f - alpha + 1e-10f);
                suffix = fmaf(dw, w, suffix);
                const float x = rec.w - 1.f;
                const float sg = x > 20.f ? 1.f : __builtin_amdgcn_rcpf(1.f + fast_exp(-x));
                const float dsm = dalpha * (z_hi - z_lo) * (1.f - alpha) * sg;
                tape_s[i + 1] = make_float4(z_hi, 0.5f * (w + w_next), 0.5f * (dsm + dsm_next), 0.f);
                dsm_next = dsm; w_next = w; z_hi = z_lo;
            }
            tape_s[0] = make_float4(z_hi, 0.5f * w_next, 0.5f * dsm_next, 0.f);
        }
        return;
    }

    // ------------------------------ epilogue --------------------------------------------------------
    const float bg = a.white_back ? (1.f - w_sum) : 0.f;
    if (live) {
        float* dst = a.feat + (size_t)g * (NNETS * 32);
#pragma unroll
        for (int n = 0; n < NNETS; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {                 // accumulator rows 8q + 4h + {0..3}: one 16-B store
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (acc[n][q * 4 + e] + bg) * 2.f - 1.f;
                *(f32x4*)(dst + n * 32 + q * 8 + h * 4) = v;
            }
        if (h == 0) {
            a.depth[g] = wz_sum / w_sum;                  // NaN when nothing was hit; finished by the clamp pass
            a.wsum[g] = w_sum;
        }
    }
    // global depth range (ray_marcher.py:50 clamps to min/max over the WHOLE depth tensor)
    float zmin = live ? z_first : INFINITY, zmax = live ? z_prev : -INFINITY;
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) { zmin = fminf(zmin, __shfl_xor(zmin, s, 64)); zmax = fmaxf(zmax, __shfl_xor(zmax, s, 64)); }
    if (lane == 0) { atomicMin(a.minmax, order_key(zmin)); atomicMax(a.minmax + 1, order_key(zmax)); }
}


} // namespace p3d
