// Fused tri-plane ray-marcher for gfx950.
//
// One kernel does what training/volumetric_rendering/renderer.py:88-140 (ImportanceRenderer.forward)
// spreads over ~120 tensor ops: stratified depths (:169-192) -> tri-plane bilinear taps (:55-65) ->
// OSG decoder MLP(s) (training/triplane.py:112-135, training/triplane_cond.py:926-970) -> midpoint
// compositing (ray_marcher.py:25-57) -> importance resampling (:194-253) -> merge of coarse+fine
// depths (:157-167) -> second decode -> final composite.  Nothing per-sample ever goes to HBM.
//
// Mapping (wave64, v_mfma_f32_32x32x2_f32):
//   * a wave owns 32 rays; lane = (ray j = lane&31, half h = lane>>5).
//   * planes are read channels-last ([N][3][H][W][32], one 128-B line per texel); lane (j,h) fetches
//     channels [16h,16h+16) of each of the 12 taps as 4 x 16-B loads, lanes j / j+32 complete a line.
//   * the MLP runs transposed, H^T = W1 * X^T and O^T = W2 * H^T: the weights are the MFMA A operand
//     (streamed from LDS, pre-permuted per lane), the samples are the B operand, so lane (j,h) feeds
//     B[k][j] straight from the registers that hold its 16 channels, receives its ray's hidden units
//     in the accumulator, applies softplus in place and feeds them to layer 2 as B again: no
//     cross-lane traffic between gather, layer 1, layer 2 and compositing.  f32-in MFMA is exact
//     fp32 (k-ordered fma chain), so parity with the fp32 reference is rounding-order only.
//   * compositing state (transmittance, accumulated colour, previous sample) lives in registers;
//     per-ray coarse weights / fine depths sit in a 33-float-pitch LDS tile per wave.
//   * the coarse pass only needs densities, so it runs layer 1 of the density net alone (1/4 of the
//     MLP); the final pass decodes all S_c+S_f merged samples in depth order with a two-pointer
//     merge, which replaces the reference's sort + 64-channel gather.
//   * importance sampling is wave-cooperative per ray (lane = bin / fine sample): sequential fp32
//     pdf/cdf (contraction off) so the searchsorted indices are reproducible bit for bit, then a
//     64-lane bitonic sort of the fine depths.
#include "p3d_common.h"
#include <math.h>

namespace p3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4  __attribute__((ext_vector_type(4)));

// ---- decoder stream layout (floats); identical in the packed global buffer and in LDS ------------
constexpr int kNetStride = 4096;                 // per net: 64 MFMA steps x 64 lanes
constexpr int OFF_B1   = 2 * kNetStride;         // [net][half][32]  hidden biases in accumulator order
constexpr int OFF_B2   = OFF_B1 + 128;           // [net][half][16]  colour biases in accumulator order
constexpr int OFF_W2S  = OFF_B2 + 64;            // [half][32]       density row of the density net
constexpr int OFF_B2S  = OFF_W2S + 64;           // [1]              density bias
constexpr int kDecoderFloats = 8464;             // padded to 16 floats
constexpr int kPitch = 33;                       // LDS pitch of the per-wave [sample][ray] tile
constexpr int kMaxS = 64;                        // max coarse / fine samples per ray
constexpr int kWaveTile = kMaxS * kPitch + 128;  // + two 64-float scratch rows
constexpr int kWavesPerBlock = 8;

struct RenderArgs {
    const float* planes;      // [N][3][H][W][32]
    const float* decoder;     // kDecoderFloats, see p3d_pack_decoder
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
__device__ __forceinline__ void gather_features(const RenderArgs& a, rsrc_t rsrc, unsigned img_off, int h,
                                                float px, float py, float pz, float (&feat)[16])
{
    const int W = a.W, H = a.H;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // inverse plane bases (renderer.py:23-53): plane 0 -> (x, y), plane 1 -> (x, z), plane 2 -> (z, x)
        const float gx = (p == 2) ? pz : px;
        const float gy = (p == 0) ? py : (p == 1 ? pz : px);
        // grid_sample(align_corners=False): pixel = ((g + 1) * size - 1) / 2, zero padding
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
        const float w00 = (vx0 & vy0) ? wx0 * wy0 : 0.f;    // nw
        const float w10 = (vx1 & vy0) ? wx1 * wy0 : 0.f;    // ne
        const float w01 = (vx0 & vy1) ? wx0 * wy1 : 0.f;    // sw
        const float w11 = (vx1 & vy1) ? wx1 * wy1 : 0.f;    // se
        // 32-bit byte offsets into one buffer resource (wave-uniform descriptor): no 64-bit address arithmetic per tap
        const unsigned pbase = img_off + (unsigned)p * a.plane_bytes + (unsigned)h * 64u;
        const unsigned o00 = pbase + __umul24(__umul24(cy0, W) + cx0, a.pix_bytes);
        const unsigned o10 = pbase + __umul24(__umul24(cy0, W) + cx1, a.pix_bytes);
        const unsigned o01 = pbase + __umul24(__umul24(cy1, W) + cx0, a.pix_bytes);
        const unsigned o11 = pbase + __umul24(__umul24(cy1, W) + cx1, a.pix_bytes);
        f32x4 v00[4], v10[4], v01[4], v11[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v00[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o00 + 16 * q, 0, 0));
            v10[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o10 + 16 * q, 0, 0));
            v01[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o01 + 16 * q, 0, 0));
            v11[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o11 + 16 * q, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = v00[q][e] * w00;
                s = fmaf(v10[q][e], w10, s);
                s = fmaf(v01[q][e], w01, s);
                s = fmaf(v11[q][e], w11, s);
                acc[q * 4 + e] += s;
            }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) feat[c] = acc[c] * (1.f / 3.f);
}

// ---- decoder pieces -------------------------------------------------------------------------------
// Layer 1 of net `n` for this wave's 32 samples: returns the 64 hidden units (post-softplus) as two
// accumulator tiles; lane (j,h) holds hidden unit 32t + (r&3) + 8(r>>2) + 4h of sample j in tile[t][r].
__device__ __forceinline__ void mlp_layer1(const float* lds, int n, int lane, int h, const float (&feat)[16],
                                           f32x16& h0, f32x16& h1)
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
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = softplus20(h0[r]); h1[r] = softplus20(h1[r]); }
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

// ---- importance sampling for one ray, wave-cooperative (renderer.py:194-253) ----------------------
// lane i holds coarse weight w_i (i < Sc-1) and coarse depth z_i (i < Sc); lane j returns fine depth j
// (unsorted), +inf for j >= Sf.  sA / sB: two 64-float LDS scratch rows of this wave.
__device__ __forceinline__ float importance_depth(int Sc, int Sf, int lane, float w_i, float z_i, float u, float* sA, float* sB)
{
#pragma clang fp contract(off)
    const float ninf = -INFINITY;
    const float wi  = (lane < Sc - 1) ? w_i : ninf;
    float wl = __shfl_up(wi, 1, 64);  if (lane == 0) wl = ninf;
    const float mp = fmaxf(wl, wi);                        // max_pool1d(k=2, s=1, pad=1): Sc values
    const float mpn = __shfl_down(mp, 1, 64);
    const float ap = (mp + mpn) / 2.f;                     // avg_pool1d(k=2, s=1): Sc-1 values
    const float wk = (ap + 0.01f) + 1e-5f;                 // "+ 0.01" then sample_pdf's "+ eps"
    const float zn = __shfl_down(z_i, 1, 64);
    const float zmid = 0.5f * (z_i + zn);                  // bins: Sc-1 midpoints
    const int nw = Sc - 3;                                 // pdf entries = smoothed[1:-1]
    wave_sync();
    if (lane >= 1 && lane <= nw) sA[lane - 1] = wk;
    if (lane <= Sc - 2) sB[lane] = zmid;
    wave_sync();
    float total = 0.f;
    for (int k = 0; k < nw; ++k) total = total + sA[k];
    wave_sync();
    if (lane < nw) sA[lane] = sA[lane] / total;
    wave_sync();
    // cdf_0 = 0, cdf_{k+1} = cdf_k + pdf_k (Sc-2 entries); inds = #{cdf <= u} (searchsorted right=True)
    float cdf = 0.f, cb = 0.f, zb = sB[0], ca = 0.f, za = 0.f;
    bool found = false;
    for (int k = 0; k <= nw; ++k) {
        if (k > 0) cdf = cdf + sA[k - 1];
        const float zk = sB[k];
        if (cdf <= u) { cb = cdf; zb = zk; }
        else if (!found) { ca = cdf; za = zk; found = true; }
    }
    if (!found) { ca = cb; za = zb; }                      // above clamps to the last bin
    float denom = ca - cb;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - cb) / denom;
    const float z = zb + t * (za - zb);
    return (lane < Sf) ? z : INFINITY;
}

__device__ __forceinline__ float bitonic_sort64(float v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(v, j, 64);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & j) == 0);
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return v;
}

// ---- the fused kernel -------------------------------------------------------------------------------
template <int NNETS>
__global__ void __launch_bounds__(kWavesPerBlock * 64, 2)
render_forward_kernel(RenderArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    {   // decoder stream -> LDS (straight copy, 16 B per lane)
        const f32x4* src = (const f32x4*)a.decoder;
        f32x4* dst = (f32x4*)lds;
        for (int i = tid; i < kDecoderFloats / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    float* tile = lds + kDecoderFloats + wave * kWaveTile;      // [sample][kPitch] coarse weights, then fine depths
    float* sA = tile + kMaxS * kPitch;
    float* sB = sA + 64;
    const int SN = NNETS - 1;                                    // density comes from the last net (triplane_cond.py:958)
    const int Sc = a.Sc, Sf = a.Sf;

    // ---- ray -> (workgroup, wave, lane) assignment -------------------------------------------------------------
    // Rays of one pixel COLUMN project onto the same texels of the (x,z) plane, rays of one pixel ROW onto the same
    // texels of the (z,y) plane.  Workgroup b runs on XCD b % 8 (private L2), so when the image is R x R with
    // R = 16 * ns and ns | 8, XCD k is given the 16-pixel-wide column strip k % ns of every image: the (x,z) texels a
    // strip needs (~1 MB per image) stay in that XCD's L2 for all its rows instead of being re-fetched by 8 L2s.
    // A workgroup is a 16 x 16 pixel block, a wave two 16-pixel rows of it (same row => same (z,y) texels: one fetch
    // serves 16 lanes).  Any other shape falls back to consecutive rays.
    int ray0 = (blockIdx.x * kWavesPerBlock + wave) * 32;        // linear assignment (and the bound for 'live')
    int g_lane = ray0 + j;
    {
        const int R = a.res;
        const int ns = R >> 4;
        if (R > 0 && (R & 15) == 0 && ns <= 8 && (8 % ns) == 0 && a.rays_per_img == R * R && ((a.total_rays / 256) & 7) == 0) {
            const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
            const int strip = xcd % ns, sub = xcd / ns, per = 8 / ns;         // `per` XCDs share a strip
            const int blk = slot * per + sub;                                 // (image, row block) index within the strip
            const int n_i = blk / ns, rb = blk - n_i * ns;                    // ns row blocks of 16 rows per image
            const int row = rb * 16 + wave * 2 + (j >> 4), col = strip * 16 + (j & 15);
            g_lane = n_i * a.rays_per_img + row * R + col;
            ray0 = 0;                                                         // every lane is a real ray in this mode
        }
    }
    if (ray0 >= a.total_rays) return;
    const int g = min(g_lane, a.total_rays - 1);                 // tail lanes shadow the last ray, never store
    const bool live = g_lane < a.total_rays;
    const int n_img = g / a.rays_per_img;
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, 0, a.planes_total_bytes, 0x00020000);
    const unsigned img = (unsigned)n_img * a.img_bytes;
    const float ox = a.ray_o[g * 3 + 0], oy = a.ray_o[g * 3 + 1], oz = a.ray_o[g * 3 + 2];
    const float dx = a.ray_d[g * 3 + 0], dy = a.ray_d[g * 3 + 1], dz = a.ray_d[g * 3 + 2];
    const float cs = a.coord_scale;
    const float* uc = a.u_coarse + (size_t)g * Sc;

    // ------------------------------ phase A: coarse densities -> weights ------------------------------
    {
        float T = 1.f, z_prev = 0.f, s_prev = 0.f;
        for (int i = 0; i < Sc; ++i) {
            const float z = coarse_depth(a, g, i, uc[i]);
            float feat[16];
            gather_features(a, rsrc, img, h, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), feat);
            f32x16 h0, h1;
            mlp_layer1(lds, SN, lane, h, feat, h0, h1);
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

    // ------------------------------ phase B: importance depths, sorted --------------------------------
    for (int r = 0; r < 32; ++r) {
        const int gr = __shfl(g, r, 64);                                         // global index of the wave's r-th ray (wave-uniform)
        const bool r_live = __shfl((int)live, r, 64) != 0;
        const float w_i = (lane < Sc - 1) ? tile[lane * kPitch + r] : 0.f;
        const float z_i = (lane < Sc) ? coarse_depth(a, gr, lane, a.u_coarse[(size_t)gr * Sc + lane]) : 0.f;
        const float u   = (lane < Sf) ? a.u_fine[(size_t)gr * Sf + lane] : 2.f;
        if (a.dbg_wcoarse && lane < Sc - 1 && r_live) a.dbg_wcoarse[(size_t)gr * (Sc - 1) + lane] = w_i;
        float zf = importance_depth(Sc, Sf, lane, w_i, z_i, u, sA, sB);
        zf = bitonic_sort64(zf, lane);
        wave_sync();
        if (lane < Sf) tile[lane * kPitch + r] = zf;
        if (a.dbg_fine && lane < Sf && r_live) a.dbg_fine[(size_t)gr * Sf + lane] = zf;
    }
    wave_sync();

    // ------------------------------ phase C: merged decode + composite --------------------------------
    float acc[NNETS][16], prev[NNETS][16];
#pragma unroll
    for (int n = 0; n < NNETS; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[n][r] = 0.f; prev[n][r] = 0.f; }
    float T = 1.f, z_prev = 0.f, s_prev = 0.f, w_sum = 0.f, wz_sum = 0.f, z_first = 0.f;
    int ic = 0, jf = 0;
    float zc = coarse_depth(a, g, 0, uc[0]);
    float zf = (Sf > 0) ? tile[j] : INFINITY;
    const int S = Sc + Sf;
    for (int k = 0; k < S; ++k) {
        const bool take_c = (zc <= zf);
        const float z = take_c ? zc : zf;
        if (take_c) { ++ic; zc = (ic < Sc) ? coarse_depth(a, g, ic, uc[ic]) : INFINITY; }
        else        { ++jf; zf = (jf < Sf) ? tile[jf * kPitch + j] : INFINITY; }

        float feat[16];
        gather_features(a, rsrc, img, h, cs * fmaf(z, dx, ox), cs * fmaf(z, dy, oy), cs * fmaf(z, dz, oz), feat);
        // The density net goes first: its sigma closes interval k-1 (weight w), after which every net's
        // colours are folded into the accumulators as soon as its layer 2 retires — only `prev` (the
        // other end of the midpoint rule) stays live across samples.
        float sigma = 0.f, hw = 0.f;
#pragma unroll
        for (int idx = 0; idx < NNETS; ++idx) {
            const int n = (idx == 0) ? SN : idx - 1;
            f32x16 h0, h1, o;
            mlp_layer1(lds, n, lane, h, feat, h0, h1);
            if (idx == 0) {
                sigma = mlp_sigma(lds, h, h0, h1);
                if (k == 0) z_first = z;
                else {
                    const float dens = softplus20(0.5f * (s_prev + sigma) - 1.f);
                    const float alpha = 1.f - fast_exp(-dens * (z - z_prev));
                    const float w = alpha * T;
                    T *= (1.f - alpha + 1e-10f);
                    hw = 0.5f * w;
                    w_sum += w;
                    wz_sum = fmaf(w, 0.5f * (z_prev + z), wz_sum);
                }
            }
            mlp_layer2(lds, n, lane, h, h0, h1, o);
            const bool squash = (n == 0) || (NNETS == 1) || a.sem_sigmoid;     // raw logits for the label net (triplane_cond.py:960-964)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float c = squash ? sigmoid_clamped(o[r]) : o[r];
                acc[n][r] = fmaf(hw, prev[n][r] + c, acc[n][r]);               // hw == 0 for the first sample
                prev[n][r] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        z_prev = z; s_prev = sigma;
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

__global__ void render_init_minmax_kernel(unsigned* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

__global__ void render_clamp_depth_kernel(float* depth, int n, const unsigned* mm)
{
    const float lo = order_unkey(mm[0]), hi = order_unkey(mm[1]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float d = depth[i];
        if (d != d) d = INFINITY;                         // nan_to_num(nan=inf)
        depth[i] = fminf(fmaxf(d, lo), hi);
    }
}

// ---- point queries: sample -> decode, no compositing (renderer.py:142-148 run_model) ---------------
template <int NNETS>
__global__ void __launch_bounds__(kWavesPerBlock * 64, 2)
sample_points_kernel(RenderArgs a, const float* coords, int pts_per_img, int total_pts, float* rgb, float* sigma_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    {
        const f32x4* src = (const f32x4*)a.decoder;
        f32x4* dst = (f32x4*)lds;
        for (int i = tid; i < kDecoderFloats / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int SN = NNETS - 1;
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, 0, a.planes_total_bytes, 0x00020000);
    const int tiles = (total_pts + 31) / 32;
    for (int t = blockIdx.x * kWavesPerBlock + wave; t < tiles; t += gridDim.x * kWavesPerBlock) {
        const int p = min(t * 32 + j, total_pts - 1);
        const bool live = (t * 32 + j) < total_pts;
        const unsigned img = (unsigned)(p / pts_per_img) * a.img_bytes;
        const float cs = a.coord_scale;
        float feat[16];
        gather_features(a, rsrc, img, h, cs * coords[(size_t)p * 3], cs * coords[(size_t)p * 3 + 1], cs * coords[(size_t)p * 3 + 2], feat);
#pragma unroll
        for (int n = 0; n < NNETS; ++n) {
            f32x16 h0, h1, o;
            mlp_layer1(lds, n, lane, h, feat, h0, h1);
            if (n == SN) { const float s = mlp_sigma(lds, h, h0, h1); if (live && h == 0) sigma_out[p] = s; }
            mlp_layer2(lds, n, lane, h, h0, h1, o);
            const bool squash = (n == 0) || (NNETS == 1) || a.sem_sigmoid;
            if (live) {
                float* dst = rgb + (size_t)p * (NNETS * 32) + n * 32 + h * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = squash ? sigmoid_clamped(o[q * 4 + e]) : o[q * 4 + e];
                    *(f32x4*)(dst + q * 8) = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- importance sampling alone (unit-testable index work) ---------------------------------------------
__global__ void __launch_bounds__(64) importance_kernel(const float* z_coarse, const float* w_coarse, const float* u_fine,
                                                        float* z_fine, int rays, int Sc, int Sf, int sort)
{
    __shared__ float sA[64], sB[64];
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < rays; r += gridDim.x) {
        const float w_i = (lane < Sc - 1) ? w_coarse[(size_t)r * (Sc - 1) + lane] : 0.f;
        const float z_i = (lane < Sc) ? z_coarse[(size_t)r * Sc + lane] : 0.f;
        const float u = (lane < Sf) ? u_fine[(size_t)r * Sf + lane] : 2.f;
        float zf = importance_depth(Sc, Sf, lane, w_i, z_i, u, sA, sB);
        if (sort) zf = bitonic_sort64(zf, lane);
        if (lane < Sf) z_fine[(size_t)r * Sf + lane] = zf;
        wave_sync();
    }
}

// ---- layout / packing helpers ---------------------------------------------------------------------------
// planes NCHW [N][3*32][H][W] -> channels-last per plane [N][3][H][W][32], through a padded LDS tile so
// both the read (along W) and the write (along C) are coalesced.
__global__ void __launch_bounds__(256) planes_to_cl_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW)
{
    __shared__ float t[32][65];
    const int np = blockIdx.y;                       // n*3 + plane
    const int p0 = blockIdx.x * 64;                  // pixel tile
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 32; c += 4) {
        const int p = p0 + tx;
        t[c][tx] = (p < HW) ? src[((size_t)np * 32 + c) * HW + p] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, pr = threadIdx.x >> 5;
    for (int pp = pr; pp < 64; pp += 8) {
        const int p = p0 + pp;
        if (p < HW) dst[((size_t)np * HW + p) * 32 + c] = t[c][pp];
    }
}

struct PackArgs {
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
    int n_nets; float wg1, wg2, bg;                  // FullyConnectedLayer gains (networks_stylegan2.py:111-120)
};

__global__ void __launch_bounds__(256) pack_decoder_kernel(PackArgs p, float* out)
{
    const int SN = p.n_nets - 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kDecoderFloats; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < 2 * kNetStride) {
            const int n = i / kNetStride, e = i % kNetStride;
            const int q4 = e / 256, lane = (e % 256) / 4, q = q4 * 4 + (e & 3);
            const int row = lane & 31, h = lane >> 5;
            if (n < p.n_nets) {
                if (q < 32) {                                   // layer 1: tile t, k-step kk
                    const int t = q >> 4, kk = q & 15;
                    v = p.w1[n][(32 * t + row) * 32 + 16 * h + kk] * p.wg1;
                } else {                                        // layer 2: colour row `row`, hidden pi(s, h)
                    const int s = q - 32, t = s >> 4, r = s & 15;
                    const int hid = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    v = p.w2[n][(1 + row) * 64 + hid] * p.wg2;
                }
            }
        } else if (i < OFF_B2) {
            const int e = i - OFF_B1, n = e / 64, h = (e / 32) & 1, s = e & 31, t = s >> 4, r = s & 15;
            if (n < p.n_nets) v = p.b1[n][32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * p.bg;
        } else if (i < OFF_W2S) {
            const int e = i - OFF_B2, n = e / 32, h = (e / 16) & 1, r = e & 15;
            if (n < p.n_nets) v = p.b2[n][1 + (r & 3) + 8 * (r >> 2) + 4 * h] * p.bg;
        } else if (i < OFF_B2S) {
            const int e = i - OFF_W2S, h = e / 32, s = e & 31, t = s >> 4, r = s & 15;
            v = p.w2[SN][32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * p.wg2;
        } else if (i == OFF_B2S) {
            v = p.b2[SN][0] * p.bg;
        }
        out[i] = v;
    }
}

static int check_render_common(const p3d_render_desc* d)
{
    P3D_REQUIRE(d, "render: null descriptor");
    P3D_REQUIRE(d->n_nets == 1 || d->n_nets == 2, "render: n_nets must be 1 or 2 (got %d)", d->n_nets);
    P3D_REQUIRE(d->plane_h >= 1 && d->plane_w >= 1, "render: bad plane size");
    P3D_REQUIRE(d->box_warp != 0.f, "render: box_warp must be non-zero");
    {   // 32-bit buffer addressing: the whole plane tensor must span < 2 GiB, a pixel stride < 64 KiB (24-bit multiplies)
        const int64_t istr = d->pixel_stride > 0 ? d->image_stride : (int64_t)3 * d->plane_h * d->plane_w * 32;
        if ((int64_t)d->n_img * istr * 4 >= ((int64_t)1 << 31) || (int64_t)d->plane_h * d->plane_w >= (1 << 24) || d->pixel_stride * 4 >= (1 << 16))
            return fail(P3D_ERR_UNSUPPORTED, "render: plane tensor too large for 32-bit buffer addressing (%lld images)", (long long)d->n_img);
    }
    P3D_REQUIRE(d->pixel_stride == 0 || (d->pixel_stride % 4 == 0 && d->plane_stride % 4 == 0 && d->image_stride % 4 == 0), "render: plane strides must keep texels 16-byte aligned");
    return P3D_OK;
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_render_decoder_floats(void) { return kDecoderFloats; }

extern "C" int p3d_planes_to_channels_last(const float* planes_nchw, float* planes_cl, int32_t n_img, int32_t h, int32_t w, p3d_stream_t stream)
{
    P3D_REQUIRE(planes_nchw && planes_cl && n_img >= 0 && h > 0 && w > 0, "planes_to_channels_last: bad arguments");
    if (n_img == 0) return P3D_OK;
    const int HW = h * w;
    hipLaunchKernelGGL(planes_to_cl_kernel, dim3((HW + 63) / 64, n_img * 3), dim3(256), 0, (hipStream_t)stream, planes_nchw, planes_cl, HW);
    count_launch(FAM_AUX);
    return check_launch("planes_to_channels_last");
}

extern "C" int p3d_pack_decoder(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                                const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                                int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream)
{
    P3D_REQUIRE(n_nets == 1 || n_nets == 2, "pack_decoder: n_nets must be 1 or 2");
    P3D_REQUIRE(w1_a && b1_a && w2_a && b2_a && packed, "pack_decoder: null weights");
    P3D_REQUIRE(n_nets == 1 || (w1_b && b1_b && w2_b && b2_b), "pack_decoder: second net missing");
    PackArgs p;
    p.w1[0] = w1_a; p.b1[0] = b1_a; p.w2[0] = w2_a; p.b2[0] = b2_a;
    p.w1[1] = w1_b; p.b1[1] = b1_b; p.w2[1] = w2_b; p.b2[1] = b2_b;
    p.n_nets = n_nets;
    p.wg1 = lr_mul / sqrtf(32.f); p.wg2 = lr_mul / sqrtf(64.f); p.bg = lr_mul;
    hipLaunchKernelGGL(pack_decoder_kernel, dim3((kDecoderFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, packed);
    count_launch(FAM_AUX);
    return check_launch("pack_decoder");
}

static void fill_args(RenderArgs& a, const p3d_render_desc* d)
{
    a.H = d->plane_h; a.W = d->plane_w; a.Sc = d->depth_resolution; a.Sf = d->depth_resolution_importance;
    a.ray_start = d->ray_start; a.ray_end = d->ray_end; a.coord_scale = 2.f / d->box_warp;
    a.lin_step = a.Sc > 1 ? (d->ray_end - d->ray_start) / (float)(a.Sc - 1) : 0.f;
    a.disparity = d->disparity_space_sampling; a.white_back = d->white_back; a.sem_sigmoid = d->semantic_sigmoid;
    if (d->pixel_stride > 0) { a.plane_stride = d->plane_stride; a.pix_stride = d->pixel_stride; a.img_stride = d->image_stride; }
    else { a.plane_stride = (int64_t)a.H * a.W * 32; a.pix_stride = 32; a.img_stride = 3 * a.plane_stride; }
    a.plane_bytes = (unsigned)(a.plane_stride * 4); a.pix_bytes = (unsigned)(a.pix_stride * 4); a.img_bytes = (unsigned)(a.img_stride * 4);
    a.planes_total_bytes = (unsigned)((int64_t)d->n_img * a.img_stride * 4);
}

extern "C" int p3d_render_forward(const float* planes_cl, const float* decoder, const float* ray_o, const float* ray_d,
                                  const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                                  const p3d_render_desc* d, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                                  float* dbg_fine, float* dbg_wcoarse, p3d_stream_t stream)
{
    int rc = check_render_common(d);
    if (rc != P3D_OK) return rc;
    P3D_REQUIRE(planes_cl && decoder && ray_o && ray_d && u_coarse && feat && depth && wsum && minmax_ws, "render_forward: null pointer");
    P3D_REQUIRE(d->n_img >= 0 && d->rays_per_img >= 1, "render_forward: bad ray counts");
    P3D_REQUIRE((t_start == nullptr) == (t_end == nullptr), "render_forward: t_start/t_end must be given together");
    if (d->depth_resolution < 4 || d->depth_resolution > kMaxS || d->depth_resolution_importance < 1 ||
        d->depth_resolution_importance > kMaxS || !u_fine)
        return fail(P3D_ERR_UNSUPPORTED, "render_forward: fused kernel needs 4 <= depth_resolution <= %d and 1 <= depth_resolution_importance <= %d (got %d, %d)",
                    kMaxS, kMaxS, d->depth_resolution, d->depth_resolution_importance);
    const int64_t total = (int64_t)d->n_img * d->rays_per_img;
    P3D_REQUIRE(total <= INT32_MAX / 64, "render_forward: too many rays");
    if (total == 0) return P3D_OK;
    RenderArgs a{};
    fill_args(a, d);
    a.planes = planes_cl; a.decoder = decoder; a.ray_o = ray_o; a.ray_d = ray_d; a.u_coarse = u_coarse; a.u_fine = u_fine;
    a.t_start = t_start; a.t_end = t_end; a.feat = feat; a.depth = depth; a.wsum = wsum;
    a.dbg_fine = dbg_fine; a.dbg_wcoarse = dbg_wcoarse; a.minmax = minmax_ws;
    a.total_rays = (int)total; a.rays_per_img = d->rays_per_img;
    { int r = 1; while (r * r < d->rays_per_img) ++r; a.res = (r * r == d->rays_per_img && d->raster_order) ? r : 0; }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)(kDecoderFloats + kWavesPerBlock * kWaveTile) * sizeof(float);
    const int blocks = (int)((total + kWavesPerBlock * 32 - 1) / (kWavesPerBlock * 32));
    hipLaunchKernelGGL(render_init_minmax_kernel, dim3(1), dim3(1), 0, s, minmax_ws);
    if (d->n_nets == 1) {
        static hipError_t once1 = hipFuncSetAttribute((const void*)render_forward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (once1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once1));
        hipLaunchKernelGGL(render_forward_kernel<1>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a);
    } else {
        static hipError_t once2 = hipFuncSetAttribute((const void*)render_forward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (once2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once2));
        hipLaunchKernelGGL(render_forward_kernel<2>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a);
    }
    rc = check_launch("render_forward");
    if (rc != P3D_OK) return rc;
    const int cb = (int)((total + 255) / 256);
    hipLaunchKernelGGL(render_clamp_depth_kernel, dim3(cb > 1024 ? 1024 : cb), dim3(256), 0, s, depth, (int)total, minmax_ws);
    count_launch(FAM_RENDER);
    return check_launch("render_clamp_depth");
}

extern "C" int p3d_sample_points(const float* planes_cl, const float* decoder, const float* coords, const p3d_render_desc* d,
                                 int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream)
{
    int rc = check_render_common(d);
    if (rc != P3D_OK) return rc;
    P3D_REQUIRE(planes_cl && decoder && coords && rgb && sigma, "sample_points: null pointer");
    P3D_REQUIRE(d->n_img >= 0 && pts_per_img >= 1, "sample_points: bad point counts");
    const int64_t total = (int64_t)d->n_img * pts_per_img;
    P3D_REQUIRE(total <= INT32_MAX / 64, "sample_points: too many points");
    if (total == 0) return P3D_OK;
    RenderArgs a{};
    fill_args(a, d);
    a.planes = planes_cl; a.decoder = decoder;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)kDecoderFloats * sizeof(float);
    int64_t tiles = (total + 31) / 32;
    int blocks = (int)((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
    if (blocks > kNumCU * 2) blocks = kNumCU * 2;
    if (d->n_nets == 1) hipLaunchKernelGGL(sample_points_kernel<1>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a, coords, pts_per_img, (int)total, rgb, sigma);
    else                hipLaunchKernelGGL(sample_points_kernel<2>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a, coords, pts_per_img, (int)total, rgb, sigma);
    count_launch(FAM_RENDER);
    return check_launch("sample_points");
}

extern "C" int p3d_importance_sample(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine,
                                     int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted, p3d_stream_t stream)
{
    P3D_REQUIRE(z_coarse && w_coarse && u_fine && z_fine, "importance_sample: null pointer");
    if (depth_resolution < 4 || depth_resolution > kMaxS || n_importance < 1 || n_importance > kMaxS)
        return fail(P3D_ERR_UNSUPPORTED, "importance_sample: needs 4 <= depth_resolution <= %d, 1 <= n_importance <= %d", kMaxS, kMaxS);
    if (n_rays <= 0) return P3D_OK;
    const int blocks = n_rays < kNumCU * 16 ? n_rays : kNumCU * 16;
    hipLaunchKernelGGL(importance_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, z_coarse, w_coarse, u_fine, z_fine, n_rays, depth_resolution, n_importance, sorted);
    count_launch(FAM_RENDER);
    return check_launch("importance_sample");
}
