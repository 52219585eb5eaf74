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
#include "render_device.h"

namespace p3d {

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
template <int NNETS, bool DUAL = false>
__global__ void __launch_bounds__(kWavesPerBlock * 64, 2)
sample_points_kernel(RenderArgs a, const float* coords, int pts_per_img, int total_pts, float* rgb, float* sigma_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    {
        const f32x4* src = (const f32x4*)a.decoder;
        f32x4* dst = (f32x4*)lds;
        for (int i = tid; i < (DUAL ? kDecoderFloatsDual : kDecoderFloats) / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int SN = NNETS - 1;
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, 0, a.planes_total_bytes, 0x00020000);
    const rsrc_t rsrc_sem = DUAL ? __builtin_amdgcn_make_buffer_rsrc((void*)a.planes2, 0, a.planes_total_bytes, 0x00020000) : rsrc;
    const int tiles = (total_pts + 31) / 32;
    for (int t = blockIdx.x * kWavesPerBlock + wave; t < tiles; t += gridDim.x * kWavesPerBlock) {
        const int p = min(t * 32 + j, total_pts - 1);
        const bool live = (t * 32 + j) < total_pts;
        const unsigned img = (unsigned)(p / pts_per_img) * a.img_bytes;
        const float cs = a.coord_scale;
        float feat[16], feat_tex[16];
        gather_features<!DUAL>(a, rsrc_sem, img, h, cs * coords[(size_t)p * 3], cs * coords[(size_t)p * 3 + 1], cs * coords[(size_t)p * 3 + 2], feat);
        if (DUAL) gather_features<false>(a, rsrc, img, h, cs * coords[(size_t)p * 3], cs * coords[(size_t)p * 3 + 1], cs * coords[(size_t)p * 3 + 2], feat_tex);
#pragma unroll
        for (int n = 0; n < NNETS; ++n) {
            f32x16 h0, h1, o;
            if (DUAL && n == 0) mlp_layer1<false, true>(lds, n, lane, h, feat_tex, h0, h1, feat);
            else                mlp_layer1(lds, n, lane, h, feat, h0, h1);
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
                                                        float* z_fine, int32_t* bins, uint32_t* merge, int rays, int Sc, int Sf, int sort)
{
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < rays; r += gridDim.x) {
        const float w_i = (lane < Sc - 1) ? w_coarse[(size_t)r * (Sc - 1) + lane] : 0.f;
        const float z_i = (lane < Sc) ? z_coarse[(size_t)r * Sc + lane] : 0.f;
        const float u = (lane < Sf) ? u_fine[(size_t)r * Sf + lane] : 2.f;
        const float w1[1] = {w_i}, z1[1] = {z_i}, u1[1] = {u};
        float zf[1];
        int ind[1];
        importance_depth<1>(Sc, Sf, lane, w1, z1, u1, zf, &ind);
        if (bins && lane < Sf) bins[(size_t)r * Sf + lane] = ind[0];
        if (sort) bitonic_sort64<1>(zf, lane);
        if (lane < Sf) z_fine[(size_t)r * Sf + lane] = zf[0];
        if (merge) {                                             // the fused kernel's phase-C merge over the same heads (every lane walks it; lane 0 writes)
            uint32_t word[4] = {0u, 0u, 0u, 0u};
            int ic = 0, jf = 0;
            float zc = lane_bcast(z_i, 0), zfh = lane_bcast(zf[0], 0);
            for (int k = 0; k < Sc + Sf; ++k) {
                if (merge_takes_coarse(zc, zfh)) { ++ic; zc = (ic < Sc) ? lane_bcast(z_i, ic) : INFINITY; }
                else { word[k >> 5] |= 1u << (k & 31); ++jf; zfh = (jf < Sf) ? lane_bcast(zf[0], jf) : INFINITY; }
            }
            if (lane == 0) { for (int i = 0; i < 4; ++i) merge[(size_t)r * 4 + i] = word[i]; }
        }
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
    int n_nets; float wg1[2], wg2, bg;               // FullyConnectedLayer gains (networks_stylegan2.py:111-120); wg1 per net (fan-in 32 or 64)
    int w1_in[2];                                    // inputs of each net's first layer (row length of w1): 32, or 64 for the dual colour net
    int total;                                       // kDecoderFloats or kDecoderFloatsDual
    int bf16_order;                                  // weights in the [block][lane][8] order of the bf16x3 decoder (render_device.h) instead of the f32-MFMA order
};

__global__ void __launch_bounds__(256) pack_decoder_kernel(PackArgs p, float* out)
{
    const int SN = p.n_nets - 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.total; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i >= OFF_W1X) {                                     // dual colour net, first layer, inputs 32..63: same per-lane order as block q < 32
            const int e = i - OFF_W1X;
            const int q4 = e / 256, lane = (e % 256) / 4, q = q4 * 4 + (e & 3);
            const int row = lane & 31, h = lane >> 5, t = q >> 4, kk = q & 15;
            v = p.w1[0][(32 * t + row) * p.w1_in[0] + 32 + 16 * h + kk] * p.wg1[0];
        } else if (i < 2 * kNetStride && (p.bf16_order == 1 || (p.bf16_order == 2 && (i % kNetStride) < kNetStride / 2))) {   // (2: layer 1 only — p3d_pack_decoder_l1x6)
            const int n = i / kNetStride, f = i % kNetStride;
            const int block = f / 512, lane = (f % 512) / 8, e = f & 7;
            const int row = lane & 31, kb = lane >> 5;
            if (n < p.n_nets) {
                if (block < 4) {                                 // layer 1: tile t, k-step s; k (kb, e) <-> input channel 16 kb + 8 s + e
                    const int t = block >> 1, ks = block & 1;
                    v = p.w1[n][(32 * t + row) * p.w1_in[n] + 16 * kb + 8 * ks + e] * p.wg1[n];
                } else {                                         // layer 2: k-step s; k (kb, e) <-> the hidden unit accumulator register 8 (s & 1) + e of tile s >> 1 holds
                    const int ks = block - 4, r = 8 * (ks & 1) + e;
                    const int hid = 32 * (ks >> 1) + (r & 3) + 8 * (r >> 2) + 4 * kb;
                    v = p.w2[n][(1 + row) * 64 + hid] * p.wg2;
                }
            }
        } else if (i < 2 * kNetStride) {
            const int n = i / kNetStride, e = i % kNetStride;
            const int q4 = e / 256, lane = (e % 256) / 4, q = q4 * 4 + (e & 3);
            const int row = lane & 31, h = lane >> 5;
            if (n < p.n_nets) {
                if (q < 32) {                                   // layer 1: tile t, k-step kk
                    const int t = q >> 4, kk = q & 15;
                    v = p.w1[n][(32 * t + row) * p.w1_in[n] + 16 * h + kk] * p.wg1[n];
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

static int pack_decoder_impl(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                             const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                             int32_t n_nets, float lr_mul, float* packed, int bf16_order, p3d_stream_t stream)
{
    P3D_REQUIRE(n_nets == 1 || n_nets == 2, "pack_decoder: n_nets must be 1 or 2");
    P3D_REQUIRE(w1_a && b1_a && w2_a && b2_a && packed, "pack_decoder: null weights");
    P3D_REQUIRE(n_nets == 1 || (w1_b && b1_b && w2_b && b2_b), "pack_decoder: second net missing");
    PackArgs p;
    p.w1[0] = w1_a; p.b1[0] = b1_a; p.w2[0] = w2_a; p.b2[0] = b2_a;
    p.w1[1] = w1_b; p.b1[1] = b1_b; p.w2[1] = w2_b; p.b2[1] = b2_b;
    p.n_nets = n_nets;
    p.wg1[0] = p.wg1[1] = lr_mul / sqrtf(32.f); p.wg2 = lr_mul / sqrtf(64.f); p.bg = lr_mul;
    p.w1_in[0] = p.w1_in[1] = 32; p.total = kDecoderFloats; p.bf16_order = bf16_order;
    hipLaunchKernelGGL(pack_decoder_kernel, dim3((kDecoderFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, packed);
    count_launch(FAM_AUX);
    return check_launch("pack_decoder");
}

extern "C" int p3d_pack_decoder(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                                const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                                int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream)
{
    return pack_decoder_impl(w1_a, b1_a, w2_a, b2_a, w1_b, b1_b, w2_b, b2_b, n_nets, lr_mul, packed, 0, stream);
}

extern "C" int p3d_pack_decoder_l1x6(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                                     const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                                     int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream)
{
    return pack_decoder_impl(w1_a, b1_a, w2_a, b2_a, w1_b, b1_b, w2_b, b2_b, n_nets, lr_mul, packed, 2, stream);
}

extern "C" int p3d_pack_decoder_bf16x3(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                                       const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                                       int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream)
{
    return pack_decoder_impl(w1_a, b1_a, w2_a, b2_a, w1_b, b1_b, w2_b, b2_b, n_nets, lr_mul, packed, 1, stream);
}

extern "C" int p3d_render_decoder_floats_dual(void) { return kDecoderFloatsDual; }

extern "C" int p3d_pack_decoder_dual(const float* w1_tex, const float* b1_tex, const float* w2_tex, const float* b2_tex,
                                     const float* w1_sem, const float* b1_sem, const float* w2_sem, const float* b2_sem,
                                     float lr_mul, float* packed, p3d_stream_t stream)
{
    P3D_REQUIRE(w1_tex && b1_tex && w2_tex && b2_tex && w1_sem && b1_sem && w2_sem && b2_sem && packed, "pack_decoder_dual: null weights");
    PackArgs p;
    p.w1[0] = w1_tex; p.b1[0] = b1_tex; p.w2[0] = w2_tex; p.b2[0] = b2_tex;     // net 0: colours from cat(texture, semantic) features, [64, 64] first layer
    p.w1[1] = w1_sem; p.b1[1] = b1_sem; p.w2[1] = w2_sem; p.b2[1] = b2_sem;     // net 1: density + labels from the semantic features
    p.n_nets = 2;
    p.wg1[0] = lr_mul / sqrtf(64.f); p.wg1[1] = lr_mul / sqrtf(32.f); p.wg2 = lr_mul / sqrtf(64.f); p.bg = lr_mul;
    p.w1_in[0] = 64; p.w1_in[1] = 32; p.total = kDecoderFloatsDual; p.bf16_order = 0;
    hipLaunchKernelGGL(pack_decoder_kernel, dim3((kDecoderFloatsDual + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, packed);
    count_launch(FAM_AUX);
    return check_launch("pack_decoder_dual");
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

static int render_forward_impl(const float* planes_cl, const float* planes_sem_cl, const float* decoder, const float* ray_o, const float* ray_d,
                               const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                               const p3d_render_desc* d, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                               float* dbg_fine, float* dbg_wcoarse, p3d_stream_t stream, int32_t* dbg_bins = nullptr)
{
    const bool dual = planes_sem_cl != nullptr;
    int rc = check_render_common(d);
    if (rc != P3D_OK) return rc;
    P3D_REQUIRE(planes_cl && decoder && ray_o && ray_d && u_coarse && feat && depth && wsum && minmax_ws, "render_forward: null pointer");
    P3D_REQUIRE(!dual || d->n_nets == 2, "render_forward_dual: the two-plane-set renderer has two decoders (n_nets = 2)");
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
    a.planes = planes_cl; a.planes2 = planes_sem_cl; a.decoder = decoder; a.ray_o = ray_o; a.ray_d = ray_d; a.u_coarse = u_coarse; a.u_fine = u_fine;
    a.t_start = t_start; a.t_end = t_end; a.feat = feat; a.depth = depth; a.wsum = wsum;
    a.dbg_fine = dbg_fine; a.dbg_wcoarse = dbg_wcoarse; a.dbg_bins = dbg_bins; a.minmax = minmax_ws;
    a.total_rays = (int)total; a.rays_per_img = d->rays_per_img;
    { int r = 1; while (r * r < d->rays_per_img) ++r; a.res = (r * r == d->rays_per_img && d->raster_order) ? r : 0; }
    hipStream_t s = (hipStream_t)stream;
    // a block is one-per-CU (LDS): small launches take fewer waves per block so that every CU still gets one
    int wpb = dual ? kWavesPerBlockDual : kWavesPerBlock;
    while (wpb > 2 && (total + wpb * 32 - 1) / (wpb * 32) < kNumCU) wpb >>= 1;
    const size_t lds_bytes = (size_t)(dual ? kDecoderFloatsDual + kWavesPerBlockDual * kWaveTile
                                           : kDecoderFloats + kWavesPerBlock * (kWaveTile + kFeatTile + kTapTile)) * sizeof(float);      // + the cooperative gather's tiles
    const int blocks = (int)((total + wpb * 32 - 1) / (wpb * 32));
    hipLaunchKernelGGL(render_init_minmax_kernel, dim3(1), dim3(1), 0, s, minmax_ws);
    if (dual) {
        static std::atomic<uint64_t> onced_devs{0}; const hipError_t onced = reserve_lds_once((const void*)render_forward_kernel<2, false, true>, (int)lds_bytes, onced_devs);
        if (onced != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward_dual: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(onced));
        hipLaunchKernelGGL((render_forward_kernel<2, false, true>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
    } else if (d->mlp_bf16x3 == 2) {                        // layer 1 as bf16x6 (stream: p3d_pack_decoder_l1x6)
        const size_t lds6 = lds_bytes + (size_t)(kDecoderFloatsL1X6 - kDecoderFloats) * sizeof(float);
        if (d->n_nets == 1) {
            static std::atomic<uint64_t> oncex1_devs{0}; const hipError_t e1 = reserve_lds_once((const void*)render_forward_kernel<1, false, false, false, true>, (int)lds6, oncex1_devs);
            if (e1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds6, hipGetErrorString(e1));
            hipLaunchKernelGGL((render_forward_kernel<1, false, false, false, true>), dim3(blocks), dim3(wpb * 64), lds6, s, a);
        } else {
            static std::atomic<uint64_t> oncex2_devs{0}; const hipError_t e2 = reserve_lds_once((const void*)render_forward_kernel<2, false, false, false, true>, (int)lds6, oncex2_devs);
            if (e2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds6, hipGetErrorString(e2));
            hipLaunchKernelGGL((render_forward_kernel<2, false, false, false, true>), dim3(blocks), dim3(wpb * 64), lds6, s, a);
        }
    } else if (d->mlp_bf16x3) {
        if (d->n_nets == 1) {
            static std::atomic<uint64_t> onceb1_devs{0}; const hipError_t e1 = reserve_lds_once((const void*)render_forward_kernel<1, false, false, true>, (int)lds_bytes, onceb1_devs);
            if (e1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e1));
            hipLaunchKernelGGL((render_forward_kernel<1, false, false, true>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
        } else {
            static std::atomic<uint64_t> onceb2_devs{0}; const hipError_t e2 = reserve_lds_once((const void*)render_forward_kernel<2, false, false, true>, (int)lds_bytes, onceb2_devs);
            if (e2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e2));
            hipLaunchKernelGGL((render_forward_kernel<2, false, false, true>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
        }
    } else if (d->n_nets == 1) {
        static std::atomic<uint64_t> once1_devs{0}; const hipError_t once1 = reserve_lds_once((const void*)render_forward_kernel<1, false>, (int)lds_bytes, once1_devs);
        if (once1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once1));
        hipLaunchKernelGGL((render_forward_kernel<1, false>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
    } else {
        static std::atomic<uint64_t> once2_devs{0}; const hipError_t once2 = reserve_lds_once((const void*)render_forward_kernel<2, false>, (int)lds_bytes, once2_devs);
        if (once2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_forward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once2));
        hipLaunchKernelGGL((render_forward_kernel<2, false>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
    }
    rc = check_launch("render_forward");
    if (rc != P3D_OK) return rc;
    const int cb = (int)((total + 255) / 256);
    hipLaunchKernelGGL(render_clamp_depth_kernel, dim3(cb > 1024 ? 1024 : cb), dim3(256), 0, s, depth, (int)total, minmax_ws);
    count_launch(FAM_RENDER);
    return check_launch("render_clamp_depth");
}

extern "C" int p3d_render_forward(const float* planes_cl, const float* decoder, const float* ray_o, const float* ray_d,
                                  const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                                  const p3d_render_desc* d, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                                  float* dbg_fine, float* dbg_wcoarse, p3d_stream_t stream)
{
    return render_forward_impl(planes_cl, nullptr, decoder, ray_o, ray_d, u_coarse, u_fine, t_start, t_end, d, feat, depth, wsum, minmax_ws, dbg_fine, dbg_wcoarse, stream);
}

extern "C" int p3d_render_forward_debug(const float* planes_cl, const float* decoder, const float* ray_o, const float* ray_d,
                                        const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                                        const p3d_render_desc* d, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                                        float* dbg_fine, float* dbg_wcoarse, int32_t* dbg_bins, p3d_stream_t stream)
{
    return render_forward_impl(planes_cl, nullptr, decoder, ray_o, ray_d, u_coarse, u_fine, t_start, t_end, d, feat, depth, wsum, minmax_ws, dbg_fine, dbg_wcoarse, stream, dbg_bins);
}

extern "C" int p3d_render_forward_dual(const float* planes_tex_cl, const float* planes_sem_cl, const float* decoder_dual, const float* ray_o, const float* ray_d,
                                       const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                                       const p3d_render_desc* d, float* feat, float* depth, float* wsum, uint32_t* minmax_ws, p3d_stream_t stream)
{
    P3D_REQUIRE(planes_sem_cl, "render_forward_dual: null semantic planes");
    return render_forward_impl(planes_tex_cl, planes_sem_cl, decoder_dual, ray_o, ray_d, u_coarse, u_fine, t_start, t_end, d, feat, depth, wsum, minmax_ws, nullptr, nullptr, stream);
}

static int sample_points_impl(const float* planes_cl, const float* planes_sem_cl, const float* decoder, const float* coords, const p3d_render_desc* d,
                              int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream)
{
    const bool dual = planes_sem_cl != nullptr;
    int rc = check_render_common(d);
    if (rc != P3D_OK) return rc;
    P3D_REQUIRE(planes_cl && decoder && coords && rgb && sigma, "sample_points: null pointer");
    P3D_REQUIRE(!dual || d->n_nets == 2, "sample_points_dual: the two-plane-set renderer has two decoders (n_nets = 2)");
    P3D_REQUIRE(d->n_img >= 0 && pts_per_img >= 1, "sample_points: bad point counts");
    const int64_t total = (int64_t)d->n_img * pts_per_img;
    P3D_REQUIRE(total <= INT32_MAX / 64, "sample_points: too many points");
    if (total == 0) return P3D_OK;
    RenderArgs a{};
    fill_args(a, d);
    a.planes = planes_cl; a.planes2 = planes_sem_cl; a.decoder = decoder;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)(dual ? kDecoderFloatsDual : kDecoderFloats) * sizeof(float);
    int64_t tiles = (total + 31) / 32;
    int blocks = (int)((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
    if (blocks > kNumCU * 2) blocks = kNumCU * 2;
    if (dual)                hipLaunchKernelGGL((sample_points_kernel<2, true>), dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a, coords, pts_per_img, (int)total, rgb, sigma);
    else if (d->n_nets == 1) hipLaunchKernelGGL(sample_points_kernel<1>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a, coords, pts_per_img, (int)total, rgb, sigma);
    else                     hipLaunchKernelGGL(sample_points_kernel<2>, dim3(blocks), dim3(kWavesPerBlock * 64), lds_bytes, s, a, coords, pts_per_img, (int)total, rgb, sigma);
    count_launch(FAM_RENDER);
    return check_launch("sample_points");
}

extern "C" int p3d_sample_points(const float* planes_cl, const float* decoder, const float* coords, const p3d_render_desc* d,
                                 int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream)
{
    return sample_points_impl(planes_cl, nullptr, decoder, coords, d, pts_per_img, rgb, sigma, stream);
}

extern "C" int p3d_sample_points_dual(const float* planes_tex_cl, const float* planes_sem_cl, const float* decoder_dual, const float* coords,
                                      const p3d_render_desc* d, int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream)
{
    P3D_REQUIRE(planes_sem_cl, "sample_points_dual: null semantic planes");
    return sample_points_impl(planes_tex_cl, planes_sem_cl, decoder_dual, coords, d, pts_per_img, rgb, sigma, stream);
}

static int importance_sample_impl(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine, int32_t* bins, uint32_t* merge,
                                  int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted, p3d_stream_t stream)
{
    P3D_REQUIRE(z_coarse && w_coarse && u_fine && z_fine, "importance_sample: null pointer");
    P3D_REQUIRE(!merge || sorted, "importance_sample: the merge pattern is defined on the sorted importance depths");
    if (depth_resolution < 4 || depth_resolution > kMaxS || n_importance < 1 || n_importance > kMaxS)
        return fail(P3D_ERR_UNSUPPORTED, "importance_sample: needs 4 <= depth_resolution <= %d, 1 <= n_importance <= %d", kMaxS, kMaxS);
    if (n_rays <= 0) return P3D_OK;
    const int blocks = n_rays < kNumCU * 16 ? n_rays : kNumCU * 16;
    hipLaunchKernelGGL(importance_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, z_coarse, w_coarse, u_fine, z_fine, bins, merge, n_rays, depth_resolution, n_importance, sorted);
    count_launch(FAM_RENDER);
    return check_launch("importance_sample");
}

extern "C" int p3d_importance_sample(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine,
                                     int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted, p3d_stream_t stream)
{
    return importance_sample_impl(z_coarse, w_coarse, u_fine, z_fine, nullptr, nullptr, n_rays, depth_resolution, n_importance, sorted, stream);
}

extern "C" int p3d_importance_sample_index(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine, int32_t* bin_index, uint32_t* merge_words,
                                           int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted, p3d_stream_t stream)
{
    return importance_sample_impl(z_coarse, w_coarse, u_fine, z_fine, bin_index, merge_words, n_rays, depth_resolution, n_importance, sorted, stream);
}
