/*
 * p3d_hip.h — C ABI of libp3d_hip.so, the MI355X (gfx950) kernel library behind the
 * pix2pix3D generator/renderer hot path.
 *
 * Plain C: raw device pointers, explicit sizes/strides, an explicit hipStream_t (passed as
 * void*), no torch types.  Every entry point
 *   - borrows its inputs, writes into caller-allocated outputs (the caller keeps ownership,
 *     normally torch's caching allocator),
 *   - enqueues on the given stream and returns without host synchronisation,
 *   - returns P3D_OK (0) or a negative error code; never throws across the boundary;
 *     p3d_last_error() gives the message for the calling thread,
 *   - keeps no mutable global device state (filters / MLP weights are kernel arguments or LDS
 *     copies, never device globals), so concurrent streams are safe.
 *
 * Each declaration cites the reference interface it stands in for (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding used on the Python side.
 */
#ifndef P3D_HIP_H
#define P3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* p3d_stream_t;              /* hipStream_t */

enum p3d_status {
    P3D_OK              =  0,
    P3D_ERR_UNSUPPORTED = -1,            /* "no specialised kernel": same meaning as return_code -1 of
                                            filtered_lrelu_plugin (filtered_lrelu.cpp:56-60) */
    P3D_ERR_ARGUMENT    = -2,
    P3D_ERR_LAUNCH      = -3
};

enum p3d_dtype { P3D_F32 = 0, P3D_F16 = 1, P3D_F64 = 2,
                 P3D_F32_BF16X3 = 3     /* conv entry points only: fp32 tensors, fp32 accumulation, every product formed as three bf16
                                           products of (hi, lo) splits — ~2^-16 relative per product at up to 5x the fp32 matrix rate;
                                           weights come from p3d_modulate_weights with the same code ([32 x hi | 32 x lo] K rows)      */,
                 P3D_F32_BF16X6 = 4     /* conv entry points only (p3d_conv2d_nhwc*, p3d_conv2d_forward / _bwd_data / _bwd_weight*): fp32 tensors AND fp32 weights in the
                                           P3D_F32 layouts, fp32 accumulation; every operand is split IN REGISTERS into three bf16 pieces (hi + mid + lo = the
                                           fp32 value exactly: 3 x 8 significand bits) and every product formed as the six bf16 products of magnitude
                                           >= 2^-16 (hh, hm, mh, hl, lh, mm; the three dropped ones are <= 2^-23 relative together: the size of one fp32
                                           rounding) — fp32-accurate products at 6/16 of the f32-input MFMA's time, on the pipe that overlaps vector work */ };

/* ---- library services ------------------------------------------------------------------- */
const char* p3d_last_error(void);        /* message of the last failure on this host thread      */
int         p3d_abi_version(void);       /* bumped whenever a signature below changes            */
uint64_t    p3d_launch_count(void);      /* kernels enqueued by this library since load (tests use
                                            it to prove the HIP path ran, not a fallback)        */
/* per-kernel-family counters: which = 0 bias_act, 1 upfirdn2d, 2 filtered_lrelu, 3 render,
 * 4 conv/modconv, 5 layout/aux */
uint64_t    p3d_launch_count_of(int which);

/* ---- bias_act ---------------------------------------------------------------------------
 * Replaces bias_act_plugin.bias_act (torch_utils/ops/bias_act.cpp:36-94, kernel bias_act.cu:27-151).
 *   grad = 0: y = clamp(act(x + b[(i / step_b) % size_b]) * gain)
 *   grad = 1: x carries dy; xref/yref are the saved forward input/output; y = d/dx
 *   grad = 2: second-order term; dy carries the first-order upstream gradient
 * act: 1 linear, 2 relu, 3 lrelu, 4 tanh, 5 sigmoid, 6 elu, 7 selu, 8 softplus, 9 swish
 * (bias_act.py:23-33).  clamp < 0 disables clamping.  Null b/xref/yref/dy mean "absent"
 * (the plugin's empty tensor).  All tensors share x's dense layout; size_x <= INT32_MAX.   */
int p3d_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                 void* y, int dtype, int grad, int act, float alpha, float gain, float clamp,
                 int64_t size_x, int32_t size_b, int64_t step_b, p3d_stream_t stream);

/* ---- upfirdn2d --------------------------------------------------------------------------
 * Replaces upfirdn2d_plugin.upfirdn2d (torch_utils/ops/upfirdn2d.cpp:20-102, kernels
 * upfirdn2d.cu:33-204): zero-insert upsample, pad/crop, 2-D FIR, decimate, per (n, c) image.
 * Sizes are {W, H, C, N}; strides (in elements) use the same order so NCHW and channels_last
 * both work.  f is fp32 [fh, fw] with element strides f_stride = {x, y}.  The caller computes
 * out size = (in*up + pad0 + pad1 - f + down) / down (upfirdn2d.cpp:39-40); only pad0 is needed
 * here.  flip = 0 convolves (filter mirrored), flip = 1 correlates.                          */
int p3d_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                  const int32_t in_size[4], const int64_t in_stride[4],
                  const int32_t f_size[2], const int64_t f_stride[2],
                  const int32_t out_size[4], const int64_t out_stride[4],
                  int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                  int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, p3d_stream_t stream);
/* p3d_upfirdn2d that ADDS its result into y (y += upfirdn2d(x)): the skip-image sum of SynthesisBlock (training/networks_stylegan2.py:453-459:
 * img = upsample2d(img) + torgb) from the upsampling launch.  Channels-last tensors, 4-tap filters, C a multiple of 16 bytes; anything else
 * P3D_ERR_UNSUPPORTED.                                                                                                          */
int p3d_upfirdn2d_acc(const void* x, const float* f, void* y, int dtype,
                  const int32_t in_size[4], const int64_t in_stride[4],
                  const int32_t f_size[2], const int64_t f_stride[2],
                  const int32_t out_size[4], const int64_t out_stride[4],
                  int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                  int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, p3d_stream_t stream);

/* ---- filtered_lrelu ----------------------------------------------------------------------------
 * Replaces filtered_lrelu_plugin.filtered_lrelu (torch_utils/ops/filtered_lrelu.cpp:20-213; kernel parameters filtered_lrelu.h:18-72,
 * kernels filtered_lrelu.cu:143-1103):  y = fir_down( clamp( lrelu( fir_up(x + b) * up^2 * gain ) ) )  per (n, c) plane, in one pass.
 * Sizes are {W, H, C, N} and strides (in elements) use the same order.  fu / fd: dense fp32 tables [f_h][f_w] (a separable filter is
 * passed as its outer product, an absent one as the 1x1 table {1}).  The caller computes the sizes exactly as the plugin does
 * (filtered_lrelu.cpp:73-97): up-sampled extent c = x*up + pad0 + pad1 - (fu-1), y = (c - (fd-1) + down-1) / down, and for the sign
 * tensor s_h = y_h*down - (down-1) + (fd_h-1), active width likewise, rounded up to 16 elements, 4 elements per byte.
 * sign_mode 0: none.  1: WRITE the sign tensor s (uint8 [N][C][s_height][s_width_bytes], contiguous): element (x, y) of the up-sampled
 *   grid -> byte ((x + s_ofs_x) >> 2) of row (y + s_ofs_y), bits ((x + s_ofs_x) & 3) * 2: 0 passed, 1 negative (IEEE sign bit, as
 *   filtered_lrelu.cu:497-500), 2 clamped.  2: READ it instead of comparing (the backward configuration, filtered_lrelu.py:240-270):
 *   code & 1 -> times slope, code & 2 -> 0, elements outside the tensor pass unchanged; clamp is not applied.
 * sw_limit: valid bytes per sign row ((active width + 3) >> 2).  clamp = +inf disables clamping.
 * Returns P3D_ERR_UNSUPPORTED (-1, the plugin's "no specialised kernel" code, filtered_lrelu.cpp:56-60) when the tiles of this geometry
 * do not fit gfx950's 160 KB of LDS or s_ofs_x is not a multiple of 4 in write mode: the caller then takes the generic route
 * (upfirdn2d -> p3d_filtered_lrelu_act -> upfirdn2d), as the reference does.                                                        */
int p3d_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, uint8_t* s, void* y, int dtype,
                       const int32_t x_size[4], const int64_t x_stride[4], const int32_t y_size[4], const int64_t y_stride[4], int64_t b_stride,
                       int32_t fu_w, int32_t fu_h, int32_t fd_w, int32_t fd_h, int32_t up, int32_t down, int32_t pad_x0, int32_t pad_y0,
                       int32_t s_width_bytes, int32_t s_height, int32_t s_ofs_x, int32_t s_ofs_y, int32_t sw_limit,
                       float gain, float slope, float clamp, int32_t flip_filters, int32_t sign_mode, p3d_stream_t stream);
/* filtered_lrelu_plugin.filtered_lrelu_act_ (filtered_lrelu.cpp:217-296, kernel filtered_lrelu.cu:1109-1213): in place on x (fp16, fp32
 * or fp64):  x = clamp(lrelu(x * gain)).  sign_mode 1 writes s [N][C][H][s_width/4] with s_width = W rounded up to 16 ELEMENTS (code 1
 * when the scaled value is < 0 — not the sign bit: the two kernels differ on -0.0, filtered_lrelu.cu:1140-1149); sign_mode 2 reads s
 * [N][C][s_height][s_width/4] at (x + s_ofs_x, y + s_ofs_y), elements outside pass with the gain only.                              */
int p3d_filtered_lrelu_act(void* x, uint8_t* s, int dtype, const int32_t x_size[4], const int64_t x_stride[4],
                           int32_t s_width, int32_t s_height, int32_t s_ofs_x, int32_t s_ofs_y, float gain, float slope, float clamp,
                           int32_t sign_mode, p3d_stream_t stream);

/* ---- fused tri-plane ray-marcher -----------------------------------------------------------
 * Stands in for the tensor-op pipeline of training/volumetric_rendering:
 *   ImportanceRenderer.forward   renderer.py:88-140   (p3d_render_forward)
 *   ImportanceRenderer.run_model renderer.py:142-148  (p3d_sample_points)
 *   sample_importance/sample_pdf renderer.py:194-253  (p3d_importance_sample, also fused above)
 *   MipRayMarcher2.run_forward   ray_marcher.py:25-57 (fused)
 * with the OSG decoders (training/triplane.py:112-135 one net; training/triplane_cond.py:926-970
 * two nets, density from the second) evaluated on the f32 MFMA path.  Random numbers are inputs
 * (the host draws them exactly where the reference would: renderer.py:190, :237).            */
typedef struct p3d_render_desc {
    int32_t n_img;                        /* N                                                  */
    int32_t rays_per_img;                 /* M (ignored by p3d_sample_points)                   */
    int32_t plane_h, plane_w;             /* tri-plane resolution; 32 channels per plane        */
    int32_t n_nets;                       /* 1: OSGDecoder, 2: OSGDecoder_semantic_lateSeparate */
    int32_t semantic_sigmoid;             /* 2-net decoder: squash the label channels too       */
    int32_t depth_resolution;             /* S_c  rendering_options['depth_resolution']         */
    int32_t depth_resolution_importance;  /* S_f                                                */
    int32_t disparity_space_sampling;
    int32_t white_back;
    float   ray_start, ray_end;           /* used when t_start/t_end are null                   */
    float   box_warp;
    /* plane memory layout, in floats; all 0 = the default [N][3][H][W][32].  Texel (n, p, y, x) starts at
     * n*image_stride + p*plane_stride + (y*W + x)*pixel_stride and holds 32 contiguous channels; e.g. a
     * channels-last backbone output [N][H][W][96] is (image H*W*96, plane 32, pixel 96).                 */
    int64_t image_stride, plane_stride, pixel_stride;
    int32_t raster_order;                 /* != 0: ray m of an image is pixel (m / R, m % R) of an R x R raster (R*R = rays_per_img):
                                             lets the kernel assign 16 x 16 pixel blocks to workgroups for L2 locality  */
    int32_t mlp_bf16x3;                   /* p3d_render_forward only, != 0: the decoder stream comes from p3d_pack_decoder_bf16x3 and the MLPs run
                                             as three bf16 MFMAs per fp32 product (hi/lo splits, fp32 accumulation: ~5e-6 of the hidden range per
                                             layer) instead of f32-input MFMAs at 1/16 of that rate.  0 = exact fp32 (p3d_pack_decoder).
                                             2: the stream comes from p3d_pack_decoder_l1x6 and LAYER 1 of every net runs as six bf16 MFMAs
                                             per product of three-piece splits (fp32-accurate, csrc/bf16_split.h); layer 2 stays on the
                                             f32-input MFMA (its third weight image does not fit the LDS).                                   */
} p3d_render_desc;

int p3d_render_decoder_floats(void);     /* size of the packed decoder stream, in floats        */

/* planes [N][96][H][W] (backbone output, NCHW) -> [N][3][H][W][32]: one 128-byte line per texel */
int p3d_planes_to_channels_last(const float* planes_nchw, float* planes_cl, int32_t n_img, int32_t h, int32_t w,
                                p3d_stream_t stream);

/* FullyConnectedLayer weights (networks_stylegan2.py:96-130; w1 [64,32], b1 [64], w2 [33,64],
 * b2 [33], raw parameters: the lr_mul / sqrt(fan_in) gains are applied here) -> MFMA operand
 * stream.  Net a = colour net, net b = label+density net (null for the single-net decoder).   */
int p3d_pack_decoder(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                     const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                     int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream);
/* the same parameters in the operand order of the bf16x3 decoder (p3d_render_desc.mlp_bf16x3); same size, fp32 values */
int p3d_pack_decoder_bf16x3(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                            const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                            int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream);

/* the same parameters with layer 1 in the bf16 operand order and layer 2 in the f32-MFMA order (p3d_render_desc.mlp_bf16x3 == 2); same size, fp32 values */
int p3d_pack_decoder_l1x6(const float* w1_a, const float* b1_a, const float* w2_a, const float* b2_a,
                          const float* w1_b, const float* b1_b, const float* w2_b, const float* b2_b,
                          int32_t n_nets, float lr_mul, float* packed, p3d_stream_t stream);

/* ray_o, ray_d [N*M][3]; u_coarse [N*M][S_c] and u_fine [N*M][S_f] uniforms in [0,1);
 * t_start/t_end optional per-ray limits [N*M] ('auto' ray range), null otherwise.
 * Outputs: feat [N*M][32*n_nets] (already *2-1), depth [N*M] (clamped to the global sample-depth
 * range, ray_marcher.py:49-50), wsum [N*M].  minmax_ws: 2 x uint32 device scratch.
 * dbg_fine [N*M][S_f] (sorted importance depths) and dbg_wcoarse [N*M][S_c-1] are optional.
 * Returns P3D_ERR_UNSUPPORTED when S_c or S_f exceed 64 (or S_f == 0).
 * Envelope: at most 64 coarse and 1..64 fine samples per ray; clamp_mode 'softplus' (the only mode ray_marcher.py:35 accepts); no
 * density_noise term (renderer.py:116-117: training-time regulariser, 0 in every shipped configuration); the OSG 32-64-33 decoders.
 * Outside it the Python mirror takes the tensor-op formulation and says so with a RuntimeWarning (or raises under fused_policy 'require'). */
int p3d_render_forward(const float* planes_cl, const float* decoder, const float* ray_o, const float* ray_d,
                       const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                       const p3d_render_desc* desc, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                       float* dbg_fine, float* dbg_wcoarse, p3d_stream_t stream);
/* The same launch (same kernel instantiation) with one more optional record: dbg_bins [N*M][S_f] <- for every importance draw, in draw order,
 * the index torch.searchsorted(cdf, u, right=True) returns in sample_pdf (renderer.py:221-253) — the integers behind dbg_fine, so that the index
 * work of the FUSED launch can be compared exactly (tests/test_render_gpu.py), not only that of the stand-alone p3d_importance_sample_index. */
int p3d_render_forward_debug(const float* planes_cl, const float* decoder, const float* ray_o, const float* ray_d,
                             const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                             const p3d_render_desc* desc, float* feat, float* depth, float* wsum, uint32_t* minmax_ws,
                             float* dbg_fine, float* dbg_wcoarse, int32_t* dbg_bins, p3d_stream_t stream);

/* ---- two plane sets: ImportanceSemanticRenderer (training/volumetric_rendering/renderer.py:256-438) --------------------------------
 * The renderer of TriPlaneSemanticGenerator (training/triplane_cond.py:746-758): a texture and a semantic tri-plane set of the same
 * size and layout.  The label decoder (OSGDecoder_semantic, FC 32-64-33) reads the semantic planes' features and gives density +
 * 32 label channels; the colour decoder (OSGDecoder over 64 inputs, FC 64-64-33, density row unused) reads cat(texture, semantic)
 * features (renderer.py:324-333).  Sampling, merging and compositing are ImportanceRenderer's over cat(colour, label): outputs as
 * p3d_render_forward with n_nets = 2 (feat [N*M][64]; desc->semantic_sigmoid squashes the labels too).  Inference only.
 * p3d_pack_decoder_dual: raw FullyConnectedLayer parameters (w1_tex [64][64], b1_tex [64], w2_tex [33][64], b2_tex [33]; w1_sem [64][32],
 * ...) -> p3d_render_decoder_floats_dual() floats.  p3d_sample_points_dual = run_model: rgb [N*P][64] = cat(colour, label), sigma [N*P]. */
int p3d_render_decoder_floats_dual(void);
int p3d_pack_decoder_dual(const float* w1_tex, const float* b1_tex, const float* w2_tex, const float* b2_tex,
                          const float* w1_sem, const float* b1_sem, const float* w2_sem, const float* b2_sem,
                          float lr_mul, float* packed, p3d_stream_t stream);
int p3d_render_forward_dual(const float* planes_tex_cl, const float* planes_sem_cl, const float* decoder_dual, const float* ray_o, const float* ray_d,
                            const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                            const p3d_render_desc* desc, float* feat, float* depth, float* wsum, uint32_t* minmax_ws, p3d_stream_t stream);
int p3d_sample_points_dual(const float* planes_tex_cl, const float* planes_sem_cl, const float* decoder_dual, const float* coords,
                           const p3d_render_desc* desc, int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream);

/* ---- backward of p3d_render_forward (training configs) ------------------------------------------
 * What autograd derives for ImportanceRenderer.forward (renderer.py:88-140) in the reference, as two recomputing launches
 * (csrc/render_bwd.hip): the forward sweep again, driven by g_feat, hands every sample its compositing scalars on a tape; a
 * point-wise pass re-evaluates gather + MLPs per sample and back-propagates on the matrix cores.  Importance depths are constants
 * (renderer.py:198, 211).  Gradients w.r.t. the rays and w.r.t. the depth output are not produced (the training losses use
 * neither; the host falls back to the tensor-op formulation if asked for them).
 *   decoder_bwd : p3d_render_bwd_decoder_floats() floats from p3d_pack_decoder_bwd (same raw parameters as p3d_pack_decoder)
 *   g_feat [N*M][32*n_nets] = dL/dfeat, g_wsum [N*M] = dL/dwsum or null; rays / uniforms / limits exactly as in the forward call
 *   tape_intervals [N*M][S-1][4], tape_samples [N*M][S][4] fp32 scratch (S = S_c + S_f), 16-byte aligned
 *   d_planes_cl [N][3][H][W][32] <- dL/dplanes (channels-last, zeroed here, accumulated with atomics)
 *   d_decoder [p3d_render_grad_decoder_floats()] <- per net (stride 4260 floats): dW1 [64][32] @0, db1 [64] @2048, dW2 [33][64]
 *   @2112, db2 [33] @4224, gradients of the EFFECTIVE weights w * lr_mul / sqrt(fan_in), b * lr_mul (multiply by the same gains
 *   for the raw parameters).                                                                                              */
int p3d_render_bwd_decoder_floats(void);
int p3d_render_grad_decoder_floats(void);
int p3d_pack_decoder_bwd(const float* w1_a, const float* w2_a, const float* w1_b, const float* w2_b, int32_t n_nets, float lr_mul,
                         float* packed_bwd, p3d_stream_t stream);
int p3d_render_backward(const float* planes_cl, const float* decoder, const float* decoder_bwd, const float* ray_o, const float* ray_d,
                        const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                        const p3d_render_desc* desc, const float* g_feat, const float* g_wsum, float* tape_intervals, float* tape_samples,
                        float* d_planes_cl, float* d_decoder, p3d_stream_t stream);

/* coords [N*P][3] -> rgb [N*P][32*n_nets], sigma [N*P]  (G.sample / G.sample_mixed, extract_mesh) */
int p3d_sample_points(const float* planes_cl, const float* decoder, const float* coords, const p3d_render_desc* desc,
                      int32_t pts_per_img, float* rgb, float* sigma, p3d_stream_t stream);
/* Backward of p3d_sample_points: what autograd derives for ImportanceRenderer.run_model (renderer.py:142-148: grid_sample + decoder)
 * when G.sample_mixed is differentiated — the density regularisation of training/loss.py:681-706 ('Greg' phase).  One launch: per
 * point gather + MLPs again, MLP backward on the matrix cores, plane gradient by whole-texel atomics (the point-wise half of
 * p3d_render_backward, no tape).  g_rgb [N*P][32*n_nets] = dL/drgb (post-activation outputs), g_sigma [N*P] = dL/dsigma; either may
 * be null (= zero).  d_planes_cl / d_decoder / decoder_bwd exactly as in p3d_render_backward.  Coordinates receive no gradient.        */
int p3d_sample_points_backward(const float* planes_cl, const float* decoder, const float* decoder_bwd, const float* coords,
                               const p3d_render_desc* desc, int32_t pts_per_img, const float* g_rgb, const float* g_sigma,
                               float* d_planes_cl, float* d_decoder, p3d_stream_t stream);

/* z_coarse [R][S_c], w_coarse [R][S_c-1], u_fine [R][S_f] -> z_fine [R][S_f] (sorted ascending
 * when `sorted`, else in draw order as sample_pdf returns them).                               */
int p3d_importance_sample(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine,
                          int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted,
                          p3d_stream_t stream);

/* The same launch with the INTEGER results of the index work exposed (parity tests: bit-exact against torch.searchsorted / a stable argsort):
 *   bin_index   [R][S_f] int32, optional: per importance draw j the index torch.searchsorted(cdf, u, right=True) returns (renderer.py:240), in draw order;
 *   merge_words [R][4] uint32, optional (needs `sorted`): bit k set <=> sample k of the merged, depth-ordered ray (unify_samples, renderer.py:157-167) is an
 *               importance sample — computed by the merge step the fused ray-marcher runs (csrc/render_device.h: merge_takes_coarse).                       */
int p3d_importance_sample_index(const float* z_coarse, const float* w_coarse, const float* u_fine, float* z_fine, int32_t* bin_index, uint32_t* merge_words,
                                int32_t n_rays, int32_t depth_resolution, int32_t n_importance, int32_t sorted, p3d_stream_t stream);

/* ---- modulated convolution on the matrix cores (fp16 channels-last) --------------------------
 * Stand in for the per-layer inference chain of the StyleGAN2 synthesis layers:
 *   modulated_conv2d, fused branch      training/networks_stylegan2.py:34-69, 81-91
 *   conv2d / conv_transpose2d(stride 2) torch_utils/ops/conv2d_gradfix.py:37-45 via conv2d_resample.py:114-136
 *   noise add + bias_act epilogue       training/networks_stylegan2.py:319-332
 *   ToRGB 1x1 modulated conv            training/networks_stylegan2.py:355-359                    */

/* weight [Co][Ci][taps] fp32 (taps = kh*kw, PyTorch OIHW order), styles [N][Ci] fp32 ->
 * out (dtype fp16 or fp32) = weight * pre_scale * styles (* rsqrt(sum^2 + 1e-8) if demodulate), laid out
 * [N][Co][taps][Ci] (tap-major K, what p3d_conv2d_nhwc consumes) or, with oihw_order != 0, [N][Co][Ci][taps].   */
int p3d_modulate_weights(const float* weight, const float* styles, void* out, int dtype, int32_t n_img, int32_t co, int32_t ci,
                         int32_t taps, int32_t demodulate, float pre_scale, int32_t oihw_order, p3d_stream_t stream);

/* x [N][H][W][Ci], w [N or 1][Co][k*k][Ci] (w_img_stride elements between images, 0 = shared), both `dtype`
 * (P3D_F16: v_mfma_f32_32x32x16_f16; P3D_F32: v_mfma_f32_32x32x2_f32, exact fp32), fp32 accumulation.
 * resample = 0: k x k correlation (k = 1 or 3), "same" padding -> y [N][H][W][Co]; optional epilogue
 *   v = acc + noise[H][W] * noise_strength[0] + bias[co]; act (0 linear, 1 lrelu 0.2); * gain; clamp (< 0 off).
 * resample = 1 (k = 3): conv_transpose2d(stride 2, padding 0) -> y [N][2H+1][2W+1][Co], no epilogue.
 * resample = 2: valid (unpadded) correlation at stride 2 -> y [N][(H-k)/2+1][(W-k)/2+1][Co] with the epilogue: the strided
 * conv conv2d_resample runs after its low-pass FIR in the down-2 layers (conv2d_resample.py:108-111).
 * zeros128: >= 128 bytes of zeros in device memory, 16-byte aligned (source of the border rows of the LDS-DMA
 * staging).  Ci must be a multiple of 64 (fp16) / 32 (fp32), else P3D_ERR_UNSUPPORTED.             */
int p3d_conv2d_nhwc(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                    const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                    int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, p3d_stream_t stream);

/* ---- conv2d_gradfix: training-mode convolution, its data gradient and its weight gradient ---------------------------------------
 * Stand in for the vendor-library calls behind torch_utils/ops/conv2d_gradfix.py (cuDNN there, MIOpen on ROCm):
 *   forward            conv2d_gradfix.py:37-45, 107-131   F.conv2d / F.conv_transpose2d
 *   p3d_conv2d_bwd_data    :139-143   "grad_input = the transposed op" with output_padding from the shapes (:95-104)
 *   p3d_conv2d_bwd_weight  :155-194   Conv2dGradWeight (aten::convolution_backward with mask [F,T,F]; a matmul for 1x1)
 * All tensors channels-last ([N][H][W][C]), fp16 or fp32 with fp32 accumulation (dtype P3D_F32_BF16X3 for forward / bwd_data: fp32 tensors,
 * products as three bf16 MFMAs; P3D_F32_BF16X6 for all three: fp32 tensors, fp32-accurate products as six bf16 MFMAs — the weight gradient
 * takes it for whole 128 x 128 tiles and the exact kernels otherwise), groups = 1, dilation = 1, weights SHARED across the
 * batch and passed in torch's own layout and in the activation dtype.  The family (what conv2d_resample.py:96-136 ever asks for):
 *   transposed = 0: conv2d,            weight [Co][Ci][k][k]:  k in {1, 3} at stride 1 / padding k/2,  k = 3 at stride 2 / padding 0
 *   transposed = 1: conv_transpose2d,  weight [Ci][Co][k][k]:  the same two geometries; at stride 2 the output is [2H+1 | 2H+2]
 *                   (out_h / out_w; 0 = 2H+1: output_padding 0)
 * ci = channels of x, co = channels of y in every call.  w_scratch: co*ci*k*k elements of the activation dtype, 16-byte aligned (the
 * tap-major re-layout the matrix-core kernels read; unused by the skinny 1x1 route, may then be null).  zeros128 as p3d_conv2d_nhwc.
 * Returns P3D_ERR_UNSUPPORTED when ci is not a multiple of 64 (fp16) / 32 (fp32) on the 3x3 routes: pad the channels.               */
/* workspace: optional device scratch for the split-K schedule of low-resolution layers (a 512-channel 3x3 layer at 4^2 .. 32^2 is a
 * handful of output tiles with a 144-step K loop: its K steps are dealt to many work-groups and a second launch sums them);
 * p3d_conv2d_forward_workspace(...) says how many bytes a call would use (0: none), 16-byte aligned; null = never split.            */
int p3d_conv2d_forward(const void* x, const void* weight, void* y, void* w_scratch, const void* zeros128, int dtype,
                       int32_t n_img, int32_t h, int32_t w, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                       int32_t transposed, int32_t out_h, int32_t out_w, void* workspace, int64_t workspace_bytes, p3d_stream_t stream);
int64_t p3d_conv2d_forward_workspace(int dtype, int32_t n_img, int32_t h, int32_t w, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                                     int32_t transposed);
/* gy [N][gy_h][gy_w][co] -> gx [N][x_h][x_w][ci], for the FORWARD op (ci -> co, weight, kernel_size, stride, transposed) described above;
 * workspace as p3d_conv2d_forward_workspace(dtype, n_img, gy_h, gy_w, co, ci, kernel_size, stride, !transposed)                      */
int p3d_conv2d_bwd_data(const void* gy, const void* weight, void* gx, void* w_scratch, const void* zeros128, int dtype,
                        int32_t n_img, int32_t gy_h, int32_t gy_w, int32_t ci, int32_t co, int32_t kernel_size, int32_t stride,
                        int32_t transposed, int32_t x_h, int32_t x_w, void* workspace, int64_t workspace_bytes, p3d_stream_t stream);
/* gw[cs][cb][ky][kx] = sum_{n,i,j} small[n,i,j,cs] * big[n, i*stride + ky - pad, j*stride + kx - pad, cb]   (out-of-image = 0)
 * conv2d:           small = gy (cs = Co), big = x  (cb = Ci)  ->  gw [Co][Ci][k][k]
 * conv_transpose2d: small = x  (cs = Ci), big = gy (cb = Co)  ->  gw [Ci][Co][k][k]        (the roles swap, conv2d_gradfix.py:173)
 * gw in the activation dtype.  workspace: p3d_conv2d_bwd_weight_workspace(...) bytes of device scratch, 16-byte aligned (fp32
 * partial sums of the split-K work-groups; deterministic: no atomics).                                                             */
int64_t p3d_conv2d_bwd_weight_workspace(int dtype, int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t c_big, int32_t kernel_size);
int p3d_conv2d_bwd_weight(const void* small_img, const void* big_img, void* gw, void* workspace, int64_t workspace_bytes, int dtype,
                          int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t big_h, int32_t big_w, int32_t c_big,
                          int32_t kernel_size, int32_t stride, int32_t pad, p3d_stream_t stream);
/* The same sums, written as fp32 whatever the activation dtype and multiplied by `scale` on the way out — the gradient of a PARAMETER that entered the
 * convolution as (weight * gain).to(activation dtype) (Conv2dLayer, networks_stylegan2.py:177-180: the cast's and the gain's gradients in the final pass
 * over the partial sums instead of two more launches).  gw_f32 [cs][cb][k][k] float.                                                                  */
int p3d_conv2d_bwd_weight_scaled(const void* small_img, const void* big_img, float* gw_f32, void* workspace, int64_t workspace_bytes, int dtype,
                                 int32_t n_img, int32_t small_h, int32_t small_w, int32_t c_small, int32_t big_h, int32_t big_w, int32_t c_big,
                                 int32_t kernel_size, int32_t stride, int32_t pad, float scale, p3d_stream_t stream);

/* p3d_conv2d_nhwc with optional split-K scratch (see p3d_conv2d_forward): workspace of p3d_conv2d_nhwc_workspace(...) bytes, or null */
int p3d_conv2d_nhwc_ws(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                       const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                       int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, void* workspace, int64_t workspace_bytes,
                       p3d_stream_t stream);
/* The last 3x3 layer of a synthesis block and its ToRGB in one launch (fp16, Co = 128 or 256: SynthesisBlock.conv1 + .torgb + the skip-image
 * sum, training/networks_stylegan2.py:449-459): y = act(conv3x3(x, w) + bias) * gain, clamped, as p3d_conv2d_nhwc would write it, and
 *   rgb_out[n][o][pixel] += clamp(sum_c y[n][pixel][c] * rgb_w[n][o][c] + rgb_bias[o], rgb_clamp)        (rgb_out fp32 NCHW, o < rgb_co <= 8)
 * from the finished tile while it is still in LDS.  rgb_w = ToRGB weight * styles, fp32 [N][rgb_co][Co].  No noise input (layers with
 * noise keep the two-launch form).  Co = 256: a work-group walks both 128-channel blocks of its pixel patch and sums the two
 * contractions before bias / clamp.  Any other Co, Ci % 64 != 0 or images under 32 x 32: P3D_ERR_UNSUPPORTED.
 * y may be NULL: the activations are then not stored at all — the LAST block of a super-resolution head returns x to a caller that drops
 * it (training/superresolution.py:297-354 return rgb only), so its 268 MB of fp16 activations per launch have the ToRGB as only reader.  */
int p3d_conv3x3_torgb_f16(const void* x, const void* w, void* y, const float* bias, const void* zeros128, const float* rgb_w, const float* rgb_bias,
                          float* rgb_out, int32_t rgb_co, float rgb_clamp, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                          int64_t w_img_stride, int32_t act, float gain, float clamp, p3d_stream_t stream);

/* The LAST 3x3 layer of the tri-plane backbone and its wide ToRGB + skip-image sum in one launch (bf16x3 on split activations, Co = 128: SynthesisBlock.conv1 +
 * .torgb + the upsampled predecessor image, training/networks_stylegan2.py:449-459, for a block whose x the network drops — SynthesisNetwork.forward returns img only,
 * :511-528).  x_split / w_split as p3d_conv2d_nhwc_bf16x3_io takes them (x_split = 1); the layer's activations y = clamp(act(conv3x3 + noise + bias) * gain) are formed in
 * registers, split into (hi, lo) as their stored form would have been, and contracted with rgb_wmod_split ([N][rgb_co][1][Co] split K rows: p3d_modulate_weights,
 * P3D_F32_BF16X3, no demodulation) as p3d_torgb_wide_split does:
 *   img[n][pixel][o] = clamp(sum_c y[n][pixel][c] * rgb_w[n][o][c] + rgb_bias[o], rgb_clamp) + upsample2d(prev, f)[n][pixel][o]       (img, prev fp32 NHWC)
 * y itself is never written.  prev (and f4x4_host: sixteen HOST floats, the 4 x 4 filter) may be NULL: no skip term.  rgb_co in {32, 64, 96}; Co != 128, Ci % 32 != 0,
 * images under 32 x 32 or odd-sized, or fewer than 192 patches of 16 x 16 pixels: P3D_ERR_UNSUPPORTED (the caller keeps the two-launch form).                          */
int p3d_conv3x3_torgb_split(const void* x_split, const void* w_split, const float* bias, const float* noise, const float* noise_strength, const void* zeros128,
                            const void* rgb_wmod_split, const float* rgb_bias, float* img_nhwc, const float* prev_nhwc, const float* f4x4_host,
                            int32_t rgb_co, float rgb_clamp, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                            int32_t act, float gain, float clamp, p3d_stream_t stream);

/* p3d_conv2d_nhwc_ws with a per-(image, output channel) factor on the accumulator, ahead of noise / bias / activation: out_scale fp32
 * [N][Co].  With p3d_demod_coefs and p3d_bcast_fma this is the SHARED-weight form of the modulated convolution
 * (training/networks_stylegan2.py:70-79: x * styles -> convolution with the unmodulated weights -> * demodulation coefficients), which for
 * the low-resolution layers of a batch reads one weight tensor instead of one per image.  fp32 tensors (P3D_F32 / P3D_F32_BF16X3). */
int p3d_conv2d_nhwc_scaled(const void* x, const void* w, void* y, int dtype, const float* out_scale, const float* bias, const float* noise,
                           const float* noise_strength, const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                           int64_t w_img_stride, int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, void* workspace,
                           int64_t workspace_bytes, p3d_stream_t stream);
/* ... and with the style modulation of that form applied on the way INTO the matrix cores: in_scale fp32 [N][Ci] multiplies every activation as it is
 * split for the MFMAs (the same fp32 product `x * styles` a separate pass would have stored: results are bit-identical to p3d_bcast_fma followed by
 * p3d_conv2d_nhwc_scaled), so the low-resolution layers lose one launch each.  dtype P3D_F32_BF16X3 only; P3D_ERR_UNSUPPORTED when the images one
 * 128-row tile touches times Ci exceed the kernel's 2048-float table (the caller then scales x itself).                                          */
int p3d_conv2d_nhwc_scaled_in(const void* x, const void* w, void* y, int dtype, const float* in_scale, const float* out_scale, const float* bias,
                              const float* noise, const float* noise_strength, const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci,
                              int32_t co, int64_t w_img_stride, int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp,
                              void* workspace, int64_t workspace_bytes, p3d_stream_t stream);
/* Activations that stay split between the bf16x3 layers of an inference pass (training/networks_stylegan2.py:436-459: conv0 -> conv1 -> ToRGB /
 * the next block, each a modulated_conv2d :26-105 whose fp32 products this library forms as three bf16 MFMAs).  x_split != 0: x is NOT fp32 but,
 * per pixel and 32 channels, [32 x bf16 hi | 32 x bf16 lo] in the same 128 bytes (hi = bf16(v), lo = bf16(v - hi): exactly what the kernels
 * otherwise compute in registers for every tap) — as written by a call with y_split != 0 or by p3d_fir4_bias_act_nhwc_split.  The same products
 * as the plain-tensor calls: bit-identical where the same kernel runs; 3x3 'same' layers with x_split whose 16 x 16-patch grid fills the chip take a
 * ring-pipeline kernel that sums them in another order (<= 1e-6 of the range apart).  dtype is implied (P3D_F32_BF16X3: w from p3d_modulate_weights in that layout).  x_split is taken by
 * every route (3x3, 1x1, transposed, stride 2); y_split only by the 3x3 'same' layers the halo-slab / ring kernels run (Co % 32 == 0, an image of at
 * least 8 x 16 whose own grid fills the chip): otherwise P3D_ERR_UNSUPPORTED and nothing is launched — ask again with y_split = 0.           */
int p3d_conv2d_nhwc_bf16x3_io(const void* x, const void* w, void* y, const float* bias, const float* noise, const float* noise_strength,
                              const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                              int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t x_split, int32_t y_split,
                              void* workspace, int64_t workspace_bytes, p3d_stream_t stream);
/* The route p3d_conv2d_nhwc_bf16x3_io would take for these sizes, without launching anything: returns 1 when a split result would be granted
 * (want_y_split != 0 and the 3x3 halo-slab / ring kernel takes the layer), 0 when the result will be a plain tensor, or a negative error code;
 * *workspace_bytes = the split-K scratch that route wants (0: none).  Ask once per geometry, then call with the granted flag.                 */
int p3d_conv2d_nhwc_bf16x3_io_plan(int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride, int32_t kernel_size,
                                   int32_t resample, int32_t x_split, int32_t want_y_split, int64_t* workspace_bytes);
/* Wide ToRGB + skip-image sum of a synthesis block in one pass, on split activations (training/networks_stylegan2.py:355-359 ToRGBLayer and :453-459
 * img = upsample2d(img) + y):  y[n][p][o] = clamp(sum_c x[n][p][c] * wmod[n][o][c] + bias[o]) + (prev ? upfirdn2d(prev, f, up = 2, pad 2, gain 4)[n][p][o] : 0).
 * x_split [N][H][W][Ci] and wmod_split [N][Co][Ci] in the split K-row layout above (wmod: p3d_modulate_weights(..., demodulate = 0, P3D_F32_BF16X3));
 * y [N][H][W][Co] and prev [N][H/2][W/2][Co] fp32 channels-last; f4x4_host: the 16 filter taps in HOST memory, row-major (read when prev != null).
 * The skip term is summed exactly as p3d_upfirdn2d_acc sums it.  Ci in {128, 256}, Co in {32, 64, 96}, W % 32 == 0, else P3D_ERR_UNSUPPORTED.  */
int p3d_torgb_wide_split(const void* x_split, const void* wmod_split, const float* bias, float* y_nhwc, const float* prev_nhwc, const float* f4x4_host,
                         int32_t n_img, int32_t h, int32_t w, int32_t ci, int32_t co, float clamp, p3d_stream_t stream);
/* d[n][o] = rsqrt(sum_i styles[n][i]^2 * w2[o][i] + 1e-8) with w2[o][i] = sum over the taps of weight[o][i][.]^2 (networks_stylegan2.py:57-63) */
int p3d_demod_coefs(const float* styles, const float* w2, float* d, int32_t n_rows, int32_t ci, int32_t co, p3d_stream_t stream);
/* Several layers at once (one launch): job j writes d_j [n_rows][co_j] from styles_j [n_rows][ci_j] and w2_j [co_j][ci_j]; all jobs share n_rows.  The
 * job array is HOST memory, copied into the kernel arguments.  Results are bit-identical to p3d_demod_coefs per job.                          */
#define P3D_DEMOD_MAX_JOBS 24
typedef struct p3d_demod_job { const float* styles; const float* w2; float* d; int32_t ci, co; } p3d_demod_job;
int p3d_demod_coefs_multi(const p3d_demod_job* jobs_host, int32_t n_jobs, int32_t n_rows, p3d_stream_t stream);
/* Its gradient for the training passes: given gd = dL/dd [N][Co] and the forward's d, writes gs = dL/dstyles [N][Ci] and gw = dL/dweight [Co][Ci][taps]
 * (weight: the fp32 [Co][Ci][taps] tensor w2 was summed from); either output may be null.  The reference gets these from autograd through
 * (w * s).square().sum().rsqrt() on the [N][Co][Ci][k][k] product (networks_stylegan2.py:57-63).                                                    */
int p3d_demod_coefs_backward(const float* gd, const float* d, const float* styles, const float* w2, const float* weight, float* gs, float* gw,
                             int32_t n_rows, int32_t ci, int32_t co, int32_t taps, p3d_stream_t stream);

int64_t p3d_conv2d_nhwc_workspace(int dtype, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride, int32_t kernel_size,
                                  int32_t resample);

/* x [N][HW][Ci] fp16 channels-last, weight [Co][Ci] fp32, styles [N][Ci] fp32 (weight gain already applied),
 * bias [Co] or null -> y [N][Co][HW] fp32 (NCHW); accumulate != 0 adds into y (the skip-image sum).
 * Ci in {64, 128, 256}, Co <= 32, HW a multiple of 4, else P3D_ERR_UNSUPPORTED (wide outputs: p3d_conv2d_nhwc, k = 1). */
int p3d_torgb_nhwc_f16(const void* x, const float* weight, const float* styles, const float* bias, float* y_nchw,
                       int32_t n_img, int32_t hw, int32_t ci, int32_t co, float clamp, int32_t accumulate, p3d_stream_t stream);

/* ---- per-(image, channel) scaling and its gradient reductions ----------------------------------
 * The element-wise half of the unfused modulated convolution of the training passes (training/networks_stylegan2.py:70-79:
 * x * styles before the convolution, fma(y, dcoefs, noise) after it; torch_utils/ops/fma.py:17-60) and the bias-gradient sums of
 * bias_act (bias_act.py:190-193).  A dense activation tensor is passed as [n][a][b], b contiguous: NCHW -> channels_last = 0, a = C,
 * b = H*W; channels-last -> channels_last = 1, a = H*W, b = C.  dtype fp16 or fp32, fp32 arithmetic, one rounding; b must be a multiple
 * of 8 (fp16) / 4 (fp32) else P3D_ERR_UNSUPPORTED.
 *   p3d_bcast_fma:    y = x * scale[n][c] + z[z_per_image ? n : 0][pixel]      scale fp32 [n][C]; z in x's dtype, [1 or n][H*W], or null
 *   p3d_channel_dot:  out[n][c] = sum over the pixels of p * q  (q null: of p)  fp32 out; channels-last needs a workspace of
 *                     p3d_channel_dot_workspace(...) bytes (partial sums of 256-row chunks; deterministic, no atomics)
 *   p3d_pixel_sum:    out[n][pixel] = sum over the channels of p               fp32 out                                          */
int p3d_bcast_fma(const void* x, const float* scale, const void* z, void* y, int dtype, int32_t channels_last, int32_t n, int32_t a, int32_t b,
                  int32_t z_per_image, p3d_stream_t stream);
int64_t p3d_channel_dot_workspace(int32_t channels_last, int32_t n, int32_t a, int32_t b);
int p3d_channel_dot(const void* p, const void* q, float* out, void* workspace, int64_t workspace_bytes, int dtype, int32_t channels_last,
                    int32_t n, int32_t a, int32_t b, p3d_stream_t stream);
int p3d_pixel_sum(const void* p, float* out, int dtype, int32_t channels_last, int32_t n, int32_t a, int32_t b, p3d_stream_t stream);

/* ---- the x2 synthesis layer in one launch (fp16) -----------------------------------------------
 * conv_transpose2d(stride 2, 3x3; torch_utils/ops/conv2d_resample.py:114-127) -> 4x4 low-pass, pad 1 (:128) -> + noise -> + bias ->
 * lrelu(0.2) * act_gain -> clamp (training/networks_stylegan2.py:319-332), i.e. p3d_conv2d_nhwc(resample = 1) followed by
 * p3d_fir4_bias_act_nhwc without the [N][2H+1][2W+1][Co] intermediate in memory.  x [N][H][W][Ci] fp16, w [N or 1][Co][9][Ci] fp16
 * (as for p3d_conv2d_nhwc), y [N][2H][2W][Co] fp16.  fir_yx_host: EIGHT floats in HOST memory, fy[0..3] then fx[0..3], the separable
 * filter in correlation order with its gain folded in: out[oy][ox] = sum fy[a] fx[b] ct[oy - 1 + a][ox - 1 + b].  conv_gain scales the
 * transposed conv's output before it is rounded to fp16.  act: 0 linear, 1 lrelu(0.2); bias / noise may be null; clamp < 0 = off.
 * Ci and Co must be multiples of 32, else P3D_ERR_UNSUPPORTED (callers then use the two-launch form).                       */
int p3d_up2_fir_f16(const void* x, const void* w, void* y, const void* zeros128, const float* bias, const float* noise,
                    const float* noise_strength, const float* fir_yx_host, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                    int64_t w_img_stride, float conv_gain, int32_t act, float act_gain, float clamp, p3d_stream_t stream);
/* The same layer for fp32 tensors in the bf16x3 formulation (P3D_F32_BF16X3): w = p3d_modulate_weights(..., dtype P3D_F32_BF16X3)
 * [N or 1][Co][9][Ci] in its [32 x hi | 32 x lo] K rows, x / y fp32.  Arguments as p3d_up2_fir_f16.                        */
int p3d_up2_fir_bf16x3(const void* x, const void* w, void* y, const void* zeros128, const float* bias, const float* noise,
                       const float* noise_strength, const float* fir_yx_host, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                       int64_t w_img_stride, float conv_gain, int32_t act, float act_gain, float clamp, p3d_stream_t stream);

/* ---- 4x4 FIR + layer epilogue, channels-last --------------------------------------------------
 * The tail of every x2 synthesis layer in one pass: upfirdn2d(up = down = 1, 4x4 filter f [4][4] fp32 contiguous,
 * gain) (torch_utils/ops/conv2d_resample.py:128) followed by "+ noise" and bias_act (networks_stylegan2.py:319-332):
 *   y = clamp(act(fir(x) + noise[out_h][out_w] * noise_strength[0] + bias[c]) * act_gain)
 * x [N][in_h][in_w][C] -> y [N][out_h][out_w][C], dtype fp16 or fp32, out = in + pad0 + pad1 - 3 (pad1 implied by the
 * sizes).  act: 1 linear, 3 lrelu(alpha) (bias_act.py:23-33 indices); bias / noise may be null; clamp < 0 = off.
 * C must be a multiple of 64 (fp16) / 32 (fp32), else P3D_ERR_UNSUPPORTED.                              */
int p3d_fir4_bias_act_nhwc(const void* x, const float* f, void* y, int dtype, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                           int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                           const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                           float clamp, p3d_stream_t stream);
/* The same pass on fp32 input with the result written in the split layout of p3d_conv2d_nhwc_bf16x3_io (C % 32 == 0): the x2 layer's output goes
 * to the block's next bf16x3 convolution without an fp32 copy of it ever existing.                                                        */
int p3d_fir4_bias_act_nhwc_split(const void* x, const float* f, void* y, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                                 int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                                 const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                                 float clamp, p3d_stream_t stream);

/* ---- low-resolution layers and style affines: launch-count diet (csrc/small_ops.hip) -----------
 * p3d_fc_forward: FullyConnectedLayer.forward (training/networks_stylegan2.py:113-127) in one launch:
 *   y[n][o] = act((sum_i x[n][i] * w[o][i]) * weight_gain + b[o] * bias_gain) * act_gain * out_scale
 * x [n_rows][in] (n_rows <= 16, rows x_row_stride elements apart: a column of the ws tensor is read in place), w [out][in],
 * b [out] or null, fp32; act 1 = linear, 3 = lrelu(alpha).
 * out_scale carries a constant the caller multiplies the result by (ToRGBLayer's weight_gain, :356).
 *
 * p3d_im2col3x3: the 3x3 patch matrix of the whole batch,
 *   cols[n][c * 9 + ky * 3 + kx][oy * ow + ox] = x[n, c, oy * stride + ky - pad, ox * stride + kx - pad] (0 outside),
 * oh = (h + 2 pad - 3) / stride + 1.  pad 1 / stride 1 = torch.nn.functional.unfold(x, 3, padding=1); pad 0 / stride 2 feeds the
 * valid stride-2 conv of the down-2 layers (conv2d_resample.py:108-111).  x fp32 addressed by ELEMENT strides (NCHW or
 * channels-last storage), cols fp32 contiguous.
 *
 * p3d_noise_bias_act: the tail of SynthesisLayer.forward after the convolution (:326-332) on y [n_img][c][hw] fp32:
 *   y = clamp(act(x + noise[hw] * noise_strength[0] + bias[c]) * gain);  hw % 4 == 0; x may equal y; noise / bias may be null. */
int p3d_fc_forward(const float* x, const float* w, const float* b, float* y, int32_t n_rows, int32_t in_features, int32_t out_features,
                   int64_t x_row_stride, float weight_gain, float bias_gain, int32_t act, float alpha, float act_gain, float out_scale, p3d_stream_t stream);

/* Several FullyConnectedLayer evaluations on the same number of rows in ONE launch (the style affines of a synthesis network:
 * SynthesisLayer.affine / ToRGBLayer.affine, training/networks_stylegan2.py:305, 352): jobs_host is a HOST array, its contents travel in
 * the kernel arguments.  Each job is p3d_fc_forward's argument list.  At most P3D_FC_MAX_JOBS jobs.                                  */
#define P3D_FC_MAX_JOBS 40
typedef struct p3d_fc_job {
    const float* x; const float* w; const float* b; float* y;
    int64_t x_row_stride;
    int32_t in_features, out_features;
    float weight_gain, bias_gain;
    int32_t act; float alpha, act_gain, out_scale;
} p3d_fc_job;
int p3d_fc_multi(const p3d_fc_job* jobs_host, int32_t n_jobs, int32_t n_rows, p3d_stream_t stream);
int p3d_im2col3x3(const float* x, float* cols, int32_t n_img, int32_t c, int32_t h, int32_t w, int32_t pad, int32_t stride,
                  int64_t stride_n, int64_t stride_c, int64_t stride_y, int64_t stride_x, p3d_stream_t stream);
int p3d_noise_bias_act(const float* x, float* y, const float* noise, const float* noise_strength, const float* bias, int32_t n_img,
                       int32_t c, int32_t hw, int32_t act, float alpha, float gain, float clamp, p3d_stream_t stream);

/* RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-62) in one launch: cam2world [n_cam][4][4], intrinsics
 * [n_cam][3][3] (normalised fx, fy, cx, cy, skew), fp32 contiguous -> origins, dirs [n_cam][R*R][3] fp32, rays row-major,
 * directions unit length (norm clamped at 1e-12 like F.normalize).                                                        */
int p3d_ray_sample(const float* cam2world, const float* intrinsics, float* origins, float* dirs, int32_t n_cam, int32_t resolution,
                   p3d_stream_t stream);
/* The same straight from the 25-float camera labels the generators take (training/triplane.py:57-60: c[:, :16] is cam2world, c[:, 16:25] the
 * intrinsics): labels [n_cam][label_stride] fp32, label_stride >= 25 floats between rows — no contiguous copies of the two slices.       */
int p3d_ray_sample_labels(const float* labels, int64_t label_stride, float* origins, float* dirs, int32_t n_cam, int32_t resolution, p3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* P3D_HIP_H */
