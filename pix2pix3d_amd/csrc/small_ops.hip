// Launch-count diet for the low-resolution half of the synthesis network (4^2 .. 32^2) and the style affines.
// These layers hold almost no arithmetic — at batch 4 they are pure launch latency: the torch route spent ~14 launches on
// one 3x3 layer (weight scaling, addmm, layout copies, per-sample im2col, bmm, noise add, bias_act) and the whole group
// cost more than the ray-marcher.  Here:
//   p3d_fc_forward      : FullyConnectedLayer (networks_stylegan2.py:96-130) in one launch — weight gain, bias gain,
//                         activation and an output scale (ToRGB's weight_gain, :356) folded in.
//   p3d_im2col3x3       : the 3x3 / pad 1 patch matrix of the whole batch in one launch, from any NCHW-indexable layout.
//   p3d_noise_bias_act  : + noise * strength, + bias, activation, gain, clamp on a [N, C, HW] tensor in one pass
//                         (networks_stylegan2.py:326-332 after the GEMM).
#include "p3d_common.h"
#include "../../include/p3d_hip.h"

namespace p3d {

// ---- fully connected: y[n, o] = act((x[n, :] . W[o, :]) * wg + b[o] * bg) * act_gain * out_scale ---------------------
// One wave per output feature: the wave streams W[o, :] once (coalesced float4), keeps the (few) input rows in LDS and
// reduces with DPP-free shuffles.  n_rows <= 16.
constexpr int FC_MAXN = 16;
__device__ __forceinline__ void fc_block(float* xs, int block, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                         float* __restrict__ y, int n_rows, int in_f, int out_f, int64_t x_stride, float wg, float bg,
                                         int act, float alpha, float act_gain, float out_scale)
{
#pragma unroll 4
    for (int e = threadIdx.x * 4; e < n_rows * in_f; e += 256 * 4) {
        const int n = e / in_f, k = e - n * in_f;
        *(float4*)(xs + e) = *(const float4*)(x + n * x_stride + k);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = block * 4 + wave;
    if (o >= out_f) return;
    float acc[FC_MAXN];
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) acc[n] = 0.f;
    const float* wr = w + (int64_t)o * in_f;
    const float bias_raw = b ? b[o] : 0.f;                          // (requested with the weights, not behind the reduction)
#pragma unroll 4
    for (int k = lane * 4; k < in_f; k += 256) {
        const float4 wv = *(const float4*)(wr + k);
#pragma unroll
        for (int n = 0; n < FC_MAXN; ++n) {
            if (n < n_rows) {
                const float4 xv = *(const float4*)(xs + n * in_f + k);
                acc[n] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[n]))));
            }
        }
    }
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) {
        if (n < n_rows) {
            float v = acc[n];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
            acc[n] = v;
        }
    }
    if (lane == 0) {
        const float bias = b ? bias_raw * bg : 0.f;
#pragma unroll
        for (int n = 0; n < FC_MAXN; ++n) {
            if (n < n_rows) {
                float v = fmaf(acc[n], wg, bias);
                if (act == 3) v = v > 0.f ? v : v * alpha;
                y[(int64_t)n * out_f + o] = v * act_gain * out_scale;
            }
        }
    }
}
__global__ void __launch_bounds__(256) fc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                 float* __restrict__ y, int n_rows, int in_f, int out_f, int64_t x_stride, float wg, float bg,
                                                 int act, float alpha, float act_gain, float out_scale)
{
    extern __shared__ float xs[];                                   // [n_rows][in_f]
    fc_block(xs, blockIdx.x, x, w, b, y, n_rows, in_f, out_f, x_stride, wg, bg, act, alpha, act_gain, out_scale);
}

// Demodulation coefficients of the shared-weight form of the modulated convolution (networks_stylegan2.py:57-63 with the sum over the
// taps taken first): d[n][o] = rsqrt(sum_i styles[n][i]^2 * w2[o][i] + 1e-8), w2[o][i] = sum_taps w[o][i][t]^2.  One wave per output channel.
__global__ void __launch_bounds__(256) demod_coefs_kernel(const float* __restrict__ styles, const float* __restrict__ w2, float* __restrict__ d, int n_rows, int ci, int co)
{
    extern __shared__ float xs[];                                   // [n_rows][ci] of styles^2
    for (int e = threadIdx.x; e < n_rows * ci; e += 256) { const float v = styles[e]; xs[e] = v * v; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave;
    if (o >= co) return;
    float acc[FC_MAXN];
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) acc[n] = 0.f;
    for (int k = lane; k < ci; k += 64) {
        const float wv = w2[(int64_t)o * ci + k];
#pragma unroll
        for (int n = 0; n < FC_MAXN; ++n)
            if (n < n_rows) acc[n] = fmaf(wv, xs[n * ci + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) {
        if (n < n_rows) {
            float v = acc[n];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
            if (lane == 0) d[(int64_t)n * co + o] = rsqrtf(v + 1e-8f);
        }
    }
}

// Several layers' coefficients in one launch (device inference: every shared-weight layer of a network depends on the latents alone, so the host issues
// them together ahead of the convolutions: modconv.premodulate_many): block b serves job j with first_block[j] <= b < first_block[j + 1].
struct DemodJobs { p3d_demod_job job[P3D_DEMOD_MAX_JOBS]; int first_block[P3D_DEMOD_MAX_JOBS + 1]; int njobs; int n_rows; };
__global__ void __launch_bounds__(256) demod_coefs_multi_kernel(DemodJobs a)
{
    extern __shared__ float xs[];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_block[j + 1]) ++j;
    const p3d_demod_job& q = a.job[j];
    const int ci = q.ci, co = q.co, n_rows = a.n_rows;
#pragma unroll 8
    for (int e = threadIdx.x; e < n_rows * ci; e += 256) { const float v = q.styles[e]; xs[e] = v * v; }      // (unrolled: rolled, every iteration was a memory round trip of its own)
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = ((int)blockIdx.x - a.first_block[j]) * 4 + wave;
    if (o >= co) return;
    float acc[FC_MAXN];
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) acc[n] = 0.f;
#pragma unroll 8
    for (int k = lane; k < ci; k += 64) {                           // (demod_coefs_kernel's arithmetic, operation for operation: the results are bit-identical)
        const float wv = q.w2[(int64_t)o * ci + k];
#pragma unroll
        for (int n = 0; n < FC_MAXN; ++n)
            if (n < n_rows) acc[n] = fmaf(wv, xs[n * ci + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) {
        if (n < n_rows) {
            float v = acc[n];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
            if (lane == 0) q.d[(int64_t)n * co + o] = rsqrtf(v + 1e-8f);
        }
    }
}

// Gradient of the demodulation coefficients (training passes).  With t[n][o] = gd[n][o] * d[n][o]^3:
//   gs[n][i]      = -styles[n][i] * sum_o t[n][o] * w2[o][i]                     (demod_bwd_styles_kernel: a thread per input channel, four slices of o per block)
//   gw[o][i][tap] = -weight[o][i][tap] * sum_n t[n][o] * styles[n][i]^2          (demod_bwd_weight_kernel: a thread per weight element)
// (d = u^-1/2 with u = sum_i s^2 w2 + eps: dd/du = -d^3 / 2, du/ds = 2 s w2, du/dw = 2 w s^2.)
__global__ void __launch_bounds__(256) demod_bwd_styles_kernel(const float* __restrict__ gd, const float* __restrict__ d, const float* __restrict__ styles,
                                                               const float* __restrict__ w2, float* __restrict__ gs, int n_rows, int ci, int co)
{
    extern __shared__ float ts[];                                   // [n_rows][co] of t, then [4][64][n_rows] partial sums
#pragma unroll 4
    for (int e = threadIdx.x; e < n_rows * co; e += 256) { const float dv = d[e]; ts[e] = gd[e] * dv * dv * dv; }
    __syncthreads();
    const int il = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float acc[FC_MAXN];
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n) acc[n] = 0.f;
    if (i < ci)
        // (unrolled by sixteen: the launch is ci / 64 blocks whose threads each walk co / 4 weight rows — rolled, 128 dependent memory round trips, 67 us per launch and 117
        // launches per three training iterations: profiles/round6_zk_kernel_pmc_train6.txt; the sums are formed in the same order)
#pragma unroll 16
        for (int o = slice; o < co; o += 4) {
            const float wv = w2[(int64_t)o * ci + i];
#pragma unroll
            for (int n = 0; n < FC_MAXN; ++n)
                if (n < n_rows) acc[n] = fmaf(ts[n * co + o], wv, acc[n]);
        }
    __syncthreads();                                                // t is done with: the partial sums take its place
    float* const part = ts;
#pragma unroll
    for (int n = 0; n < FC_MAXN; ++n)
        if (n < n_rows) part[(slice * 64 + il) * n_rows + n] = acc[n];
    __syncthreads();
    if (slice == 0 && i < ci)
        for (int n = 0; n < n_rows; ++n) {
            const float sum = (part[il * n_rows + n] + part[(64 + il) * n_rows + n]) + (part[(128 + il) * n_rows + n] + part[(192 + il) * n_rows + n]);
            gs[(int64_t)n * ci + i] = -styles[(int64_t)n * ci + i] * sum;
        }
}

__global__ void __launch_bounds__(256) demod_bwd_weight_kernel(const float* __restrict__ gd, const float* __restrict__ d, const float* __restrict__ styles,
                                                               const float* __restrict__ weight, float* __restrict__ gw, int n_rows, int ci, int co, int taps)
{
    const int64_t total = (int64_t)co * ci * taps;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t oi = e / taps;
        const int o = (int)(oi / ci), i = (int)(oi - (int64_t)o * ci);
        float coef = 0.f;
#pragma unroll 4
        for (int n = 0; n < n_rows; ++n) {
            const float dv = d[n * co + o], sv = styles[n * ci + i];
            coef = fmaf(gd[n * co + o] * dv * dv * dv, sv * sv, coef);
        }
        gw[e] = -weight[e] * coef;
    }
}

// Several independent FC layers in ONE launch (the 20 style affines of a synthesis network are 20 launches of ~6 us each otherwise):
// the jobs travel in the kernel arguments; first_block[j] is the first block of job j.
constexpr int FC_MAX_JOBS = P3D_FC_MAX_JOBS;
struct FcJobs { p3d_fc_job job[FC_MAX_JOBS]; int first_block[FC_MAX_JOBS + 1]; int njobs; int n_rows; };
__global__ void __launch_bounds__(256) fc_multi_kernel(FcJobs a)
{
    extern __shared__ float xs[];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_block[j + 1]) ++j;
    const p3d_fc_job& q = a.job[j];
    fc_block(xs, blockIdx.x - a.first_block[j], q.x, q.w, q.b, q.y, a.n_rows, q.in_features, q.out_features, q.x_row_stride, q.weight_gain, q.bias_gain,
             q.act, q.alpha, q.act_gain, q.out_scale);
}

// ---- im2col, 3x3: cols[n][ci*9 + t][p] = x[n, ci, oy*stride + t/3 - pad, ox*stride + t%3 - pad] ------------------------
// x is addressed through element strides (NCHW-contiguous or channels-last alike); one thread per (n, ci, output pixel) writes
// the nine taps, so the nine stores of a wave are each a coalesced run along p.  pad 1 / stride 1 is F.unfold(x, 3, padding=1);
// pad 0 / stride 2 is the patch matrix of the valid stride-2 conv that follows the low-pass in the down-2 layers.
__global__ void __launch_bounds__(256) im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ cols, int N, int C, int H, int W,
                                                        int OH, int OW, int pad, int stride, int64_t sn, int64_t sc, int64_t sy, int64_t sx)
{
    const int HW = OH * OW;
    const int64_t total = (int64_t)N * C * HW;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % HW);
    const int64_t nc = e / HW;
    const int c = (int)(nc % C), n = (int)(nc / C);
    const int py = p / OW, px = p - py * OW;
    const float* xb = x + n * sn + c * sc;
    float* cb = cols + ((int64_t)n * C + c) * 9 * HW + p;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = py * stride + t / 3 - pad, ix = px * stride + t % 3 - pad;
        const bool ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
        cb[(int64_t)t * HW] = ok ? xb[iy * sy + ix * sx] : 0.f;
    }
}

// ---- y[n, c, p] = clamp(act(y + noise[p] * strength + bias[c]) * gain) -------------------------------------------------
__global__ void __launch_bounds__(256) noise_bias_act_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ noise,
                                                             const float* __restrict__ noise_strength, const float* __restrict__ bias,
                                                             int64_t total, int C, int HW, int act, float alpha, float gain, float clamp)
{
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= total) return;
    const float ns = noise ? noise_strength[0] : 0.f;
    const int p = (int)(e % HW);                                     // HW % 4 == 0: the four elements share n and c
    const int c = (int)((e / HW) % C);
    const float b = bias ? bias[c] : 0.f;
    float4 v = *(const float4*)(x + e);
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = r[k] + b;
        if (noise) t = fmaf(noise[p + k], ns, r[k]) + b;
        if (act == 3) t = t > 0.f ? t : t * alpha;
        t *= gain;
        if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
        r[k] = t;
    }
    *(float4*)(y + e) = make_float4(r[0], r[1], r[2], r[3]);
}

// ---- pixel-centre rays (training/volumetric_rendering/ray_sampler.py:24-62): one thread per ray --------------------------
// Pixel (row i, col j) -> image coords ((j+.5)/R, (i+.5)/R) -> camera-frame point at z = 1 (OpenCV intrinsics with skew)
// -> world space through cam2world -> unit direction from the camera origin.  Same operation order as the tensor-op form.
__global__ void __launch_bounds__(256) ray_sample_kernel(const float* __restrict__ c2w, const float* __restrict__ intr, float* __restrict__ origins,
                                                         float* __restrict__ dirs, int N, int R, int64_t c2w_stride, int64_t intr_stride)
{
    const int M = R * R;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * M) return;
    const int n = e / M, m = e - n * M;
    const int i = m / R, j = m - i * R;
    const float* K = intr + n * intr_stride;
    const float* C = c2w + n * c2w_stride;
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const float inv = 1.f / (float)R, half = 0.5f / (float)R;
    const float x_img = (float)j * inv + half, y_img = (float)i * inv + half;
    const float x_cam = (x_img - cx + cy * sk / fy - sk * y_img / fy) / fx;
    const float y_cam = (y_img - cy) / fy;
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wk = C[k * 4 + 0] * x_cam + C[k * 4 + 1] * y_cam + C[k * 4 + 2] + C[k * 4 + 3];
        d[k] = wk - C[k * 4 + 3];
    }
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[(int64_t)e * 3 + k] = C[k * 4 + 3];
        dirs[(int64_t)e * 3 + k] = d[k] / nrm;
    }
}

} // namespace p3d

extern "C" int p3d_ray_sample(const float* cam2world, const float* intrinsics, float* origins, float* dirs, int32_t n_cam, int32_t resolution,
                              p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(cam2world && intrinsics && origins && dirs, "ray_sample: null pointer");
    P3D_REQUIRE(n_cam >= 1 && resolution >= 1 && (int64_t)n_cam * resolution * resolution < (1ll << 31), "ray_sample: bad sizes");
    const int total = n_cam * resolution * resolution;
    hipLaunchKernelGGL(ray_sample_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, cam2world, intrinsics, origins, dirs,
                       n_cam, resolution, (int64_t)16, (int64_t)9);
    count_launch(FAM_AUX);
    return check_launch("ray_sample");
}

extern "C" int p3d_ray_sample_labels(const float* labels, int64_t label_stride, float* origins, float* dirs, int32_t n_cam, int32_t resolution, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(labels && origins && dirs, "ray_sample_labels: null pointer");
    P3D_REQUIRE(n_cam >= 1 && resolution >= 1 && label_stride >= 25 && (int64_t)n_cam * resolution * resolution < (1ll << 31), "ray_sample_labels: bad sizes");
    const int total = n_cam * resolution * resolution;
    hipLaunchKernelGGL(ray_sample_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, labels, labels + 16, origins, dirs,
                       n_cam, resolution, label_stride, label_stride);
    count_launch(FAM_AUX);
    return check_launch("ray_sample_labels");
}

extern "C" int p3d_fc_forward(const float* x, const float* w, const float* b, float* y, int32_t n_rows, int32_t in_features,
                              int32_t out_features, int64_t x_row_stride, float weight_gain, float bias_gain, int32_t act, float alpha, float act_gain,
                              float out_scale, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && w && y, "fc_forward: null pointer");
    P3D_REQUIRE(n_rows >= 1 && n_rows <= FC_MAXN, "fc_forward: n_rows=%d outside [1, %d]", n_rows, FC_MAXN);
    P3D_REQUIRE(in_features >= 4 && in_features % 4 == 0 && out_features >= 1, "fc_forward: in_features must be a positive multiple of 4");
    P3D_REQUIRE(act == 1 || act == 3, "fc_forward: act must be linear (1) or lrelu (3)");
    P3D_REQUIRE(((((uintptr_t)x) | ((uintptr_t)w)) & 15u) == 0 && x_row_stride % 4 == 0, "fc_forward: x, its rows and w must be 16-byte aligned");
    const size_t shm = (size_t)n_rows * in_features * sizeof(float);
    P3D_REQUIRE(shm <= 64 * 1024, "fc_forward: n_rows * in_features too large for the LDS stage");
    hipLaunchKernelGGL(fc_kernel, dim3((out_features + 3) / 4), dim3(256), shm, (hipStream_t)stream, x, w, b, y, n_rows, in_features,
                       out_features, x_row_stride, weight_gain, bias_gain, act, alpha, act_gain, out_scale);
    count_launch(FAM_AUX);
    return check_launch("fc_forward");
}

extern "C" int p3d_demod_coefs(const float* styles, const float* w2, float* d, int32_t n_rows, int32_t ci, int32_t co, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(styles && w2 && d, "demod_coefs: null pointer");
    P3D_REQUIRE(n_rows >= 1 && n_rows <= FC_MAXN && ci >= 1 && co >= 1, "demod_coefs: 1 .. %d rows", FC_MAXN);
    const size_t shm = (size_t)n_rows * ci * sizeof(float);
    P3D_REQUIRE(shm <= 64 * 1024, "demod_coefs: n_rows * ci too large for the LDS stage");
    hipLaunchKernelGGL(demod_coefs_kernel, dim3((co + 3) / 4), dim3(256), shm, (hipStream_t)stream, styles, w2, d, n_rows, ci, co);
    count_launch(FAM_AUX);
    return check_launch("demod_coefs");
}

extern "C" int p3d_demod_coefs_multi(const p3d_demod_job* jobs_host, int32_t n_jobs, int32_t n_rows, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= P3D_DEMOD_MAX_JOBS, "demod_coefs_multi: 1 .. %d jobs", P3D_DEMOD_MAX_JOBS);
    P3D_REQUIRE(n_rows >= 1 && n_rows <= FC_MAXN, "demod_coefs_multi: 1 .. %d rows", FC_MAXN);
    DemodJobs a{};
    a.njobs = n_jobs; a.n_rows = n_rows;
    int blocks = 0, max_ci = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const p3d_demod_job& q = jobs_host[j];
        P3D_REQUIRE(q.styles && q.w2 && q.d && q.ci >= 1 && q.co >= 1, "demod_coefs_multi: job %d has a null pointer or an empty size", j);
        a.job[j] = q; a.first_block[j] = blocks;
        blocks += (q.co + 3) / 4;
        if (q.ci > max_ci) max_ci = q.ci;
    }
    a.first_block[n_jobs] = blocks;
    const size_t shm = (size_t)n_rows * max_ci * sizeof(float);
    P3D_REQUIRE(shm <= 64 * 1024, "demod_coefs_multi: n_rows * ci too large for the LDS stage");
    hipLaunchKernelGGL(demod_coefs_multi_kernel, dim3(blocks), dim3(256), shm, (hipStream_t)stream, a);
    count_launch(FAM_AUX);
    return check_launch("demod_coefs_multi");
}

extern "C" int p3d_demod_coefs_backward(const float* gd, const float* d, const float* styles, const float* w2, const float* weight, float* gs, float* gw,
                                        int32_t n_rows, int32_t ci, int32_t co, int32_t taps, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(gd && d && styles && w2 && weight, "demod_coefs_backward: null pointer");
    P3D_REQUIRE(n_rows >= 1 && n_rows <= FC_MAXN && ci >= 1 && co >= 1 && taps >= 1, "demod_coefs_backward: 1 .. %d rows", FC_MAXN);
    hipStream_t s = (hipStream_t)stream;
    if (gs) {
        const size_t a = (size_t)n_rows * co, b = (size_t)256 * n_rows;
        const size_t shm = (a > b ? a : b) * sizeof(float);
        P3D_REQUIRE(shm <= 64 * 1024, "demod_coefs_backward: n_rows * co too large for the LDS stage");
        hipLaunchKernelGGL(demod_bwd_styles_kernel, dim3((ci + 63) / 64), dim3(256), shm, s, gd, d, styles, w2, gs, n_rows, ci, co);
        count_launch(FAM_AUX);
        int rc = check_launch("demod_coefs_backward (styles)");
        if (rc != P3D_OK) return rc;
    }
    if (gw) {
        const int64_t total = (int64_t)co * ci * taps;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(demod_bwd_weight_kernel, dim3(blocks), dim3(256), 0, s, gd, d, styles, weight, gw, n_rows, ci, co, taps);
        count_launch(FAM_AUX);
        return check_launch("demod_coefs_backward (weight)");
    }
    return P3D_OK;
}

extern "C" int p3d_fc_multi(const p3d_fc_job* jobs_host, int32_t n_jobs, int32_t n_rows, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= FC_MAX_JOBS, "fc_multi: 1 .. %d jobs", FC_MAX_JOBS);
    P3D_REQUIRE(n_rows >= 1 && n_rows <= FC_MAXN, "fc_multi: n_rows=%d outside [1, %d]", n_rows, FC_MAXN);
    FcJobs a{};
    a.njobs = n_jobs; a.n_rows = n_rows;
    int blocks = 0, max_in = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const p3d_fc_job& q = jobs_host[j];
        P3D_REQUIRE(q.x && q.w && q.y, "fc_multi: null pointer in job %d", j);
        P3D_REQUIRE(q.in_features >= 4 && q.in_features % 4 == 0 && q.out_features >= 1, "fc_multi: in_features must be a positive multiple of 4");
        P3D_REQUIRE(q.act == 1 || q.act == 3, "fc_multi: act must be linear (1) or lrelu (3)");
        P3D_REQUIRE(((((uintptr_t)q.x) | ((uintptr_t)q.w)) & 15u) == 0 && q.x_row_stride % 4 == 0, "fc_multi: x, its rows and w must be 16-byte aligned");
        a.job[j] = q;
        a.first_block[j] = blocks;
        blocks += (q.out_features + 3) / 4;
        if (q.in_features > max_in) max_in = q.in_features;
    }
    a.first_block[n_jobs] = blocks;
    const size_t shm = (size_t)n_rows * max_in * sizeof(float);
    P3D_REQUIRE(shm <= 64 * 1024, "fc_multi: n_rows * in_features too large for the LDS stage");
    hipLaunchKernelGGL(fc_multi_kernel, dim3(blocks), dim3(256), shm, (hipStream_t)stream, a);
    count_launch(FAM_AUX);
    return check_launch("fc_multi");
}

extern "C" int p3d_im2col3x3(const float* x, float* cols, int32_t n_img, int32_t c, int32_t h, int32_t w, int32_t pad, int32_t stride,
                             int64_t stride_n, int64_t stride_c, int64_t stride_y, int64_t stride_x, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && cols, "im2col3x3: null pointer");
    P3D_REQUIRE(n_img >= 1 && c >= 1 && h >= 1 && w >= 1, "im2col3x3: bad sizes");
    P3D_REQUIRE((pad == 0 || pad == 1) && (stride == 1 || stride == 2), "im2col3x3: pad must be 0 or 1, stride 1 or 2");
    const int oh = (h + 2 * pad - 3) / stride + 1, ow = (w + 2 * pad - 3) / stride + 1;
    P3D_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, "im2col3x3: image smaller than the window");
    const int64_t total = (int64_t)n_img * c * oh * ow;
    P3D_REQUIRE(total < (1ll << 31) * 256, "im2col3x3: too large");
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, cols, n_img, c, h, w,
                       oh, ow, pad, stride, stride_n, stride_c, stride_y, stride_x);
    count_launch(FAM_AUX);
    return check_launch("im2col3x3");
}

extern "C" int p3d_noise_bias_act(const float* x, float* y, const float* noise, const float* noise_strength, const float* bias, int32_t n_img,
                                  int32_t c, int32_t hw, int32_t act, float alpha, float gain, float clamp, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && y, "noise_bias_act: null pointer");
    P3D_REQUIRE(n_img >= 1 && c >= 1 && hw >= 4 && hw % 4 == 0, "noise_bias_act: hw must be a positive multiple of 4");
    P3D_REQUIRE(act == 1 || act == 3, "noise_bias_act: act must be linear (1) or lrelu (3)");
    P3D_REQUIRE(!noise || noise_strength, "noise_bias_act: noise needs its strength");
    P3D_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0, "noise_bias_act: x and y must be 16-byte aligned");
    const int64_t total = (int64_t)n_img * c * hw;
    hipLaunchKernelGGL(noise_bias_act_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, noise,
                       noise_strength, bias, total, c, hw, act, alpha, gain, clamp);
    count_launch(FAM_AUX);
    return check_launch("noise_bias_act");
}
