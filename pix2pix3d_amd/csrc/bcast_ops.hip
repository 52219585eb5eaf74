// Per-(image, channel) scaling and its gradient reductions, gfx950 — the element-wise half of the UNFUSED modulated convolution that
// training runs (training/networks_stylegan2.py:70-79 of the reference):
//     x * styles[n, c]                      before the shared-weight convolution,
//     fma(y, dcoefs[n, c], noise)           after it (torch_utils/ops/fma.py:17-60),
// and the backward of both: a product with the same per-(n, c) factor, a dot product over the pixels per (n, c), and for the noise
// a sum over the channels per pixel; plus the bias gradient of bias_act (bias_act.py:190-193: dx.sum over everything but the channel).
// As tensor ops each of these is a broadcast multiply that writes a full-size temporary and a reduction that reads it back; here a
// tensor is read once.  HBM-bound: bytes = one read (+ one write) of the activation.
//
// Layout: a dense tensor is seen as [N][A][B] with B contiguous — NCHW: A = C, B = H*W ("ROW": the factor belongs to the row);
// channels-last: A = H*W, B = C ("COL": the factor belongs to the column).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "p3d_common.h"
#include "../../include/p3d_hip.h"

namespace p3d {

template <class T> struct Vec16;
template <> struct Vec16<float>  { static constexpr int N = 4; typedef float  V __attribute__((ext_vector_type(4))); };
template <> struct Vec16<__half> { static constexpr int N = 8; typedef _Float16 V __attribute__((ext_vector_type(8))); };

// y[n][a][b] = x[n][a][b] * s[n][COL ? b : a] (+ z[zn][COL ? a : b]),  fp32 arithmetic, one rounding.  B % VEC == 0.
// COL (channels-last): block = (column vectors) x (row lanes) over a chunk of rows of image blockIdx.z; a thread keeps its VEC factors in
// registers for all its rows.  ROW (NCHW): blockIdx.y = the (n, channel) row, one factor per block.  No per-element index arithmetic.
template <class T>
__global__ void __launch_bounds__(256) bcast_fma_col_kernel(const T* __restrict__ x, const float* __restrict__ s, const T* __restrict__ z, T* __restrict__ y,
                                                            int A, int B, int64_t z_img_stride, int rows_per_chunk)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    const int n = blockIdx.z, bv = B / VEC;
    const int cpb = bv < 32 ? bv : 32, rl = 256 / cpb;
    const int cv = threadIdx.x % cpb, r0 = threadIdx.x / cpb;
    const int col = blockIdx.y * cpb + cv;
    if (col >= bv || r0 >= rl) return;
    float sv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) sv[e] = s[(int64_t)n * B + col * VEC + e];
    const int a_begin = blockIdx.x * rows_per_chunk, a_end = min(A, a_begin + rows_per_chunk);
    const int64_t base = (int64_t)n * A * bv + col;
    const T* zr = z ? z + n * z_img_stride : nullptr;
    for (int a = a_begin + r0; a < a_end; a += rl) {
        const V xv = ((const V*)x)[base + (int64_t)a * bv];
        const float zz = zr ? (float)zr[a] : 0.f;
        V out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = fmaf((float)xv[e], sv[e], zz);
        ((V*)y)[base + (int64_t)a * bv] = out;
    }
}
template <class T>
__global__ void __launch_bounds__(256) bcast_fma_row_kernel(const T* __restrict__ x, const float* __restrict__ s, const T* __restrict__ z, T* __restrict__ y,
                                                            int A, int B, int64_t z_img_stride)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    const int row = blockIdx.y, n = row / A, bv = B / VEC;
    const float ss = s[row];
    const V* xr = (const V*)x + (int64_t)row * bv;
    V* yr = (V*)y + (int64_t)row * bv;
    const V* zr = z ? (const V*)(z + n * z_img_stride) : nullptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < bv; i += gridDim.x * 256) {
        const V xv = xr[i];
        V out;
        if (zr) {
            const V zv = zr[i];
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[e] = fmaf((float)xv[e], ss, (float)zv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[e] = (float)xv[e] * ss;
        }
        yr[i] = out;
    }
}

// COL: part[n][chunk][b] = sum over the chunk's rows a of p[n][a][b] * q[n][a][b]   (q == null: of p alone)
// Block = 256 threads = (B / VEC column groups) x (256 / that) row lanes; rows are strided over the row lanes, then reduced through LDS.
template <class T>
__global__ void __launch_bounds__(256) col_dot_kernel(const T* __restrict__ p, const T* __restrict__ q, float* __restrict__ part, int A, int B, int rows_per_chunk)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    __shared__ float red[256 * 8];
    const int n = blockIdx.z, chunk = blockIdx.x, cg = blockIdx.y;    // cg: group of 256 / RL ... see host: columns handled per block = CPB
    const int bv = B / VEC;
    const int cpb = bv < 32 ? bv : 32;                                // column vectors per block
    const int rl = 256 / cpb;                                         // row lanes
    const int cv = threadIdx.x % cpb, r0 = threadIdx.x / cpb;
    const int col = cg * cpb + cv;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int a_begin = chunk * rows_per_chunk, a_end = min(A, a_begin + rows_per_chunk);
    if (col < bv && r0 < rl) {
        const int64_t base = (int64_t)n * A * bv + col;
        // four rows in flight per tensor: the loop is nothing but dependent-free 16-byte loads, and with one pair per iteration a wave waited a full
        // memory round trip per 32 bytes (0.3 of the HBM peak at ~1 wave per SIMD in a training iteration: profiles/round4_d_kernel_pmc_train6.txt)
        int a = a_begin + r0;
        for (; a + 3 * rl < a_end; a += 4 * rl) {
            V pv[4], qv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pv[u] = ((const V*)p)[base + (int64_t)(a + u * rl) * bv];
            if (q) {
#pragma unroll
                for (int u = 0; u < 4; ++u) qv[u] = ((const V*)q)[base + (int64_t)(a + u * rl) * bv];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf((float)pv[u][e], (float)qv[u][e], acc[e]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += (float)pv[u][e];
            }
        }
        for (; a < a_end; a += rl) {
            const V pv = ((const V*)p)[base + (int64_t)a * bv];
            if (q) {
                const V qv = ((const V*)q)[base + (int64_t)a * bv];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf((float)pv[e], (float)qv[e], acc[e]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += (float)pv[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    if (r0 == 0 && col < bv) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = 0.f;
            for (int r = 0; r < rl; ++r) v += red[(r * cpb + cv) * 8 + e];
            part[((int64_t)n * gridDim.x + chunk) * B + col * VEC + e] = v;
        }
    }
}
// out[n][b] = sum over chunks of part[n][chunk][b]; eight lanes share an output (chunks are dealt round-robin, then a butterfly)
__global__ void __launch_bounds__(256) col_dot_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int chunks, int B, int64_t total)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    float v = 0.f;
    if (i < total) {
        const int64_t n = i / B; const int b = (int)(i - n * B);
        for (int c = l; c < chunks; c += 8) v += part[(n * chunks + c) * B + b];
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (l == 0 && i < total) out[i] = v;
}

// ROW: out[row] = sum over b of p[row][b] * q[row][b]  (q == null: of p alone); one block per row (B = H*W elements)
template <class T>
__global__ void __launch_bounds__(256) row_dot_kernel(const T* __restrict__ p, const T* __restrict__ q, float* __restrict__ out, int B)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int bv = B / VEC;
    float acc = 0.f;
    for (int i = threadIdx.x; i < bv; i += 256) {
        const V pv = ((const V*)p)[row * bv + i];
        if (q) {
            const V qv = ((const V*)q)[row * bv + i];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc = fmaf((float)pv[e], (float)qv[e], acc);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += (float)pv[e];
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = red[0] + red[1] + red[2] + red[3];
}

// COL: out[n][a] = sum over b (channels) of p[n][a][b]: one wave per row group
template <class T>
__global__ void __launch_bounds__(256) row_sum_cl_kernel(const T* __restrict__ p, float* __restrict__ out, int B, int64_t rows)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    const int bv = B / VEC;
    const int lanes = bv >= 64 ? 64 : (bv >= 32 ? 32 : (bv >= 16 ? 16 : (bv >= 8 ? 8 : (bv >= 4 ? 4 : (bv >= 2 ? 2 : 1)))));   // lanes per row (power of two)
    const int rpw = 64 / lanes;                                       // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / lanes, l = lane - sub * lanes;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + sub; row < rows; row += (int64_t)gridDim.x * 4 * rpw) {
        float acc = 0.f;
        for (int i = l; i < bv; i += lanes) {
            const V pv = ((const V*)p)[row * bv + i];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += (float)pv[e];
        }
        for (int sft = lanes >> 1; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
        if (l == 0) out[row] = acc;
    }
}
// ROW: out[n][b] = sum over a (channels) of p[n][a][b]
template <class T>
__global__ void __launch_bounds__(256) col_sum_nchw_kernel(const T* __restrict__ p, float* __restrict__ out, int A, int B, int64_t total_vec)
{
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::V V;
    const int bv = B / VEC;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // n * bv + column vector
    if (i >= total_vec) return;
    const int64_t n = i / bv; const int c = (int)(i - n * bv);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int a = 0; a < A; ++a) {
        const V pv = ((const V*)p)[(n * A + a) * bv + c];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += (float)pv[e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[n * B + c * VEC + e] = acc[e];
}

static int vec_of(int dtype) { return dtype == P3D_F16 ? 8 : 4; }

} // namespace p3d

extern "C" int p3d_bcast_fma(const void* x, const float* scale, const void* z, void* y, int dtype, int32_t channels_last, int32_t n, int32_t a, int32_t b,
                             int32_t z_per_image, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && scale && y, "bcast_fma: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32, "bcast_fma: dtype must be fp16 or fp32");
    P3D_REQUIRE(n >= 1 && a >= 1 && b >= 1, "bcast_fma: bad sizes");
    if (b % vec_of(dtype) != 0) return fail(P3D_ERR_UNSUPPORTED, "bcast_fma: inner extent %d must be a multiple of %d", b, vec_of(dtype));
    P3D_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)z)) & 15u) == 0, "bcast_fma: tensors must be 16-byte aligned");
    const int64_t zs = z_per_image ? (channels_last ? a : b) : 0;
    hipStream_t s = (hipStream_t)stream;
    const int bv = b / vec_of(dtype);
    if (channels_last) {
        const int cpb = bv < 32 ? bv : 32, rl = 256 / cpb;
        int chunks = (int)(((int64_t)kNumCU * 8 + (int64_t)n * ((bv + cpb - 1) / cpb) - 1) / ((int64_t)n * ((bv + cpb - 1) / cpb)));    // ~8 blocks per CU in all
        const int max_chunks = (a + 4 * rl - 1) / (4 * rl);                                  // at least four rows per thread
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
        const int rows_per_chunk = (a + chunks - 1) / chunks;
        dim3 grid((a + rows_per_chunk - 1) / rows_per_chunk, (bv + cpb - 1) / cpb, n);
        if (dtype == P3D_F16) hipLaunchKernelGGL(bcast_fma_col_kernel<__half>, grid, dim3(256), 0, s, (const __half*)x, scale, (const __half*)z, (__half*)y, a, b, zs, rows_per_chunk);
        else                  hipLaunchKernelGGL(bcast_fma_col_kernel<float>,  grid, dim3(256), 0, s, (const float*)x, scale, (const float*)z, (float*)y, a, b, zs, rows_per_chunk);
    } else {
        P3D_REQUIRE((int64_t)n * a < 65536, "bcast_fma: n * channels must be below 65536");
        int bx = (bv + 256 * 4 - 1) / (256 * 4); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
        dim3 grid(bx, n * a);
        if (dtype == P3D_F16) hipLaunchKernelGGL(bcast_fma_row_kernel<__half>, grid, dim3(256), 0, s, (const __half*)x, scale, (const __half*)z, (__half*)y, a, b, zs);
        else                  hipLaunchKernelGGL(bcast_fma_row_kernel<float>,  grid, dim3(256), 0, s, (const float*)x, scale, (const float*)z, (float*)y, a, b, zs);
    }
    count_launch(FAM_AUX);
    return check_launch("bcast_fma");
}

extern "C" int64_t p3d_channel_dot_workspace(int32_t channels_last, int32_t n, int32_t a, int32_t b)
{
    if (!channels_last) return 0;
    int chunks = (a + 255) / 256; if (chunks > 256) chunks = 256; if (chunks < 1) chunks = 1;
    return (int64_t)n * chunks * b * 4;
}

extern "C" int p3d_channel_dot(const void* p, const void* q, float* out, void* workspace, int64_t workspace_bytes, int dtype, int32_t channels_last,
                               int32_t n, int32_t a, int32_t b, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(p && out, "channel_dot: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32, "channel_dot: dtype must be fp16 or fp32");
    P3D_REQUIRE(n >= 1 && a >= 1 && b >= 1, "channel_dot: bad sizes");
    if (b % vec_of(dtype) != 0) return fail(P3D_ERR_UNSUPPORTED, "channel_dot: inner extent %d must be a multiple of %d", b, vec_of(dtype));
    P3D_REQUIRE(((((uintptr_t)p) | ((uintptr_t)q)) & 15u) == 0, "channel_dot: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (!channels_last) {                                             // one block per (n, channel) row
        if (dtype == P3D_F16) hipLaunchKernelGGL(row_dot_kernel<__half>, dim3((unsigned)((int64_t)n * a)), dim3(256), 0, s, (const __half*)p, (const __half*)q, out, b);
        else                  hipLaunchKernelGGL(row_dot_kernel<float>,  dim3((unsigned)((int64_t)n * a)), dim3(256), 0, s, (const float*)p, (const float*)q, out, b);
        count_launch(FAM_AUX);
        return check_launch("channel_dot");
    }
    const int64_t need = p3d_channel_dot_workspace(1, n, a, b);
    P3D_REQUIRE(workspace && workspace_bytes >= need, "channel_dot: workspace too small (p3d_channel_dot_workspace)");
    int chunks = (a + 255) / 256; if (chunks > 256) chunks = 256; if (chunks < 1) chunks = 1;
    const int rows_per_chunk = (a + chunks - 1) / chunks;
    const int bv = b / vec_of(dtype), cpb = bv < 32 ? bv : 32;
    dim3 grid(chunks, (bv + cpb - 1) / cpb, n);
    if (dtype == P3D_F16) hipLaunchKernelGGL(col_dot_kernel<__half>, grid, dim3(256), 0, s, (const __half*)p, (const __half*)q, (float*)workspace, a, b, rows_per_chunk);
    else                  hipLaunchKernelGGL(col_dot_kernel<float>,  grid, dim3(256), 0, s, (const float*)p, (const float*)q, (float*)workspace, a, b, rows_per_chunk);
    const int64_t total = (int64_t)n * b;
    hipLaunchKernelGGL(col_dot_finish_kernel, dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, out, chunks, b, total);
    count_launch(FAM_AUX);
    return check_launch("channel_dot");
}

extern "C" int p3d_pixel_sum(const void* p, float* out, int dtype, int32_t channels_last, int32_t n, int32_t a, int32_t b, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(p && out, "pixel_sum: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32, "pixel_sum: dtype must be fp16 or fp32");
    P3D_REQUIRE(n >= 1 && a >= 1 && b >= 1, "pixel_sum: bad sizes");
    if (b % vec_of(dtype) != 0) return fail(P3D_ERR_UNSUPPORTED, "pixel_sum: inner extent %d must be a multiple of %d", b, vec_of(dtype));
    P3D_REQUIRE((((uintptr_t)p) & 15u) == 0, "pixel_sum: tensor must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (channels_last) {
        const int64_t rows = (int64_t)n * a;
        const unsigned blocks = (unsigned)((rows + 3) / 4 < (int64_t)kNumCU * 16 ? (rows + 3) / 4 : (int64_t)kNumCU * 16);
        if (dtype == P3D_F16) hipLaunchKernelGGL(row_sum_cl_kernel<__half>, dim3(blocks), dim3(256), 0, s, (const __half*)p, out, b, rows);
        else                  hipLaunchKernelGGL(row_sum_cl_kernel<float>,  dim3(blocks), dim3(256), 0, s, (const float*)p, out, b, rows);
    } else {
        const int64_t total = (int64_t)n * (b / vec_of(dtype));
        if (dtype == P3D_F16) hipLaunchKernelGGL(col_sum_nchw_kernel<__half>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const __half*)p, out, a, b, total);
        else                  hipLaunchKernelGGL(col_sum_nchw_kernel<float>,  dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)p, out, a, b, total);
    }
    count_launch(FAM_AUX);
    return check_launch("pixel_sum");
}
