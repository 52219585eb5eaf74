// Shared host/device helpers for libp3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/p3d_hip.h"

namespace p3d {

constexpr int kWave = 64;                       // gfx950 wavefront
constexpr int kNumCU = 256;                     // MI355X
enum Family { FAM_BIAS_ACT = 0, FAM_UPFIRDN = 1, FAM_FLRELU = 2, FAM_RENDER = 3, FAM_CONV = 4, FAM_AUX = 5, FAM_COUNT = 6 };

void  set_error(const char* fmt, ...);
int   fail(int code, const char* fmt, ...);
void  count_launch(int family);
int   check_launch(const char* what);         // hipGetLastError -> status

template <class T> struct Acc            { typedef float  type; };
template <>        struct Acc<double>    { typedef double type; };

template <class T> __device__ __forceinline__ typename Acc<T>::type ld(const T* p)            { return (typename Acc<T>::type)(*p); }
template <>        __device__ __forceinline__ float ld<__half>(const __half* p)                 { return __half2float(*p); }
template <class T> __device__ __forceinline__ void st(T* p, typename Acc<T>::type v)          { *p = (T)v; }
template <>        __device__ __forceinline__ void st<__half>(__half* p, float v)               { *p = __float2half(v); }

// One-time opt-in to more than 64 KB of dynamic LDS for kernel `fn`, once PER DEVICE (the attribute belongs to the device's
// code object: a process that drives several GPUs must set it on each); `done` holds one bit per device ordinal.
inline hipError_t reserve_lds_once(const void* fn, int bytes, std::atomic<uint64_t>& done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// csrc/conv2d.hip: p3d_conv2d_nhwc with an explicit output size for the transposed form (conv2d_grad.hip: output_padding)
int conv2d_nhwc_run(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                    const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                    int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t out_h, int32_t out_w,
                    void* workspace, int64_t workspace_bytes, int64_t* query, const float* out_scale, p3d_stream_t stream);
int conv2d_nhwc_run_io(const void* x, const void* w, void* y, int dtype, const float* bias, const float* noise, const float* noise_strength,
                       const void* zeros128, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co, int64_t w_img_stride,
                       int32_t kernel_size, int32_t resample, int32_t act, float gain, float clamp, int32_t out_h, int32_t out_w,
                       void* workspace, int64_t workspace_bytes, int64_t* query, const float* out_scale, int32_t x_split, int32_t y_split,
                       p3d_stream_t stream);

// 4 x 4 transpose of one dword per (lane of a quad, register): on return register c of quad lane t holds what register t of quad lane c held.  Two DPP stages:
// lane ^ 1 inside the register pairs (0, 1), (2, 3), then lane ^ 2 inside (0, 2), (1, 3).  What turns the MFMA accumulator layout (a lane = ONE output channel,
// four consecutive GEMM rows in registers 4 q .. 4 q + 3) into "a lane = four consecutive channels of one row": 16-byte stores instead of four 4-byte (or 2-byte) ones.
__device__ __forceinline__ void quad_transpose4(unsigned (&w)[4], const bool odd1, const bool odd2)
{
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const unsigned send = odd1 ? w[2 * pr] : w[2 * pr + 1];
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
        w[2 * pr]     = odd1 ? recv : w[2 * pr];
        w[2 * pr + 1] = odd1 ? w[2 * pr + 1] : recv;
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const unsigned send = odd2 ? w[pr] : w[pr + 2];
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
        w[pr]     = odd2 ? recv : w[pr];
        w[pr + 2] = odd2 ? w[pr + 2] : recv;
    }
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

} // namespace p3d

#define P3D_REQUIRE(cond, ...) do { if (!(cond)) return p3d::fail(P3D_ERR_ARGUMENT, __VA_ARGS__); } while (0)
