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

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

} // namespace p3d

#define P3D_REQUIRE(cond, ...) do { if (!(cond)) return p3d::fail(P3D_ERR_ARGUMENT, __VA_ARGS__); } while (0)
