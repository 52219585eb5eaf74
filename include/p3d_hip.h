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

enum p3d_dtype { P3D_F32 = 0, P3D_F16 = 1, P3D_F64 = 2 };

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

#ifdef __cplusplus
}
#endif
#endif /* P3D_HIP_H */
