// bias_act for gfx950: y = clamp(act(x + b) * gain), plus its first/second-order gradient forms.
//
// Behavioural contract: torch_utils/ops/bias_act.cu:27-151 of the reference (activation table
// bias_act.py:23-33).  Written from that contract, not from its code: this version streams
// 16 bytes per lane (float4 / 8 halves / 2 doubles), derives the bias index once per vector with
// 32-bit arithmetic, and grid-strides a launch capped at a few workgroups per CU — the op is pure
// HBM streaming (2*sizeof(T) B/elem forward) so the only goals are wide coalesced accesses and
// enough loads in flight.
#include "p3d_common.h"

namespace p3d {

struct BiasActArgs {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int grad; float alpha, gain, clamp;
    uint32_t size_x, size_b, step_b;
};

template <class S> __device__ __forceinline__ S exp_(S v);
template <> __device__ __forceinline__ float  exp_<float>(float v)   { return __expf(v); }
template <> __device__ __forceinline__ double exp_<double>(double v) { return exp(v); }
template <class S> __device__ __forceinline__ S log_(S v);
template <> __device__ __forceinline__ float  log_<float>(float v)   { return __logf(v); }
template <> __device__ __forceinline__ double log_<double>(double v) { return log(v); }

// One element.  `v` is the streamed operand: the pre-activation input for G=0, the incoming
// gradient for G>=1.  `xr` = saved forward input (+bias), `yr` = saved forward output.
// Returns the value before the common "* gain * dy" and clamp steps; may rewrite yr (swish).
template <class S, int ACT>
__device__ __forceinline__ S act_eval(int G, S v, S xr, S& yr, S alpha, S gain)
{
    const S one = 1, two = 2, lim = 80;
    const S yy = (gain != S(0)) ? yr / gain : S(0);      // forward output with the gain undone
    if (ACT == 1) return (G <= 1) ? v : S(0);
    if (ACT == 2) return (G == 0) ? (v > 0 ? v : S(0)) : (G == 1 ? (yy > 0 ? v : S(0)) : S(0));
    if (ACT == 3) return (G == 0) ? (v > 0 ? v : v * alpha) : (G == 1 ? (yy > 0 ? v : v * alpha) : S(0));
    if (ACT == 4) {
        if (G == 0) { S e = exp_(v), r = one / e; return (v < -lim) ? -one : (v > lim) ? one : (e - r) / (e + r); }
        S g1 = v * (one - yy * yy);
        return (G == 1) ? g1 : g1 * (-two * yy);
    }
    if (ACT == 5) {
        if (G == 0) return (v < -lim) ? S(0) : one / (exp_(-v) + one);
        S g1 = v * yy * (one - yy);
        return (G == 1) ? g1 : g1 * (one - two * yy);
    }
    if (ACT == 6) {
        if (G == 0) return (v >= 0) ? v : exp_(v) - one;
        S t = v * (yy + one);
        return (G == 1) ? (yy >= 0 ? v : t) : (yy >= 0 ? S(0) : t);
    }
    if (ACT == 7) {
        const S sc = (S)1.0507009873554804934193349852946, sa = sc * (S)1.6732632423543772848170429916717;
        if (G == 0) return (v >= 0) ? sc * v : sa * (exp_(v) - one);
        S t = v * (yy + sa);
        return (G == 1) ? (yy >= 0 ? v * sc : t) : (yy >= 0 ? S(0) : t);
    }
    if (ACT == 8) {
        if (G == 0) return (v > lim) ? v : log_(exp_(v) + one);
        S e = exp_(-yy);
        return (G == 1) ? v * (one - e) : v * e * (one - e);
    }
    if (ACT == 9) {
        if (G == 0) return (v < -lim) ? S(0) : v / (exp_(-v) + one);
        S e = exp_(xr), d = e + one, r;
        if (G == 1) r = (xr > S(40)) ? v : v * e * (xr + d) / (d * d);
        else        r = (xr > S(40)) ? S(0) : v * e * (xr * (two - d) + two * d) / (d * d * d);
        yr = (xr < -lim) ? S(0) : xr / (exp_(-xr) + one) * gain;   // swish saves x, not y
        return r;
    }
    return S(0);
}

template <class S, int ACT>
__device__ __forceinline__ S bias_act_elem(const BiasActArgs& a, S v, S bias, S xr, S yr, S up)
{
    const int G = a.grad;
    if (G == 0) v += bias; else xr += bias;
    S r = act_eval<S, ACT>(G, v, xr, yr, (S)a.alpha, (S)a.gain);
    r *= (S)a.gain * up;
    const S c = (S)a.clamp;
    if (c >= 0) {
        if (G == 0) r = (r > -c && r < c) ? r : (r >= 0 ? c : -c);
        else        r = (yr > -c && yr < c) ? r : S(0);
    }
    return r;
}

template <class T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };

// VEC elements per lane per iteration (16 bytes when VEC > 1); VEC = 1 is the unaligned path.
template <class T, int ACT, int VEC>
__global__ void __launch_bounds__(256) bias_act_kernel(BiasActArgs a)
{
    typedef typename Acc<T>::type S;
    typedef Pack<T, VEC> P;
    const uint32_t nvec   = a.size_x / VEC;
    const uint32_t stride = gridDim.x * blockDim.x;
    const T* xp = (const T*)a.x;   const T* bp = (const T*)a.b;
    const T* xrp = (const T*)a.xref; const T* yrp = (const T*)a.yref; const T* dyp = (const T*)a.dy;
    T* yp = (T*)a.y;

    for (uint32_t iv = blockIdx.x * blockDim.x + threadIdx.x; iv < nvec; iv += stride) {
        const uint32_t i0 = iv * VEC;
        P px = *(const P*)(xp + i0), pxr, pyr, pdy, out;
        if (xrp) pxr = *(const P*)(xrp + i0);
        if (yrp) pyr = *(const P*)(yrp + i0);
        if (dyp) pdy = *(const P*)(dyp + i0);
        uint32_t q = 0, r = 0;                       // bias row and position inside it
        // channels-last (the bias axis is the fastest one, a whole vector never wraps): the VEC biases are ONE vector load at i0 % C — the general path
        // below re-derives row and position per element (this kernel ran at 0.67 of the VALU issue rate for a memory pass, profiles/round4_d_kernel_pmc_train6.txt)
        const bool cl_bias = bp && VEC > 1 && a.step_b == 1 && (a.size_b % VEC) == 0 && ((((uintptr_t)bp) & (sizeof(P) - 1)) == 0);
        P pb;
        if (cl_bias) pb = *(const P*)(bp + (i0 % a.size_b));
        else if (bp) { q = i0 / a.step_b; r = i0 - q * a.step_b; q %= a.size_b; }
        S bias = (bp && !cl_bias) ? ld(bp + q) : S(0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (cl_bias) bias = ld(&pb.v[k]);
            else if (bp && k > 0 && ++r == a.step_b) { r = 0; if (++q == a.size_b) q = 0; bias = ld(bp + q); }
            S v  = ld(&px.v[k]);
            S xr = xrp ? ld(&pxr.v[k]) : S(0);
            S yr = yrp ? ld(&pyr.v[k]) : S(0);
            S up = dyp ? ld(&pdy.v[k]) : S(1);
            st(&out.v[k], bias_act_elem<S, ACT>(a, v, bias, xr, yr, up));
        }
        *(P*)(yp + i0) = out;
    }
    // ragged tail (< VEC elements), handled by the first lanes of block 0
    if (VEC > 1 && blockIdx.x == 0) {
        uint32_t i = nvec * VEC + threadIdx.x;
        if (i < a.size_x) {
            S bias = bp ? ld(bp + (i / a.step_b) % a.size_b) : S(0);
            S v  = ld(xp + i);
            S xr = xrp ? ld(xrp + i) : S(0);
            S yr = yrp ? ld(yrp + i) : S(0);
            S up = dyp ? ld(dyp + i) : S(1);
            st(yp + i, bias_act_elem<S, ACT>(a, v, bias, xr, yr, up));
        }
    }
}

template <class T, int ACT>
static int launch_bias_act(const BiasActArgs& a, bool aligned, hipStream_t s)
{
    constexpr int VEC = 16 / sizeof(T);
    const int threads = 256;
    const int64_t work = aligned ? (a.size_x + VEC - 1) / VEC : a.size_x;
    int blocks = (int)((work + threads - 1) / threads);
    const int cap = kNumCU * 16;                         // grid-stride beyond ~16 blocks per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (aligned) hipLaunchKernelGGL((bias_act_kernel<T, ACT, VEC>), dim3(blocks), dim3(threads), 0, s, a);
    else         hipLaunchKernelGGL((bias_act_kernel<T, ACT, 1>),   dim3(blocks), dim3(threads), 0, s, a);
    count_launch(FAM_BIAS_ACT);
    return check_launch("bias_act");
}

template <class T>
static int dispatch_act(int act, const BiasActArgs& a, bool aligned, hipStream_t s)
{
    switch (act) {
        case 1: return launch_bias_act<T, 1>(a, aligned, s);
        case 2: return launch_bias_act<T, 2>(a, aligned, s);
        case 3: return launch_bias_act<T, 3>(a, aligned, s);
        case 4: return launch_bias_act<T, 4>(a, aligned, s);
        case 5: return launch_bias_act<T, 5>(a, aligned, s);
        case 6: return launch_bias_act<T, 6>(a, aligned, s);
        case 7: return launch_bias_act<T, 7>(a, aligned, s);
        case 8: return launch_bias_act<T, 8>(a, aligned, s);
        case 9: return launch_bias_act<T, 9>(a, aligned, s);
    }
    return fail(P3D_ERR_ARGUMENT, "bias_act: unknown activation index %d", act);
}

} // namespace p3d

extern "C" int p3d_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                            void* y, int dtype, int grad, int act, float alpha, float gain, float clamp,
                            int64_t size_x, int32_t size_b, int64_t step_b, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && y, "bias_act: x and y must be non-null");
    P3D_REQUIRE(size_x >= 0 && size_x <= INT32_MAX, "bias_act: size_x=%lld outside [0, INT32_MAX]", (long long)size_x);
    P3D_REQUIRE(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2 (got %d)", grad);
    P3D_REQUIRE(!b || (size_b > 0 && step_b > 0), "bias_act: bias given but size_b=%d step_b=%lld", size_b, (long long)step_b);
    if (size_x == 0) return P3D_OK;
    BiasActArgs a;
    a.x = x; a.b = b; a.xref = xref; a.yref = yref; a.dy = dy; a.y = y;
    a.grad = grad; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    a.size_x = (uint32_t)size_x; a.size_b = b ? (uint32_t)size_b : 1u;
    a.step_b = b ? (uint32_t)(step_b > INT32_MAX ? INT32_MAX : step_b) : 1u;
    auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15u) == 0; };
    const bool aligned = al16(x) && al16(y) && al16(xref) && al16(yref) && al16(dy);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case P3D_F32: return dispatch_act<float>(act, a, aligned, s);
        case P3D_F16: return dispatch_act<__half>(act, a, aligned, s);
        case P3D_F64: return dispatch_act<double>(act, a, aligned, s);
    }
    return fail(P3D_ERR_ARGUMENT, "bias_act: unknown dtype %d", dtype);
}
