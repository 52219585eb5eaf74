// In-register splits of fp32 MFMA operands into bf16 pieces (bf16x3: hi + lo; bf16x6: hi + mid + lo), shared by the convolution kernels
// (conv2d.hip: forward / data gradient; conv2d_grad.hip: weight gradient).  Include inside namespace p3d after the f32x4 typedef.
#pragma once
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(const f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)); }
__device__ __forceinline__ f32x2_t bf16_pair_as_f32(const uint32_t p) { return f32x2_t{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)}; }
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16x8(const f32x4& a0, const f32x4& a1, bf8& hi, bf8& lo)
{
    u32x4_t h, l;                                                     // on pairs (see split3_bf16x8 below): 2.5 - 3 vector instructions per value instead of 4
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t x = p < 2 ? f32x2_t{a0[2 * p], a0[2 * p + 1]} : f32x2_t{a1[2 * p - 4], a1[2 * p - 3]};
        h[p] = cvt_pk_bf16(x);                                        // round to nearest even
        l[p] = cvt_pk_bf16(x - bf16_pair_as_f32(h[p]));
    }
    hi = __builtin_bit_cast(bf8, h); lo = __builtin_bit_cast(bf8, l);
    // every converted register complete before the first MFMA reads any of them: see the hazard note at split8 in render_device.h
    // (v_cvt_pk_bf16_f32 -> MFMA operand at the compiler's two wait states returned stale 16-lane groups in the ray-marcher)
    asm volatile("s_nop 4" : "+v"(hi), "+v"(lo));
}

// "bf16x6" (P3D_F32_BF16X6): three bf16 pieces per fp32 value — hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both subtractions are exact in fp32 and
// after them at most 8 significant bits are left, so hi + mid + lo == x — and the six products of magnitude >= 2^-16: fp32-accurate, no pre-split layout anywhere.
// Written on PAIRS: one v_cvt_pk_bf16_f32 makes two pieces and IS the operand register; the pieces go back to fp32 with one shift / one mask.  (The
// element-wise form compiled to 7.5 vector instructions per value — the compiler converted every value twice, once alone for the subtraction and once
// in a pair for the operand — and the kernel was bound by them: 300 vector instructions per 24 MFMAs.  This form: 5.5, or 4.5 where the subtractions pair
// up as v_pk_add_f32.)  Same roundings, same bits.
// Non-finite and out-of-range operands: for x = +-inf, and for finite |x| > 3.396e38 (bf16(x) rounds to inf), the residual x - hi is inf - inf or -+inf, so every output
// the operand reaches is NaN — where the f32-input MFMA (P3D_F32_BF16X6=0) and the reference give +-inf when the sum is well defined.  Non-finite in, non-finite out
// either way; the KIND differs (the reference loop's nan_to_num maps nan -> 0, +-inf -> +-1e5).  A guard would cost one compare + one select per value in kernels that are
// bound by this very vector work, so it is stated instead (pinned by tests/test_conv_gpu.py::test_bf16x6_non_finite_operands).
__device__ __forceinline__ void split3_bf16x8(const f32x4& a0, const f32x4& a1, bf8& hi, bf8& mid, bf8& lo)
{
    u32x4_t h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t x = p < 2 ? f32x2_t{a0[2 * p], a0[2 * p + 1]} : f32x2_t{a1[2 * p - 4], a1[2 * p - 3]};
        h[p] = cvt_pk_bf16(x);
        const f32x2_t r1 = x - bf16_pair_as_f32(h[p]);
        m[p] = cvt_pk_bf16(r1);
        l[p] = cvt_pk_bf16(r1 - bf16_pair_as_f32(m[p]));
    }
    hi = __builtin_bit_cast(bf8, h); mid = __builtin_bit_cast(bf8, m); lo = __builtin_bit_cast(bf8, l);
    asm volatile("s_nop 4" : "+v"(hi), "+v"(mid), "+v"(lo));         // conversion -> MFMA operand hazard: see split_bf16x8
}
// term t of the six: (A piece, B piece) = (h,h) (h,m) (m,h) (h,l) (l,h) (m,m)
#define P3D_X6_A(t, h, m, l) ((t) == 2 || (t) == 5 ? (m) : ((t) == 4 ? (l) : (h)))
#define P3D_X6_B(t, h, m, l) ((t) == 1 || (t) == 5 ? (m) : ((t) == 3 ? (l) : (h)))

