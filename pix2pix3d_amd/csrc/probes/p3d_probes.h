/* libp3d_probes.so — hardware probes of gfx950 used by the tests and by the kernel studies under profiles/.  NOT part of the product
 * ABI (include/p3d_hip.h) and not loaded by any op of the package: built only as a separate library (P3D_BUILD_PROBES), bound by
 * pix2pix3d_amd/diagnostics.py.  Nothing here has a counterpart in the reference.                                                    */
#ifndef P3D_PROBES_H
#define P3D_PROBES_H
#include <stdint.h>
#include "../../../include/p3d_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* gfx950 hazard behind the `s_nop 4` of the bf16x3 kernels (csrc/render_device.h split8, csrc/conv2d.hip split_bf16x8):
 * v_mfma_f32_32x32x16_bf16 reading, as SrcB (src_a = 0: the ray-marcher's decoder) or SrcA (src_a = 1: the convolutions' activations),
 * VGPRs written by v_cvt_pk_bf16_f32 `wait_states` (0..8) wait states earlier, against the same MFMA 16 wait states later, on every CU,
 * `iters` iterations per wave.  src_a = 2 probes the opposite order (write-after-read): the MFMA reads the registers as SrcB and a
 * v_cvt_pk_bf16_f32 overwrites them `wait_states` later, against the same overwrite 64 wait states later.
 * counts[0] <- lanes whose results differ, counts[1] <- differing accumulator registers (both 0 = no stale read observed).            */
int p3d_probe_cvt_mfma_hazard(int32_t wait_states, int32_t src_a, int32_t iters, uint32_t* counts, p3d_stream_t stream);

/* What the matrix pipe sustains with NOTHING but v_mfma_f32_32x32x16_f16 in the loop, in conv3x3_h2_f16_kernel's register blocking: a wave
 * holds 2 A fragments x 4 B fragments x 2 K sub-steps (loaded once from `operands`: 6 x 64 lanes x 16 bytes per sub-step, per wave slot
 * 0..7, reused by every wave with that slot) and issues groups of 16 MFMAs into `chains` (1, 2, 4, 8) independent 32x32 accumulators,
 * `iters` groups per wave.  256-thread blocks; `waves_per_simd` (1, 2, 4) = blocks per CU, enforced by the launch bounds of the
 * instantiation and by a dynamic LDS reservation; `blocks` of them.  stamps[block * 4 + wave] <- {s_memtime ticks (shader clock),
 * s_memrealtime ticks (100 MHz)} spent in the loop, from which the caller gets cycles per MFMA and the clock the chip held.
 * sink <- one float per thread (keeps the accumulators alive).                                                                        */
int p3d_probe_mfma_rate(const void* operands, float* sink, uint64_t* stamps, int32_t chains, int32_t waves_per_simd, int32_t blocks,
                        int32_t iters, p3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
