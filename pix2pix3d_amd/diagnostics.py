"""Hardware regression probes of libp3d_hip.so (not part of the reference's interface).

``cvt_mfma_hazard``: the gfx950 hazard behind the ``s_nop 4`` of the bf16x3 kernels — an MFMA reading, as SrcB, registers that
``v_cvt_pk_bf16_f32`` wrote a few wait states earlier (csrc/hazard_probe.hip)."""
import ctypes

import torch

from . import _lib

_lib.register('p3d_probe_cvt_mfma_hazard', ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p])


def cvt_mfma_hazard(wait_states, iters=4000, device='cuda', src_a=False, war=False):
    """(lanes with a stale read, differing accumulator registers) for ``wait_states`` (0..8) between conversion and MFMA,
    the converted registers being the MFMA's SrcB (default) or SrcA; ``war``: the opposite order — the MFMA reads the registers (SrcB) and a
    conversion overwrites them ``wait_states`` later; (0, 0) = clean."""
    counts = torch.zeros([2], dtype=torch.int32, device=device)
    _lib.check(_lib.lib().p3d_probe_cvt_mfma_hazard(int(wait_states), 2 if war else int(bool(src_a)), int(iters), _lib.ptr(counts), _lib.stream_of(counts)), 'probe_cvt_mfma_hazard')
    lanes, regs = counts.cpu().tolist()
    return int(lanes), int(regs)
