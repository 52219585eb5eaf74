"""The `v_cvt_pk_bf16_f32 -> MFMA SrcB` hazard of gfx950: the workaround of the bf16x3 kernels — hold the wave five
wait states (`s_nop 4`) between the conversions and the first MFMA that reads them — pinned by a hand-written-asm probe instead of remembered.
The probe runs the pair at every distance 0..8 against the same MFMA issued 16 wait states later; results go to gpurun_out/hazard_probe.json."""
import json
import os

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_cvt_to_mfma_distance_of_the_workaround_is_clean(hip_lib):
    from pix2pix3d_amd import diagnostics
    res = {w: diagnostics.cvt_mfma_hazard(w, iters=4000) for w in range(9)}
    res_a = {w: diagnostics.cvt_mfma_hazard(w, iters=4000, src_a=True) for w in range(9)}
    res_w = {w: diagnostics.cvt_mfma_hazard(w, iters=4000, war=True) for w in range(9)}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'hazard_probe.json'), 'w') as f:
        json.dump({'what': 'v_cvt_pk_bf16_f32 -> v_mfma_f32_32x32x16_bf16 (SrcB) at 0..8 wait states vs 16; 512 blocks x 8 waves x 4000 iterations',
                   'stale_lanes_by_wait_states': {str(w): r[0] for w, r in res.items()},
                   'stale_registers_by_wait_states': {str(w): r[1] for w, r in res.items()},
                   'srcA_stale_lanes_by_wait_states': {str(w): r[0] for w, r in res_a.items()},
                   'write_after_read_lanes_by_wait_states': {str(w): r[0] for w, r in res_w.items()},
                   'lanes_total': 512 * 512}, f, indent=1)
    print('stale lanes by wait states: SrcB', {w: r[0] for w, r in res.items()}, 'SrcA', {w: r[0] for w, r in res_a.items()}, 'WAR', {w: r[0] for w, r in res_w.items()})
    for w in (5, 6, 7, 8):                     # the distance the kernels guarantee (s_nop 4 = 5 wait states) and everything beyond it
        assert res[w] == (0, 0) and res_a[w] == (0, 0), (w, res, res_a)
    # the compiler's own distance (2 wait states) is reported, not asserted: whether the stale read shows in isolation depends on what the
    # SIMD's other wave is issuing — the kernels never run at that distance any more


def test_the_kernels_still_carry_the_guard():
    """Source-level pin: every in-register split helper (bf16x3's two pieces, bf16x6's three) ends in the `s_nop 4` that ties every converted register to one point."""
    csrc = os.path.join(ROOT, 'pix2pix3d_amd', 'csrc')
    for fn, helper in (('render_device.h', 'split8'), ('bf16_split.h', 'split_bf16x8'), ('bf16_split.h', 'split3_bf16x8'), ('up2_fir.hip', 'ub_split')):
        src = open(os.path.join(csrc, fn)).read()
        body = src[src.index(helper + '('):]
        body = body[:body.index('\n}\n')]
        assert 'asm volatile("s_nop 4"' in body, (fn, helper)
