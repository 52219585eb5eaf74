"""Model level on the GPU: G.synthesis / sample_mixed through the HIP kernels against outputs recorded from the
reference generator (same name-seeded weights, latents, cameras, uniforms).

Tolerances: all-fp32 run (force_fp32): rendered pixels (image_raw / semantic_raw) and SR images <= 1e-3
relative-to-max, depth <= 1e-4; default precision (fp16 super-resolution blocks with fp32 accumulation, as the
reference runs on a GPU): SR images <= 8e-3 (3 x the worst measured), rendered pixels unchanged (the renderer is always fp32)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, record_error
from model_cases import build_generator, uniforms, replay_uniforms, compare_outputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['seg2cat', 'edge2car'])
@pytest.mark.parametrize('force_fp32', [True, False])
def test_synthesis_on_gpu_matches_reference(hip_lib, name, force_fp32):
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    g = load_golden('model_' + name)
    G = build_generator(name, 'cuda')
    ws, c, nrr = torch.tensor(g['ws'], device='cuda'), torch.tensor(g['c'], device='cuda'), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)
    before = {k: _lib.launch_count(k) for k in ('conv', 'upfirdn2d', 'render')}      # (bias + activation ride in the conv epilogues)
    prev_pol, rmod.fused_policy = rmod.fused_policy, 'require'
    prev_en, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        with replay_uniforms(u_c, u_f), torch.no_grad():
            out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const', force_fp32=force_fp32)
        torch.cuda.synchronize()
    finally:
        rmod.fused_policy, conv2d_gradfix.enabled = prev_pol, prev_en
    for k, v in before.items():
        assert _lib.launch_count(k) > v, f'{k}: HIP kernel did not run'
    assert out['image'].dtype == torch.float32
    # edge2car + fp16 heads: the no-upsampling SR block adds its fp16 ToRGB output into 'image_raw' IN PLACE (reference
    # quirk, superresolution.py:281), so the "raw" images inherit fp16 rounding there; everywhere else they are pure fp32
    # bounds = 3 x the worst value measured on an MI355X (profiles/round5_a_parity_errors.json: fp32 <= 2.9e-5; fp16 heads: images <= 2.5e-3, edge2car's raw <= 1.5e-3)
    tol_raw = 5e-3 if (name == 'edge2car' and not force_fp32) else 1e-4
    errs = compare_outputs(out, g, tol_raw=tol_raw, tol_sr=1e-4 if force_fp32 else 8e-3)
    print(name, 'fp32' if force_fp32 else 'fp16-sr', errs)
    record_error(f'model.{name}.' + ('fp32' if force_fp32 else 'fp16-sr'), errs)


@pytest.mark.parametrize('name', ['seg2cat', 'edge2car'])
def test_sample_mixed_on_gpu(hip_lib, name):
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    g = load_golden('model_' + name)
    G = build_generator(name, 'cuda')
    prev_pol, rmod.fused_policy = rmod.fused_policy, 'require'
    try:
        with torch.no_grad():
            sm = G.sample_mixed(torch.tensor(g['pts'], device='cuda'), None, torch.tensor(g['ws'], device='cuda'), noise_mode='const')
    finally:
        rmod.fused_policy = prev_pol
    assert rel_err(sm['rgb'].cpu().numpy(), g['pts_rgb']) < 1e-3 and rel_err(sm['sigma'].cpu().numpy(), g['pts_sigma']) < 1e-3


def test_const_batch_cache_follows_data_writes(hip_lib):
    """b4's cached batch of learned constants (networks_stylegan2.SynthesisBlock._entry_features): a write through ``.data`` bumps neither the
    version nor the pointer, so ``pix2pix3d_amd.invalidate_weight_caches()`` — what dp.broadcast_module / misc.copy_params_and_buffers / the checkpoint
    loader call — has to drop it; every (batch, dtype, layout) keeps its own tensor (a captured hipGraph goes on reading the one it was captured with)."""
    import pix2pix3d_amd
    from pix2pix3d_amd.training import networks_stylegan2 as ns
    G = build_generator('edge2car', 'cuda')
    b4 = G.backbone.synthesis.b4 if hasattr(G.backbone, 'synthesis') else G.backbone.b4
    fmt = torch.contiguous_format
    with torch.no_grad():
        a2 = b4._entry_features(None, 2, torch.float32, fmt)
        a3 = b4._entry_features(None, 3, torch.float32, fmt)
        assert b4._entry_features(None, 2, torch.float32, fmt) is a2, 'a second batch size evicted the first one'
        assert torch.equal(a2[1], b4.const) and a3.shape[0] == 3
        b4.const.data.copy_(b4.const.data * 2 + 1)                     # what dist.broadcast(param.data) does: no version bump, same pointer
        pix2pix3d_amd.invalidate_weight_caches()
        fresh = b4._entry_features(None, 2, torch.float32, fmt)
        assert fresh is not a2 and torch.equal(fresh[0], b4.const)
        b4.const.mul_(0.5)                                             # an in-place update is seen through the version counter
        assert torch.equal(b4._entry_features(None, 2, torch.float32, fmt)[1], b4.const)
    assert len(ns._const_batches) >= 1


def test_prefetch_plan_waits_are_few_and_change_nothing(hip_lib):
    """The captured step's graph edges: with the position rule of modconv.take_plan a synthesis pass makes a handful of cross-stream waits (the backbone's
    first layer, its ToRGB group, one "everything issued" wait; the heads' plans, issued ahead, none) where one wait per layer is ~30 — and the images are bit-identical
    either way (the rule only drops waits whose event the stream already stands behind)."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    g = load_golden('model_seg2cat')
    G = build_generator('seg2cat', 'cuda')
    ws, c, nrr = torch.tensor(g['ws'], device='cuda'), torch.tensor(g['c'], device='cuda'), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)

    def run():
        counts = {'wait_event': 0, 'wait_stream': 0}
        we, wst = torch.cuda.Stream.wait_event, torch.cuda.Stream.wait_stream

        def count_we(self, ev):
            counts['wait_event'] += 1
            return we(self, ev)

        def count_ws(self, st):
            counts['wait_stream'] += 1
            return wst(self, st)
        torch.cuda.Stream.wait_event, torch.cuda.Stream.wait_stream = count_we, count_ws
        try:
            with replay_uniforms(u_c, u_f), torch.no_grad():
                out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
            torch.cuda.synchronize()
        finally:
            torch.cuda.Stream.wait_event, torch.cuda.Stream.wait_stream = we, wst
        return out, counts
    saved = (modconv.plan_wait_elision, modconv.sr_prefetch_ahead, modconv.premodulate_rgb)
    try:
        run()                                                                    # (weight caches warm: both timed passes launch the same work)
        out1, n1 = run()
        modconv.plan_wait_elision, modconv.sr_prefetch_ahead, modconv.premodulate_rgb = False, False, False
        out0, n0 = run()
    finally:
        modconv.plan_wait_elision, modconv.sr_prefetch_ahead, modconv.premodulate_rgb = saved
    print('waits with the position rule', n1, 'one per layer', n0)
    assert n1['wait_event'] <= 6 and n0['wait_event'] >= 20, (n1, n0)
    assert n1['wait_event'] + n1['wait_stream'] < n0['wait_event'] + n0['wait_stream']
    for k in ('image', 'semantic', 'image_raw', 'semantic_raw', 'image_depth'):
        assert torch.equal(out1[k], out0[k]), k
