"""The two-plane-set renderer (ImportanceSemanticRenderer, renderer.py:256-438) on the device: the DUAL variant of the fused kernel
(p3d_render_forward_dual / p3d_sample_points_dual) against the reference's records (tests/golden/semrenderer_*.npz) and against the
numpy oracle at a size that exercises whole waves, the raster schedule and the plane-set strides.

Tolerances as for the one-plane-set kernel: features / wsum <= 1e-3 relative-to-max (measured ~1e-5), depth <= 1e-4."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import render_oracle as RO

pytestmark = pytest.mark.gpu


def _decoders(g, device, lr, sem_sigmoid):
    from pix2pix3d_amd.training.triplane import OSGDecoder
    from pix2pix3d_amd.training.triplane_cond import OSGDecoder_semantic
    dec_t = OSGDecoder(64, {'decoder_lr_mul': lr, 'decoder_output_dim': 32})
    dec_s = OSGDecoder_semantic(32, {'decoder_lr_mul': lr, 'decoder_output_dim': 32, 'sigmoid': sem_sigmoid})
    with torch.no_grad():
        for dec, pre in ((dec_t, 'dect_'), (dec_s, 'decs_')):
            dec.net[0].weight.copy_(torch.tensor(g[pre + 'w1'])); dec.net[0].bias.copy_(torch.tensor(g[pre + 'b1']))
            dec.net[2].weight.copy_(torch.tensor(g[pre + 'w2'])); dec.net[2].bias.copy_(torch.tensor(g[pre + 'b2']))
    return dec_t.to(device).requires_grad_(False), dec_s.to(device).requires_grad_(False)


def _opts(g):
    from render_cases import _parse
    return {k: _parse(v) for k, v in zip(g['opt_keys'].tolist(), g['opt_vals'].tolist())}


def test_dual_kernel_matches_reference_records(hip_lib):
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.training.volumetric_rendering import renderer as R
    g = load_golden('semrenderer_a')
    opts = _opts(g)
    dec_t, dec_s = _decoders(g, 'cuda', float(g['lr_mul']), bool(g['sem_sigmoid']))
    rend = R.ImportanceSemanticRenderer()
    pt, ps = torch.tensor(g['planes_t'], device='cuda'), torch.tensor(g['planes_s'], device='cuda')
    prev, R.fused_policy = R.fused_policy, 'require'
    try:
        n0 = _lib.launch_count('render')
        with torch.no_grad():
            pm = rend.run_model(pt, ps, dec_t, dec_s, torch.tensor(g['pts'], device='cuda'), None, opts)
            with R._replay_draws(torch.tensor(g['u_coarse'], device='cuda'), torch.tensor(g['u_fine'], device='cuda')):
                feat, depth, wsum = rend(pt, ps, dec_t, dec_s, torch.tensor(g['ray_o'], device='cuda'), torch.tensor(g['ray_d'], device='cuda'), opts)
        assert _lib.launch_count('render') == n0 + 2
    finally:
        R.fused_policy = prev
    for key in ('rgb', 'sigma', 'semantic'):
        assert rel_err(pm[key].cpu().numpy(), g['pts_' + key]) < 1e-4, key
    assert rel_err(feat.cpu().numpy(), g['feat']) < 1e-3 and np.abs(depth.cpu().numpy() - g['depth']).max() < 1e-4 and rel_err(wsum.cpu().numpy(), g['wsum']) < 1e-3


@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_dual_kernel_matches_oracle_at_raster_size(hip_lib, layout):
    """2 images x 32^2 rays x 24+24 samples on 64^2 planes: whole waves, the R x R raster assignment, labels squashed or not,
    plane sets given as plain NCHW tensors or as channels-last backbone outputs read in place."""
    from pix2pix3d_amd.training.volumetric_rendering import renderer as R
    from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    g = load_golden('semrenderer_a')
    for sem_sigmoid in (False, True):
        dec_t, dec_s = _decoders(g, 'cuda', float(g['lr_mul']), sem_sigmoid)
        torch.manual_seed(5)
        n, res, s = 2, 32, 24
        if layout == 'channels_last':
            pt = torch.randn(n, 96, 64, 64, device='cuda').contiguous(memory_format=torch.channels_last).view(n, 3, 32, 64, 64)
            ps = torch.randn(n, 96, 64, 64, device='cuda').contiguous(memory_format=torch.channels_last).view(n, 3, 32, 64, 64)
        else:
            pt, ps = torch.randn(n, 3, 32, 64, 64, device='cuda'), torch.randn(n, 3, 32, 64, 64, device='cuda')
        gg = load_golden('renderer_seg')
        c2w = torch.tensor(gg['c2w'][:n], device='cuda')
        K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]], device='cuda').repeat(n, 1, 1)
        o, d = RaySampler()(c2w, K, res)
        opts = dict(_opts(g), depth_resolution=s, depth_resolution_importance=s)
        u_c, u_f = torch.rand(n, res * res, s, 1, device='cuda'), torch.rand(n * res * res, s, device='cuda')
        rend = R.ImportanceSemanticRenderer()
        prev, R.fused_policy = R.fused_policy, 'require'
        try:
            with torch.no_grad(), R._replay_draws(u_c, u_f):
                feat, depth, wsum = rend(pt, ps, dec_t, dec_s, o, d, opts)
        finally:
            R.fused_policy = prev
        dt = {k[5:]: g[k] for k in g.files if k.startswith('dect_')}
        ds = {k[5:]: g[k] for k in g.files if k.startswith('decs_')}
        dt['lr_mul'] = ds['lr_mul'] = float(g['lr_mul'])
        ds['sigmoid'] = sem_sigmoid
        fo, do, wo = RO.render_semantic(pt.cpu().numpy(), ps.cpu().numpy(), dt, ds, o.cpu().numpy(), d.cpu().numpy(), opts, u_c.cpu().numpy()[..., 0], u_f.cpu().numpy())
        assert rel_err(feat.cpu().numpy(), fo) < 2e-4 and np.abs(depth.cpu().numpy()[..., 0] - do).max() < 5e-5 and rel_err(wsum.cpu().numpy()[..., 0], wo) < 2e-4
