"""Renderer oracle (oracle/render_oracle.py) against records of the reference renderer.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import render_oracle as R
from render_cases import CASES, load_case


@pytest.mark.parametrize('name', CASES)
def test_ray_sampler_oracle(name):
    g, _, _ = load_case(name)
    o, d = R.ray_sampler(g['c2w'], g['K'], int(g['res']))
    assert np.abs(o - g['ray_o']).max() < 1e-6 and np.abs(d - g['ray_d']).max() < 2e-6


@pytest.mark.parametrize('name', CASES)
def test_render_oracle_matches_reference(name):
    g, opts, dec = load_case(name)
    auto = opts['ray_start'] == 'auto'
    kw = {}
    if auto:                                  # per-ray limits are host-side torch code in both implementations
        import torch
        from pix2pix3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
        t0, t1 = ImportanceRenderer()._ray_limits(torch.tensor(g['ray_o']), torch.tensor(g['ray_d']), opts)
        kw = dict(t_start=t0.numpy(), t_end=t1.numpy())
    feat, depth, wsum, det = R.render(g['planes'], dec, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'], details=True, **kw)
    n, m = g['ray_o'].shape[:2]
    assert np.abs(det['z_coarse'] - g['z_coarse'].reshape(n * m, -1)).max() < 1e-6
    assert rel_err(det['w_coarse'], g['w_coarse'].reshape(n * m, -1)) < 2e-5
    assert np.abs(det['z_fine'] - g['z_fine'].reshape(n * m, -1)).max() < 2e-5
    assert np.abs(det['z_all'] - g['z_all'].reshape(n * m, -1)).max() < 2e-5
    assert rel_err(feat, g['feat']) < 1e-4
    assert np.abs(depth - g['depth'][..., 0]).max() < 1e-4
    assert rel_err(wsum, g['wsum'][..., 0]) < 1e-4
    rgb, sigma = R.run_model(g['planes'], dec, g['pts'], opts['box_warp'])
    assert rel_err(rgb, g['pts_rgb']) < 1e-5 and rel_err(sigma, g['pts_sigma'][..., 0]) < 1e-5


def test_importance_sampling_oracle_indices_and_values():
    g = load_golden('renderer_importance')
    zf = R.sample_importance(g['z'], g['w'], g['u'])
    # values agree with the reference to fp32 rounding; a flipped bin would show up as an O(bin width) error
    assert np.abs(zf - g["z_fine"]).max() < 3e-5          # bin width is ~2e-2; peaked pdfs amplify cdf rounding by 1/denom
    bins, w = R.importance_bins(g['z'], g['w'])
    _, inds = R.sample_pdf(bins, w, g['u'], return_index=True)
    assert inds.min() >= 1 and inds.max() <= w.shape[1] + 1
    # empty rays (first 8): pdf is uniform over the S-3 bins, so the index is floor(u * (S-3)) + 1 up to cdf rounding
    exp = np.floor(g['u'][:8] * w.shape[1]).astype(np.int64) + 1
    assert (np.abs(inds[:8] - exp) <= 1).all() and (inds[:8] == exp).mean() > 0.98


def test_oracle_index_work_is_torchs_index_work():
    """The integer results the oracle hands the GPU tests are what the reference's torch calls return on the same inputs: its bin index ==
    torch.searchsorted(cdf, u, right=True) (renderer.py:240) and its stable argsort of cat(coarse, fine) == the permutation torch.sort applies
    in unify_samples (renderer.py:157-167)."""
    import torch
    g = load_golden('renderer_importance')
    bins, w = R.importance_bins(g['z'], g['w'])
    zf, inds = R.sample_pdf(bins, w, g['u'], return_index=True)
    wt = torch.tensor(w) + 1e-5                                               # the reference's own tensor ops (renderer.py:230-240)
    pdf = wt / torch.sum(wt, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    ti = torch.searchsorted(cdf, torch.tensor(g['u']).contiguous(), right=True).numpy()
    differ = ti != inds                                                       # cumsum may round its running sum differently than the sequential loop
    assert differ.mean() < 2e-3
    rows, cols = np.nonzero(differ)
    assert np.all(np.abs(ti[rows, cols] - inds[rows, cols]) == 1)
    edge = np.minimum(cdf.numpy()[rows, np.minimum(ti, inds)[rows, cols]], 1.0)
    assert np.all(np.abs(edge - g['u'][rows, cols]) <= 4 * np.spacing(np.float32(1)))      # ... only where u sits on a cdf entry to an ulp
    z_all = np.concatenate([g['z'], zf], 1)
    order = np.argsort(z_all, axis=1, kind='stable')
    _, tidx = torch.sort(torch.tensor(z_all), dim=1, stable=True)
    assert np.array_equal(order, tidx.numpy())
    assert np.array_equal(np.take_along_axis(z_all, order, 1), torch.sort(torch.tensor(z_all), dim=1)[0].numpy())


@pytest.mark.parametrize('white_back', [False, True])
def test_compositing_backward_oracle_is_autograd_of_the_ray_marcher(white_back):
    """oracle.render_oracle.ray_march_backward (the two sweeps the fused backward runs on the device) against autograd through this
    package's MipRayMarcher2 (itself pinned to the reference's records): d colours, d densities, with and without a wsum gradient."""
    import torch
    from oracle import render_oracle as RO
    from pix2pix3d_amd.training.volumetric_rendering.ray_marcher import MipRayMarcher2
    rng = np.random.RandomState(3)
    r, s, c = 7, 13, 5
    colors = torch.tensor(rng.rand(1, r, s, c), dtype=torch.float64, requires_grad=True)
    sigmas = torch.tensor(rng.randn(1, r, s, 1) * 2 + 1, dtype=torch.float64, requires_grad=True)
    depths = torch.tensor(np.sort(rng.rand(1, r, s, 1) * 1.0 + 2.25, axis=2), dtype=torch.float64)
    rgb, depth, weights = MipRayMarcher2()(colors, sigmas, depths, {'clamp_mode': 'softplus', 'white_back': white_back})
    g_rgb = torch.tensor(rng.randn(1, r, c), dtype=torch.float64)
    g_w = torch.tensor(rng.randn(1, r), dtype=torch.float64)
    loss = (rgb * g_rgb).sum() + (weights.sum(2)[..., 0] * g_w).sum()
    gc, gs = torch.autograd.grad(loss, [colors, sigmas])
    dcol, dsig, cw = RO.ray_march_backward(colors.detach()[0].numpy(), sigmas.detach()[0, ..., 0].numpy(), depths[0, ..., 0].numpy(),
                                           g_rgb[0].numpy(), g_w[0].numpy(), white_back=white_back)
    assert rel_err(dcol, gc[0].numpy()) < 1e-10
    assert rel_err(dsig, gs[0, ..., 0].numpy()) < 1e-9
    assert np.allclose(cw[:, 1:-1], ((weights[0, :, :-1, 0] + weights[0, :, 1:, 0]) / 2).detach().numpy())


def _sem_case(name):
    g = load_golden('semrenderer_' + name)
    from render_cases import _parse
    opts = {k: _parse(v) for k, v in zip(g['opt_keys'].tolist(), g['opt_vals'].tolist())}
    lr = float(g['lr_mul'])
    dec_t = {k[5:]: g[k] for k in g.files if k.startswith('dect_')}
    dec_s = {k[5:]: g[k] for k in g.files if k.startswith('decs_')}
    dec_t['lr_mul'], dec_s['lr_mul'], dec_s['sigmoid'] = lr, lr, bool(g['sem_sigmoid'])
    return g, opts, dec_t, dec_s


@pytest.mark.parametrize('name', ['a', 'b'])
def test_semantic_renderer_oracle_matches_reference_records(name):
    """oracle.render_oracle.render_semantic / run_model_semantic against what the reference's ImportanceSemanticRenderer produced."""
    g, opts, dec_t, dec_s = _sem_case(name)
    rgb, sig, sem = R.run_model_semantic(g['planes_t'], g['planes_s'], dec_t, dec_s, g['pts'], opts['box_warp'])
    assert rel_err(rgb, g['pts_rgb']) < 2e-5 and rel_err(sig, g['pts_sigma'][..., 0]) < 2e-5 and rel_err(sem, g['pts_semantic']) < 2e-5
    feat, depth, wsum = R.render_semantic(g['planes_t'], g['planes_s'], dec_t, dec_s, g['ray_o'], g['ray_d'], opts, g['u_coarse'][..., 0], g['u_fine'])
    assert rel_err(feat, g['feat']) < 1e-4 and rel_err(depth, g['depth'][..., 0]) < 1e-4 and rel_err(wsum, g['wsum'][..., 0]) < 1e-4
