"""Parity of the fused HIP ray-marcher (through the C ABI) with the numpy oracle and with records of the
reference renderer, on the same planes / decoder weights / rays / uniforms.

Tolerances (fp32 path): rendered features and wsum <= 1e-3 relative-to-max (the north-star bar; measured ~1e-5),
depths <= 1e-4 absolute, importance depths: index-exact bins (values <= 2e-6) given identical inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import render_oracle as R
from render_cases import CASES, load_case, make_decoder

pytestmark = pytest.mark.gpu


def _renderer():
    from pix2pix3d_amd.training.volumetric_rendering import renderer
    return renderer


@pytest.mark.parametrize('name', CASES)
def test_fused_render_matches_reference_and_oracle(hip_lib, name):
    from pix2pix3d_amd import _lib
    rmod = _renderer()
    g, opts, dec_arrays = load_case(name)
    dec = make_decoder(g, 'cuda')
    dev = 'cuda'
    planes = torch.tensor(g['planes'], device=dev)
    o, d = torch.tensor(g['ray_o'], device=dev), torch.tensor(g['ray_d'], device=dev)
    t0 = t1 = None
    if opts['ray_start'] == 'auto':
        t0, t1 = rmod.ImportanceRenderer()._ray_limits(o, d, opts)
    n0 = _lib.launch_count('render')
    out = rmod.fused_render(planes, dec, o, d, opts, torch.tensor(g['u_coarse'], device=dev), torch.tensor(g['u_fine'], device=dev), t0, t1, debug=True)
    assert out is not None and _lib.launch_count('render') == n0 + 1
    feat, depth, wsum, z_fine, w_coarse = [t.cpu().numpy() for t in out]
    n, m = g['ray_o'].shape[:2]
    # --- against the reference's own record
    assert rel_err(w_coarse, g['w_coarse'].reshape(n * m, -1)) < 1e-4
    zf_ref = np.sort(g['z_fine'].reshape(n * m, -1), axis=1)
    assert np.abs(z_fine - zf_ref).max() < 5e-5
    assert rel_err(feat, g['feat']) < 1e-3
    assert np.abs(depth - g['depth']).max() < 1e-4
    assert rel_err(wsum, g['wsum']) < 1e-3
    # --- against the oracle (tighter: same sequential cdf convention)
    kw = dict(t_start=t0.cpu().numpy(), t_end=t1.cpu().numpy()) if t0 is not None else {}
    fo, do, wo, det = R.render(g['planes'], dec_arrays, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'], details=True, **kw)
    assert rel_err(feat, fo) < 2e-4 and np.abs(depth[..., 0] - do).max() < 5e-5 and rel_err(wsum[..., 0], wo) < 2e-4
    assert np.abs(z_fine - np.sort(det['z_fine'], axis=1)).max() < 2e-5


@pytest.mark.parametrize('name', CASES)
def test_layer1_as_bf16x6_is_fp32_accurate(hip_lib, name):
    """The exact forward (training; bf16x3 off) with layer 1 of the decoder MLPs as bf16x6 (render_forward_kernel<.., L1X6>, p3d_pack_decoder_l1x6: six bf16 MFMAs per
    product of three-piece splits, weights split once per work-group) against the same launch with every product on the f32-input MFMA: both hold the oracle's bound, they
    differ from each other by rounding only, and they are not the same bits (the other arithmetic did run)."""
    rmod = _renderer()
    from pix2pix3d_amd.torch_utils.ops import modconv
    g, opts, dec_arrays = load_case(name)
    dec = make_decoder(g, 'cuda')
    dev = 'cuda'
    planes = torch.tensor(g['planes'], device=dev)
    o, d = torch.tensor(g['ray_o'], device=dev), torch.tensor(g['ray_d'], device=dev)
    t0 = t1 = None
    if opts['ray_start'] == 'auto':
        t0, t1 = rmod.ImportanceRenderer()._ray_limits(o, d, opts)
    kw = dict(t_start=t0.cpu().numpy(), t_end=t1.cpu().numpy()) if t0 is not None else {}
    fo, do, wo = R.render(g['planes'], dec_arrays, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'], **kw)
    outs = {}
    prev = (rmod.mlp_l1x6, modconv.f32_x6)
    try:
        modconv.f32_x6 = True
        for l1x6 in (True, False):
            rmod.mlp_l1x6 = l1x6
            out = rmod.fused_render(planes, dec, o, d, opts, torch.tensor(g['u_coarse'], device=dev), torch.tensor(g['u_fine'], device=dev), t0, t1, exact_fp32=True)
            feat, depth, wsum = [t.cpu().numpy() for t in out[:3]]
            assert rel_err(feat, fo) < 2e-4 and np.abs(depth[..., 0] - do).max() < 5e-5 and rel_err(wsum[..., 0], wo) < 2e-4, l1x6
            outs[l1x6] = (feat, depth, wsum)
    finally:
        rmod.mlp_l1x6, modconv.f32_x6 = prev
    e = rel_err(outs[True][0], outs[False][0])
    print(name, 'layer 1 as bf16x6 vs f32-input MFMA', e, 'vs oracle', rel_err(outs[True][0], fo), rel_err(outs[False][0], fo))
    assert e < 2e-4 and not np.array_equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize('name', CASES)
def test_module_level_dispatch_uses_the_fused_kernel(hip_lib, name):
    """ImportanceRenderer.forward / run_model on device tensors: fused path, reference RNG order."""
    from pix2pix3d_amd import _lib
    rmod = _renderer()
    g, opts, _ = load_case(name)
    dec = make_decoder(g, 'cuda')
    planes = torch.tensor(g['planes'], device='cuda')
    o, d = torch.tensor(g['ray_o'], device='cuda'), torch.tensor(g['ray_d'], device='cuda')
    rend = rmod.ImportanceRenderer()
    prev, rmod.fused_policy = rmod.fused_policy, 'require'
    try:
        torch.manual_seed(11)
        n0 = _lib.launch_count('render')
        with torch.no_grad():
            feat, depth, wsum = rend(planes, dec, o, d, opts)
        assert _lib.launch_count('render') == n0 + 1
        # same seed through the tensor-op restatement on the same device: identical draws, so near-identical output
        rmod.fused_policy = 'never'
        torch.manual_seed(11)
        with torch.no_grad():
            f2, d2, w2 = rend(planes, dec, o, d, opts)
        assert rel_err(feat.cpu().numpy(), f2.cpu().numpy()) < 1e-3
        assert (depth - d2).abs().max().item() < 2e-4 and rel_err(wsum.cpu().numpy(), w2.cpu().numpy()) < 1e-3
        rmod.fused_policy = 'require'
        with torch.no_grad():
            pm = rend.run_model(planes, dec, torch.tensor(g['pts'], device='cuda'), None, opts)
        assert rel_err(pm['rgb'].cpu().numpy(), g['pts_rgb']) < 1e-4
        assert rel_err(pm['sigma'].cpu().numpy(), g['pts_sigma']) < 1e-4
    finally:
        rmod.fused_policy = prev


def test_importance_sampling_is_index_exact(hip_lib):
    """Index/gather work: identical (z, w, u) in -> the same bins out as the oracle, so values agree to the last
    few ulps of the final interpolation; and the sorted variant is exactly the sorted unsorted one."""
    rmod = _renderer()
    g = load_golden('renderer_importance')
    z, w, u = (torch.tensor(g[k], device='cuda') for k in ('z', 'w', 'u'))
    zf = rmod.importance_sample_native(z, w, u).cpu().numpy()
    zo = R.sample_importance(g['z'], g['w'], g['u'])
    assert np.abs(zf - zo).max() <= 4e-7, np.abs(zf - zo).max()            # a flipped bin would be >= 1e-3
    assert (zf == zo).mean() > 0.9
    zs = rmod.importance_sample_native(z, w, u, sort=True).cpu().numpy()
    assert np.array_equal(zs, np.sort(zf, axis=1))                         # bitonic network == sort, bit for bit
    assert np.abs(zf - g['z_fine']).max() < 3e-5                           # and the reference's record


def _check_index_work(rmod, z, w, u, allow_ties):
    """Integers against integers: the device's searchsorted index per draw == the oracle's (``sample_pdf(return_index=True)``, pinned to
    torch.searchsorted by tests/test_render_oracle.py) and the device's merge pattern == a stable argsort of cat(coarse, the device's own sorted
    fine depths).  ``allow_ties``: at 65 536 rays x 64 draws a handful of draws sit on a cdf entry to the last bit, where the (bit-identical in
    exact arithmetic) running sums of the two implementations may round apart — each such draw must BE such a tie, and is counted."""
    zt, wt, ut = (torch.tensor(v, device='cuda') for v in (z, w, u))
    zf_dev, bins_dev, merged_dev = rmod.importance_sample_index_native(zt, wt, ut)
    zf_dev, bins_dev, merged_dev = zf_dev.cpu().numpy(), bins_dev.cpu().numpy().astype(np.int64), merged_dev.cpu().numpy()
    bins, wp = R.importance_bins(z, w)
    zo, inds = R.sample_pdf(bins, wp, u, return_index=True)
    differ = bins_dev != inds
    n_diff = int(differ.sum())
    if not allow_ties:
        assert n_diff == 0, n_diff
    else:
        rows, cols = np.nonzero(differ)
        assert np.all(np.abs(bins_dev[rows, cols] - inds[rows, cols]) == 1)
        wq = (wp + np.float32(1e-5)).astype(np.float32)
        cdf = np.concatenate([np.zeros([wq.shape[0], 1]), np.cumsum(wq.astype(np.float64), 1) / wq.astype(np.float64).sum(1, keepdims=True)], 1)
        edge = cdf[rows, np.minimum(bins_dev, inds)[rows, cols]]
        assert np.all(np.abs(edge - u[rows, cols]) <= 1e-6), np.abs(edge - u[rows, cols]).max()       # u within ~8 ulp of the cdf entry that decides the bin
        assert n_diff <= 1e-5 * differ.size, (n_diff, differ.size)
    same = ~differ.any(axis=1)
    assert np.array_equal(zf_dev[same], np.sort(rmod.importance_sample_native(zt, wt, ut).cpu().numpy(), axis=1)[same])
    assert np.abs(zf_dev[same] - np.sort(zo, axis=1)[same]).max() <= 4e-7                           # identical bins -> values to the last ulps of the interpolation
    z_all = np.concatenate([z, zf_dev], 1)
    order = np.argsort(z_all, axis=1, kind='stable')
    assert np.array_equal(merged_dev, order >= z.shape[1])                                          # which slots of the merged ray hold importance samples
    return n_diff, differ.size


def test_importance_bins_and_merge_pattern_are_integer_exact(hip_lib):
    rmod = _renderer()
    g = load_golden('renderer_importance')
    n_diff, total = _check_index_work(rmod, g['z'], g['w'], g['u'], allow_ties=False)
    assert total == 64 * 48 and n_diff == 0


def test_index_work_at_bench_size(hip_lib):
    """The same comparison on what the benchmark's launch actually samples: seg2cat, batch 4, 128^2 rays x 64+64 — the coarse weights come out of
    the fused kernel's own coarse pass (debug output), every one of the 65 536 rays is compared."""
    from model_cases import build_generator, uniforms
    rmod = _renderer()
    g = load_golden('model_full_seg2cat_128')
    G = build_generator('seg2cat', 'cuda', depth=tuple(int(v) for v in g['depth']))
    ws, c, nrr = torch.tensor(g['ws'], device='cuda'), torch.tensor(g['c'], device='cuda'), int(g['nrr'])
    rk = G.rendering_kwargs
    u_c, u_f = uniforms(g, ws.shape[0], nrr, rk)
    with torch.no_grad():
        planes = G.backbone.synthesis(ws, noise_mode='const')
        planes = planes.view(planes.shape[0], 3, 32, planes.shape[-2], planes.shape[-1])
        o, d = G.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), nrr)
        _, _, _, zf_fused, w_coarse, bins_fused = rmod.fused_render(planes, G.decoder, o, d, rk, u_c.to('cuda'), u_f.to('cuda'), debug='bins')
    n, m, sc = u_c.shape[0], u_c.shape[1], u_c.shape[2]
    z_c = R.sample_stratified(u_c.numpy().reshape(n, m, sc), rk['ray_start'], rk['ray_end']).reshape(n * m, sc)
    n_diff, total = _check_index_work(rmod, z_c, w_coarse.cpu().numpy(), u_f.numpy(), allow_ties=True)
    print(f'bench-size index work: {n_diff} of {total} draws are cdf ties that rounded apart')
    # and the fused launch itself produced the depths of the stand-alone sampler on its own coarse pass (same device function, same inputs up to the
    # coarse depths' last ulp, which only enter the final interpolation)
    zf_alone = rmod.importance_sample_native(torch.tensor(z_c, device='cuda'), w_coarse, u_f.to('cuda'), sort=True)
    assert float((zf_alone - zf_fused).abs().max()) <= 5e-7
    # ... and the FUSED launch's own integers (p3d_render_forward_debug: the bin index of every draw, written by the kernel that rendered): equal to the
    # stand-alone entry point's on every one of the 4.2 M draws, and to the oracle's searchsorted up to the cdf ties counted above
    _, bins_alone, _ = rmod.importance_sample_index_native(torch.tensor(z_c, device='cuda'), w_coarse, u_f.to('cuda'))
    assert bins_fused.shape == bins_alone.shape and torch.equal(bins_fused.to(bins_alone.dtype), bins_alone)
    bins_o, wp_o = R.importance_bins(z_c, w_coarse.cpu().numpy())
    _, inds_o = R.sample_pdf(bins_o, wp_o, u_f.numpy(), return_index=True)
    n_fused_diff = int((bins_fused.cpu().numpy().astype(np.int64) != inds_o).sum())
    assert n_fused_diff == n_diff, (n_fused_diff, n_diff)
    print(f'fused launch: bin indices of {bins_fused.numel()} draws == the stand-alone sampler, {n_fused_diff} cdf ties vs the oracle')


def test_edge_cases_empty_space_tail_tile_and_oob(hip_lib):
    """Zero density everywhere (depth -> nan -> clamp), a ray count that is not a multiple of 32, points far
    outside the box (all taps zero-padded)."""
    rmod = _renderer()
    g, opts, dec_arrays = load_case('seg')
    dec = make_decoder(g, 'cuda')
    n, m = 1, 37
    torch.manual_seed(0)
    planes = torch.randn(n, 3, 32, 8, 8, device='cuda')
    o = torch.tensor(np.concatenate([g['ray_o'][:1], g['ray_o'][:1, :1]], 1), device='cuda')          # 36 + 1 rays
    d = torch.tensor(np.concatenate([g['ray_d'][:1], g['ray_d'][1:2, :1]], 1), device='cuda')
    assert o.shape[1] == m
    d[:, -5:] = torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.1]], device='cuda'), dim=1)      # leaves the box sideways
    uc = torch.rand(n, m, opts['depth_resolution'], device='cuda')
    uf = torch.rand(n * m, opts['depth_resolution_importance'], device='cuda')
    feat, depth, wsum = rmod.fused_render(planes, dec, o, d, opts, uc, uf)
    fo, do, wo = R.render(planes.cpu().numpy(), dec_arrays, o.cpu().numpy(), d.cpu().numpy(), opts, uc.cpu().numpy(), uf.cpu().numpy())
    assert rel_err(feat.cpu().numpy(), fo) < 2e-4 and np.abs(depth.cpu().numpy()[..., 0] - do).max() < 5e-5
    # empty space: force the density bias very negative
    with torch.no_grad():
        dec.net_semantic[2].bias[0] = -1e4
    feat, depth, wsum = rmod.fused_render(planes, dec, o, d, opts, uc, uf)
    assert torch.all(wsum < 1e-6) and torch.isfinite(depth).all()
    assert torch.allclose(feat, torch.full_like(feat, -1.0), atol=1e-5)       # nothing composited -> 0*2-1
    assert torch.allclose(depth, depth.max().expand_as(depth))                 # nan -> +inf -> clamped to the far bound


def test_bench_sized_render_properties(hip_lib):
    """BASELINE config size (128^2 rays, 48+48 and 64+64 samples): invariants that do not need the oracle —
    outputs finite and in range, wsum in [0, 1], depth inside [ray_start, ray_end], determinism, and
    permutation equivariance over rays."""
    rmod = _renderer()
    g, opts, _ = load_case('seg')
    dec = make_decoder(g, 'cuda')
    from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    torch.manual_seed(1)
    planes = torch.randn(2, 3, 32, 256, 256, device='cuda')
    c2w = torch.tensor(g['c2w'][:2], device='cuda')
    K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]], device='cuda').repeat(2, 1, 1)
    o, d = RaySampler()(c2w, K, 128)
    for sc, sf in ((48, 48), (64, 64)):
        op = dict(opts, depth_resolution=sc, depth_resolution_importance=sf)
        uc = torch.rand(2, 128 * 128, sc, device='cuda')
        uf = torch.rand(2 * 128 * 128, sf, device='cuda')
        feat, depth, wsum = rmod.fused_render(planes, dec, o, d, op, uc, uf)
        assert torch.isfinite(feat).all() and torch.isfinite(depth).all()
        assert wsum.min() >= 0 and wsum.max() <= 1 + 1e-5
        assert depth.min() >= op['ray_start'] - 1e-5 and depth.max() <= op['ray_end'] + (op['ray_end'] - op['ray_start']) / (sc - 1) + 1e-4
        assert feat[..., :32].min() >= -1.0021 and feat[..., :32].max() <= 1.0021          # colours are squashed
        f2, d2, w2 = rmod.fused_render(planes, dec, o, d, op, uc, uf)
        assert torch.equal(feat, f2) and torch.equal(wsum, w2)                            # deterministic
        perm = torch.randperm(128 * 128, device='cuda')
        f3, _, w3 = rmod.fused_render(planes, dec, o[:, perm], d[:, perm], op, uc[:, perm], uf.view(2, -1, sf)[:, perm].reshape(-1, sf))
        assert torch.allclose(f3, feat[:, perm], atol=1e-6) and torch.allclose(w3, wsum[:, perm], atol=1e-6)
