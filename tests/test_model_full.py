"""Parity at the sizes the benchmark and the BASELINE configurations actually run (SURVEY §8 shape table): 128^2 rays, batch 4,
48+48 samples (cfg 2/3) and 64+64 (the metric's "128 depth"), and seg2face with its 19 label channels (cfg 5) — against
records of the reference generator at those sizes (tests/golden/model_full_*.npz, written by make_golden.py model_full) and,
for the ray-marcher alone, against the oracle's torch renderer on every ray of the batch.

At these sizes the fused kernel takes the schedule bench.py times (R = 128 raster, XCD column strips, 8 waves per block), which
the small renderer goldens never reach.

Tolerances: rendered pixels <= 1e-3 relative-to-max, depth <= 1e-4 absolute, SR images <= 1e-3 (all-fp32) / 3e-2 (fp16 SR
blocks, the reference's GPU precision); per-image means of every output (which see all 16 384 rays) to the same bounds; per-tile
abs-maxima and sums of EVERY output over its whole area (8x8 tiles at 512^2: all 262 144 pixels of every channel are in a record)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, record_error
from model_cases import build_generator, uniforms, replay_uniforms

FULL = {'seg2cat_96': 'seg2cat', 'seg2cat_128': 'seg2cat', 'seg2face_96': 'seg2face', 'edge2car_128': 'edge2car'}


def compare_full(out, g, tol_raw, tol_sr, tol_depth=1e-4):
    errs = {}
    for k in ('image_raw', 'semantic_raw', 'image_depth', 'image', 'semantic'):
        t = out[k].float().cpu()
        step, h = int(g[k + '_step']), t.shape[-1]
        c0 = h // 2 - 16
        scale = 1.0 if k == 'image_depth' else max(float(g[k + '_absmax']), 1e-30)
        e = [np.abs(t[..., ::step, ::step].numpy() - g[k + '_thumb']).max(),
             np.abs(t[..., c0:c0 + 32, c0:c0 + 32].numpy() - g[k + '_crop']).max(),
             np.abs(t.double().mean(dim=[2, 3]).numpy() - g[k + '_mean']).max()]
        errs[k] = float(max(e) / scale)
        tol = tol_depth if k == 'image_depth' else (tol_raw if k.endswith('_raw') else tol_sr)
        assert errs[k] < tol, (k, errs[k], e)
        # EVERY pixel: per-tile abs-max and sum over the whole output (8x8 tiles of the 512^2 images, 4x4 of the 128^2 renderings) — a defect in any
        # tile of any channel moves one of these records.  abs-max to the pixel tolerance; a tile's sum to tile^2 / 2 x tolerance (every pixel of the
        # tile off by half the tolerance in one direction; a full row or column off by the tolerance — a patch-edge defect of a convolution kernel — is a
        # quarter of that for the 8 x 8 tiles)
        tile = int(g[k + '_tile'])
        v = t.double().reshape(t.shape[0], t.shape[1], h // tile, tile, h // tile, tile)
        e_max = float(np.abs(v.abs().amax(dim=(3, 5)).numpy() - g[k + '_tile_max']).max() / scale)
        e_sum = float(np.abs(v.sum(dim=(3, 5)).numpy() - g[k + '_tile_sum']).max() / scale)
        errs[k + '.tile_max'], errs[k + '.tile_sum'] = e_max, e_sum
        # (measured on an MI355X, profiles/round5_*_parity_errors.json: tile sums <= 5.9e-4 in the fp32 legs, <= 3.6e-2 with fp16 SR heads — whose rounding
        # errors are correlated over a tile — against 3.2e-3 / 9.6e-2 here)
        assert e_max < tol and e_sum < tile * tile * tol / 2, (k, 'tile records', e_max, e_sum)
    return errs


def _inputs(tag, device):
    g = load_golden('model_full_' + tag)
    depth = tuple(int(v) for v in g['depth'])
    G = build_generator(FULL[tag], device, depth=depth)
    ws, c, nrr = torch.tensor(g['ws'], device=device), torch.tensor(g['c'], device=device), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)
    return g, G, ws, c, nrr, u_c, u_f


def test_oracle_matches_reference_at_seg2face_size():
    """The oracle itself at a real BASELINE size: seg2face (19 label channels), 128^2 rays x 48+48, batch 2."""
    from oracle import model_oracle as M
    from pix2pix3d_amd import configs
    g, G, ws, c, nrr, u_c, u_f = _inputs('seg2face_96', 'cpu')
    sd = {k: v.float() for k, v in G.state_dict().items()}
    with torch.no_grad():
        out = M.synthesis(sd, configs.oracle_cfg('seg2face', depth=(48, 48)), ws, c, u_c, u_f, nrr=nrr, noise_mode='const')
    assert out['semantic'].shape == (2, 19, 512, 512)
    print(compare_full(out, g, tol_raw=2e-4, tol_sr=2e-4))


def test_oracle_matches_reference_at_edge2car_size():
    """BASELINE configs[3] per GPU: edge2car (edge maps, white background, sigmoid label channel, Hybrid2X heads), 64^2 rays x 64+64, batch 8 -> 128^2."""
    from oracle import model_oracle as M
    from pix2pix3d_amd import configs
    g, G, ws, c, nrr, u_c, u_f = _inputs('edge2car_128', 'cpu')
    sd = {k: v.float() for k, v in G.state_dict().items()}
    with torch.no_grad():
        out = M.synthesis(sd, configs.oracle_cfg('edge2car', depth=(64, 64)), ws, c, u_c, u_f, nrr=nrr, noise_mode='const')
    assert out['image'].shape == (8, 3, 128, 128) and out['semantic'].shape == (8, 1, 128, 128)
    print(compare_full(out, g, tol_raw=2e-4, tol_sr=2e-4))


def test_product_cpu_path_matches_reference_at_seg2face_size():
    g, G, ws, c, nrr, u_c, u_f = _inputs('seg2face_96', 'cpu')
    with replay_uniforms(u_c, u_f), torch.no_grad():
        out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
    print(compare_full(out, g, tol_raw=2e-4, tol_sr=2e-4))


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(FULL))
@pytest.mark.parametrize('force_fp32', [True, False])
def test_synthesis_at_baseline_size_matches_reference(hip_lib, tag, force_fp32):
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    g, G, ws, c, nrr, u_c, u_f = _inputs(tag, 'cuda')
    n0 = _lib.launch_count('render')
    prev_pol, rmod.fused_policy = rmod.fused_policy, 'require'
    prev_en, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        with replay_uniforms(u_c, u_f), torch.no_grad():
            out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const', force_fp32=force_fp32)
        torch.cuda.synchronize()
    finally:
        rmod.fused_policy, conv2d_gradfix.enabled = prev_pol, prev_en
    assert _lib.launch_count('render') > n0
    # Bounds = 3 x the worst value this suite has measured on an MI355X (profiles/round5_a_parity_errors.json: fp32 legs <= 1.3e-5, fp16-SR legs
    # <= 9.4e-4 of the range on every record, tile records included) — a 1 %-level defect of an fp16 kernel (conv3x3_h2_f16 / up2_fir_f16) fails here.
    # edge2car + fp16 heads: the no-upsampling SR block adds its fp16 ToRGB output into 'image_raw' IN PLACE (reference quirk, superresolution.py:281),
    # so the "raw" images carry fp16 rounding there.
    fp16_raw = tag.startswith('edge2car') and not force_fp32
    errs = compare_full(out, g, tol_raw=5e-3 if fp16_raw else 1e-4, tol_sr=1e-4 if force_fp32 else (8e-3 if fp16_raw else 3e-3))
    print(tag, 'fp32' if force_fp32 else 'fp16-sr', errs)
    record_error(f'model_full.{tag}.' + ('fp32' if force_fp32 else 'fp16-sr'), errs)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['seg2cat_96', 'seg2cat_128'])
def test_ray_marcher_at_bench_size_matches_oracle_on_every_ray(hip_lib, tag):
    """The fused forward on the real 256^2 x 96 planes of a batch of 4, 128^2 rays: every ray of every image against the
    oracle's torch renderer (same planes, rays and replayed draws)."""
    from oracle import model_oracle as M
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    g, G, ws, c, nrr, u_c, u_f = _inputs(tag, 'cuda')
    rk = G.rendering_kwargs
    with torch.no_grad():
        planes = G.backbone.synthesis(ws, noise_mode='const')
        n = planes.shape[0]
        planes = planes.view(n, 3, 32, planes.shape[-2], planes.shape[-1])
        o, d = G.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), nrr)
        feat, depth, wsum = rmod.fused_render(planes, G.decoder, o, d, rk, u_c.to('cuda'), u_f.to('cuda'))
    sd = {k: v.float().cpu() for k, v in G.state_dict().items()}
    with torch.no_grad():
        fo, do, wo = M.render(sd, planes.float().cpu().contiguous(), o.cpu(), d.cpu(), rk, u_c, u_f, two_nets=True, sem_sigmoid=False,
                              lr_mul=rk.get('decoder_lr_mul', 1.0))
    e_feat = rel_err(feat.cpu().numpy(), fo.numpy())
    e_depth = float((depth.cpu().reshape(do.shape) - do).abs().max())
    e_w = rel_err(wsum.cpu().numpy().reshape(wo.shape), wo.numpy())
    print(tag, dict(feat=e_feat, depth=e_depth, wsum=e_w))
    # measured ~1e-5 (bf16x3 decoder) / ~2e-6 (exact): 1e-4 also catches the v_cvt_pk_bf16_f32 -> MFMA hazard (5 % of the rays off by 1e-3) at a tenth of its amplitude
    assert e_feat < 1e-4 and e_depth < 1e-4 and e_w < 1e-4
    # a flipped importance bin moves a sample by >= 1e-3 of the ray: count rays whose depth differs visibly
    bad = ((depth.cpu().reshape(do.shape) - do).abs() > 2e-5).float().mean().item()
    assert bad < 1e-3, bad
