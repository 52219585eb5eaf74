"""Entry points and switches around the hot path that the other model tests leave at their defaults (truncation, w_avg tracking, plane
cache, random noise, camera-conditioning switches, density noise, G.forward / G.sample, the discriminator's camera noise) against records
from the reference (tests/golden/make_golden.py group ``api``).  CPU: every random draw is the CPU generator's, replayed from the seed."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from model_cases import weights

TOL = 5e-5


@pytest.fixture(scope='module')
def setup():
    from pix2pix3d_amd import configs, dnnlib
    g = load_golden('model_api')
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**configs.generator_kwargs('edge2car', cbase=1024, cmax=16)).eval().requires_grad_(False)
    weights.seed_module(G, seed=13)
    c, z, mask = torch.tensor(g['c']), torch.tensor(g['z']), torch.tensor(g['mask'])
    return G, g, c, z, {'mask': mask, 'pose': c}


def test_truncation_and_w_avg_tracking(setup):
    G, g, c, z, batch = setup
    with torch.no_grad():
        assert rel_err(G.mapping(z, c, batch, truncation_psi=0.4).numpy(), g['ws_trunc_all']) < TOL
        before = G.backbone.mapping.w_avg.clone()
        G.mapping(z, c, batch, update_emas=True)
        after = G.backbone.mapping.w_avg.clone()
        G.backbone.mapping.w_avg.copy_(before)
        assert rel_err(after.numpy(), g['w_avg_after']) < TOL and not torch.equal(before, after)
        assert rel_err(G.mapping(z, c, batch).numpy(), g['ws']) < TOL
        # the reference's disentangled mapping networks keep a per-layer w_avg and cannot apply a cutoff (triplane_cond.py:591 raises)
        with pytest.raises(RuntimeError):
            G.mapping(z, c, batch, truncation_psi=0.6, truncation_cutoff=9)


def test_camera_conditioning_switches(setup):
    G, g, c, z, batch = setup
    rk = G.rendering_kwargs
    try:
        with torch.no_grad():
            rk['c_gen_conditioning_zero'] = True
            assert rel_err(G.mapping(z, c, batch).numpy(), g['ws_czero']) < TOL
            rk['c_gen_conditioning_zero'] = False
            rk['c_scale'] = 0.25
            assert rel_err(G.mapping(z, c, batch).numpy(), g['ws_cscale']) < TOL
    finally:
        rk['c_gen_conditioning_zero'], rk['c_scale'] = False, 1.0


def test_forward_cache_and_random_noise(setup):
    G, g, c, z, batch = setup
    ws = torch.tensor(g['ws'])
    with torch.no_grad():
        torch.manual_seed(5)
        out = G(z, c, batch, neural_rendering_resolution=16, noise_mode='const')
        assert rel_err(out['image_raw'].numpy(), g['fwd_image_raw']) < TOL and rel_err(out['semantic_raw'].numpy(), g['fwd_semantic_raw']) < TOL
        torch.manual_seed(5)
        G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const', cache_backbone=True)
        assert G._last_planes is not None
        torch.manual_seed(5)
        out2 = G.synthesis(ws.flip(0), c, neural_rendering_resolution=16, noise_mode='const', use_cached_backbone=True)
        G._last_planes = None
        assert rel_err(out2['image_raw'].numpy(), g['cached_image_raw']) < TOL and rel_err(out2['image'][..., ::4, ::4].numpy(), g['cached_image_thumb']) < 2e-4
        torch.manual_seed(6)
        out3 = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='random')
        assert rel_err(out3['image_raw'].numpy(), g['rand_image_raw']) < TOL and rel_err(out3['image'][..., ::4, ::4].numpy(), g['rand_image_thumb']) < 2e-4


def test_sample_with_density_noise(setup):
    G, g, c, z, batch = setup
    try:
        G.rendering_kwargs['density_noise'] = 0.5
        with torch.no_grad():
            torch.manual_seed(7)
            sm = G.sample(torch.tensor(g['pts']), None, z, c, batch, noise_mode='const')
        assert rel_err(sm['sigma'].numpy(), g['sample_sigma']) < TOL and rel_err(sm['rgb'].numpy(), g['sample_rgb']) < TOL
    finally:
        G.rendering_kwargs['density_noise'] = 0


def test_discriminator_camera_noise(setup):
    from pix2pix3d_amd import dnnlib
    _, g, c, _, _ = setup
    D = dnnlib.util.construct_class_by_name(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=128, img_channels=3,
                                            channel_base=1024, channel_max=16, num_fp16_res=0, conv_clamp=None, disc_c_noise=0.5).eval().requires_grad_(False)
    weights.seed_module(D, seed=14)
    with torch.no_grad():
        torch.manual_seed(8)
        logits = D({'image': torch.tensor(g['d_image']), 'image_raw': torch.tensor(g['d_image_raw'])}, c.clone())
    assert rel_err(logits.numpy(), g['d_logits_cnoise']) < TOL


@pytest.mark.gpu
def test_switches_on_the_device_path(setup):
    """The same switches on device tensors: deterministic ones against the records (fp32 forced: 1e-3), the ones that draw on the device
    generator (random noise, density noise) for shape / finiteness, and the plane cache against its own uncached run."""
    import copy
    G0, g, c, z, batch = setup
    G = copy.deepcopy(G0).to('cuda')
    c, z = c.cuda(), z.cuda()
    batch = {'mask': batch['mask'].cuda(), 'pose': c}
    ws = torch.tensor(g['ws']).cuda()
    rk = G.rendering_kwargs
    with torch.no_grad():
        assert rel_err(G.mapping(z, c, batch, truncation_psi=0.4).cpu().numpy(), g['ws_trunc_all']) < 1e-3
        rk['c_gen_conditioning_zero'] = True
        assert rel_err(G.mapping(z, c, batch).cpu().numpy(), g['ws_czero']) < 1e-3
        rk['c_gen_conditioning_zero'] = False
        G.mapping(z, c, batch, update_emas=True)
        assert rel_err(G.backbone.mapping.w_avg.cpu().numpy(), g['w_avg_after']) < 1e-3
        torch.manual_seed(5)
        ref = G.synthesis(ws.flip(0), c, neural_rendering_resolution=16, noise_mode='const', force_fp32=True)
        torch.manual_seed(5)
        G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const', force_fp32=True, cache_backbone=True)
        torch.manual_seed(5)
        out = G.synthesis(ws.flip(0), c, neural_rendering_resolution=16, noise_mode='const', force_fp32=True, use_cached_backbone=True)
        G._last_planes = None
        assert not torch.allclose(out['image_raw'], ref['image_raw'])                      # planes of ws, SR heads of ws.flip
        out_r = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='random')
        assert out_r['image'].shape == (2, 3, 128, 128) and torch.isfinite(out_r['image'].float()).all()
        rk['density_noise'] = 0.5
        sm = G.sample(torch.tensor(g['pts']).cuda(), None, z, c, batch, noise_mode='const')
        rk['density_noise'] = 0
        assert sm['sigma'].shape == g['sample_sigma'].shape and torch.isfinite(sm['sigma']).all()
        assert rel_err(sm['rgb'].cpu().numpy(), g['sample_rgb']) < 1e-3                   # colours do not see the density noise
