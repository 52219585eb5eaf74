"""Small host-side pieces next to the hot path against records from the reference (tests/golden/make_golden.py group ``helpers``):
filtered_resizing in its four modes, sample_from_3dgrid, InfiniteSampler's rank-sharded index streams, math_utils."""
import numpy as np
import torch

from conftest import load_golden, rel_err


def test_filtered_resizing_modes():
    from pix2pix3d_amd.training.dual_discriminator import filtered_resizing
    from pix2pix3d_amd.torch_utils.ops import upfirdn2d
    g = load_golden('helpers')
    img, f = torch.tensor(g['fr.img']), upfirdn2d.setup_filter([1, 3, 3, 1])
    for mode in ('antialiased', 'classic', 'none', 0.3):
        assert rel_err(filtered_resizing(img, size=64, f=f, filter_mode=mode).numpy(), g[f'fr.{mode}.up']) < 1e-6, mode
        if mode != 'classic':
            assert rel_err(filtered_resizing(img, size=10, f=f, filter_mode=mode).numpy(), g[f'fr.{mode}.down']) < 1e-6, mode


def test_sample_from_3dgrid():
    from pix2pix3d_amd.training.volumetric_rendering.renderer import sample_from_3dgrid
    g = load_golden('helpers')
    out = sample_from_3dgrid(torch.tensor(g['g3.grid']), torch.tensor(g['g3.coords']))
    assert out.shape == g['g3.out'].shape and rel_err(out.numpy(), g['g3.out']) < 1e-6


def test_infinite_sampler_streams():
    from pix2pix3d_amd.torch_utils import misc
    g = load_golden('helpers')
    cases = [dict(rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5), dict(rank=1, num_replicas=3, shuffle=True, seed=7, window_size=0.5),
             dict(rank=2, num_replicas=4, shuffle=False), dict(rank=0, num_replicas=2, shuffle=True, seed=3, window_size=0)]
    for i, kw in enumerate(cases):
        it = iter(misc.InfiniteSampler(list(range(37)), **kw))
        assert np.array_equal(np.array([int(next(it)) for _ in range(120)]), g[f'sampler.{i}']), kw


def test_math_utils():
    from pix2pix3d_amd.training.volumetric_rendering import math_utils
    g = load_golden('helpers')
    o, d = torch.tensor(g['mu.o']), torch.tensor(g['mu.d'])
    near, far = math_utils.get_ray_limits_box(o, d, box_side_length=1.3)
    assert np.array_equal(near.numpy(), g['mu.near'], equal_nan=True) and np.array_equal(far.numpy(), g['mu.far'], equal_nan=True)
    lin = math_utils.linspace(near.clamp(-5, 5).nan_to_num(0), far.clamp(-5, 5).nan_to_num(1), 7)
    assert rel_err(lin.numpy(), g['mu.linspace']) < 1e-6
    tv = math_utils.transform_vectors(torch.tensor(g['mu.m']), torch.randn(50, 4, generator=torch.Generator().manual_seed(92)))
    assert rel_err(tv.numpy(), g['mu.tv']) < 1e-6
    assert rel_err(math_utils.normalize_vecs(o).numpy(), g['mu.nv']) < 1e-6 and rel_err(math_utils.torch_dot(o, d).numpy(), g['mu.dot']) < 1e-6


def test_native_bilinear_upsize_is_the_antialiased_interpolate():
    """dual_discriminator.bilinear_upsize (zero-insertion + triangle FIR on the upfirdn2d op) against torch's anti-aliased bilinear interpolate, which the
    reference's filtered_resizing calls for the discriminator's raw-image input (dual_discriminator.py:86-90): values, gradient and double backward."""
    import torch
    import torch.nn.functional as F
    from pix2pix3d_amd.training.dual_discriminator import bilinear_upsize
    torch.manual_seed(0)
    for s, n in ((4, 12), (2, 9), (8, 6)):
        x = torch.randn(2, 3, n, n, requires_grad=True)
        ref = F.interpolate(x, size=(n * s, n * s), mode='bilinear', align_corners=False, antialias=True)
        got = bilinear_upsize(x, s)
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-6
        g = torch.randn_like(ref)
        (ga,) = torch.autograd.grad(ref, x, g, create_graph=True)
        (gb,) = torch.autograd.grad(got, x, g, create_graph=True)
        assert float((ga - gb).abs().max()) < 1e-5
        x2 = torch.randn(2, 3, n, n, requires_grad=True)                      # double backward: d/dg of <grad(g), v> is the forward applied to v
        y = bilinear_upsize(x2, s)
        gg = torch.randn_like(y).requires_grad_(True)
        (gx,) = torch.autograd.grad(y, x2, gg, create_graph=True)
        v = torch.randn_like(gx)
        (d,) = torch.autograd.grad((gx * v).sum(), gg)
        assert float((d - F.interpolate(v, size=(n * s, n * s), mode='bilinear', align_corners=False)).abs().max()) < 2e-6


import pytest


@pytest.mark.gpu
def test_filtered_resizing_upsizes_on_the_native_kernels(hip_lib):
    """On the device the discriminator's raw-image resize (128^2 -> 512^2, config 3) runs on upfirdn2d kernels: against torch's anti-aliased interpolate
    on the same device — values, gradient, and the double backward R1 takes through it."""
    import torch
    import torch.nn.functional as F
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import upfirdn2d
    from pix2pix3d_amd.training import dual_discriminator as dd
    torch.manual_seed(1)
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
    for ch in (3, 9):
        x = torch.randn(4, ch, 128, 128, device='cuda', requires_grad=True)
        n0 = _lib.launch_count('upfirdn2d')
        got = dd.filtered_resizing(x, size=512, f=f)
        assert _lib.launch_count('upfirdn2d') > n0
        ref = F.interpolate(x, size=(512, 512), mode='bilinear', align_corners=False, antialias=True)
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 5e-6
        g = torch.randn_like(ref).requires_grad_(True)
        (ga,) = torch.autograd.grad(ref, x, g, create_graph=True)
        (gb,) = torch.autograd.grad(got, x, g, create_graph=True)
        assert float((ga - gb).abs().max()) < 5e-5 * float(ga.abs().max())
        (da,) = torch.autograd.grad(ga.square().sum(), g)                     # R1-style second differentiation: through the resize's backward
        (db,) = torch.autograd.grad(gb.square().sum(), g)
        assert float((da - db).abs().max()) <= 1e-4 * max(float(da.abs().max()), 1e-12)
    prev, dd.native_upsize = dd.native_upsize, False
    try:
        n0 = _lib.launch_count('upfirdn2d')
        dd.filtered_resizing(x.detach(), size=512, f=f)
        assert _lib.launch_count('upfirdn2d') == n0
    finally:
        dd.native_upsize = prev


def test_one_launch_training_layers_leave_the_cpu_path_alone():
    """torch_utils/ops/conv_layer.py is a device-only route: on CPU tensors its predicates decline and Conv2dLayer / FullyConnectedLayer / the demodulation
    coefficients run the operator-by-operator formulation; the tensor-operator demodulation it falls back to equals the reference's [N, O, I, k, k] product form
    (networks_stylegan2.py:57-63)."""
    from pix2pix3d_amd.torch_utils.ops import conv_layer, conv2d_gradfix
    from pix2pix3d_amd.training.networks_stylegan2 import Conv2dLayer, FullyConnectedLayer, _demod_coefficients
    g = torch.Generator().manual_seed(2)
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        conv = Conv2dLayer(32, 64, 3, activation='lrelu', down=2)
        x = torch.randn(2, 32, 16, 16, generator=g, requires_grad=True)
        assert not conv_layer.supported(x, conv.weight, conv.bias, 1, 2, 'lrelu')
        fc = FullyConnectedLayer(64, 32, activation='lrelu')
        z = torch.randn(3, 64, generator=g, requires_grad=True)
        assert not conv_layer.fc_supported(z, fc.weight, fc.bias, 'lrelu')
        w = torch.nn.Parameter(torch.randn(8, 6, 3, 3, generator=g)); s = torch.randn(2, 6, generator=g, requires_grad=True)
        assert not conv_layer.demod_supported(w, s)
        c0 = dict(conv_layer.calls)
        conv(x).sum().backward(); fc(z).sum().backward()
        d = _demod_coefficients(w, s)
        assert conv_layer.calls == c0 and conv.weight.grad is not None and fc.weight.grad is not None
    finally:
        conv2d_gradfix.enabled = prev
    ref = ((w.unsqueeze(0) * s.reshape(2, 1, 6, 1, 1)).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    assert float((d - ref).abs().max()) < 1e-6
