"""BASELINE config 3 at its real size against the reference's own gradients (tests/golden/make_golden.py groups ``train_full`` and ``greg``).

``train_full``: one Gmain-style pass — seg2cat, batch 2, G.mapping (label map -> Encoder -> ws) + G.synthesis in training mode at 128^2 rays x
48+48 samples, scalar loss over image / semantic / image_raw, backward through everything (loss.py:436-450, 509-).  ``greg``: the density
regularisation (loss.py:681-706): sigma = G.sample_mixed(points)['sigma'] at 1000 + 1000 perturbed points per image, L1 between the halves.
Recorded from the reference on the CPU (fp32 everywhere): loss, the gradient norm of EVERY parameter, heads of a dozen gradients.

Device legs (all under ``fused_policy = 'require'`` and ``conv2d_gradfix.enabled``: forward + backward of the renderer / the point queries are the
fused kernels, every convolution and gradient native): the shipped default (exact fp32 products) <= 2e-3, the bf16x3 opt-in <= 5e-3, and the
fp16 super-resolution heads of the GPU configuration at the fp16 class (3e-2)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from model_cases import replay_uniforms, uniforms, weights


def _build(device):
    from pix2pix3d_amd import configs, dnnlib
    kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
    weights.seed_module(G, seed=1)
    return G.to(device), kw


def _check_grads(G, g, tol, prefix='', head_tol=None):
    """Every parameter's gradient norm against the record (relative to its own size, floored at 1e-3 of the largest) + the recorded heads."""
    head_tol = tol if head_tol is None else head_tol
    params = dict(G.named_parameters())
    names = g['grad_names'].tolist()
    ref = g[prefix + 'grad_norms']
    assert names == [n for n, _ in G.named_parameters()]
    got = np.array([float(params[n].grad.double().norm()) if params[n].grad is not None else -1.0 for n in names])
    assert np.array_equal(got < 0, ref < 0), [n for n, a, b in zip(names, got, ref) if (a < 0) != (b < 0)]
    live = ref >= 0
    scalar = np.array([params[n].ndim == 0 for n in names])
    scale = np.maximum(ref, np.where(scalar, 2e-2, 1e-3) * ref.max())
    # a scalar parameter's gradient (noise_strength) is ONE signed sum over 10^2..10^5 products with heavy cancellation, not a norm: two fp32
    # summation orders differ by 1e-3 of it on the CPU already, so those entries get 10x the bound and a floor of 2e-2 (not 1e-3) of the largest norm
    loose = np.where(scalar, 10.0, 1.0)
    err = np.abs(got - ref) / scale / loose
    worst = int(np.argmax(np.where(live, err, 0)))
    assert err[live].max() < tol, (names[worst], got[worst], ref[worst])
    for i, nme in enumerate(g['head_names'].tolist()):
        head = params[nme].grad.reshape(-1)[:64].float().cpu().numpy()
        want = g[f'{prefix}h{i}']
        ref_norm = float(ref[names.index(nme)])
        floor = 1e-5 * float(ref[live].max()) / np.sqrt(params[nme].numel())       # a gradient that cancels analytically (d sigma_i - d sigma_p w.r.t. a bias) is 1e-10 of noise either way
        assert np.abs(head - want).max() < head_tol * max(np.abs(want).max(), ref_norm / np.sqrt(params[nme].numel()) * 3, 1e-12) + floor, (nme, np.abs(head - want).max(), np.abs(want).max())
    return float(err[live].max())


def _mapping(G, g, device):
    z, c = torch.tensor(g['z'], device=device), torch.tensor(g['c'], device=device)
    mask = torch.tensor(g['mask'].astype(np.int64), device=device)
    ws = G.mapping(z, c, {'mask': mask, 'pose': c}, update_emas=False)
    return ws, c


def _train_full(device, tol, force_fp32=True, out_tol=None, loss_scale=1.0):
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    g = load_golden('train_full_seg2cat')
    G, kw = _build(device)
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        ws, c = _mapping(G, g, device)
        assert np.abs(ws.detach().cpu().numpy() - g['ws']).max() < tol * np.abs(g['ws']).max()
        u_c, u_f = uniforms(g, 2, 128, kw['rendering_kwargs'])
        with replay_uniforms(u_c, u_f):
            out = G.synthesis(ws, c, neural_rendering_resolution=128, noise_mode='const', force_fp32=force_fp32)
        loss = out['image'].float().square().mean() + out['semantic'].float().square().mean() * 0.1 + out['image_raw'].square().mean()
        (loss * loss_scale).backward()
        if loss_scale != 1.0:
            for p_ in G.parameters():
                if p_.grad is not None:
                    p_.grad.div_(loss_scale)
    finally:
        conv2d_gradfix.enabled = prev
    out_tol = tol if out_tol is None else out_tol
    assert abs(loss.item() - float(g['loss'])) < out_tol * abs(float(g['loss']))
    for k in ('image_raw', 'semantic_raw', 'image', 'semantic'):
        m = out[k].detach().double().mean(dim=[2, 3]).cpu().numpy()
        assert np.abs(m - g[k + '_mean']).max() < out_tol * max(np.abs(g[k + '_mean']).max(), 0.1), k
    return _check_grads(G, g, tol)


def _greg(device, tol, which='sigma', head_tol=None):
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    g = load_golden('greg_seg2cat')
    G, kw = _build(device)
    rk = kw['rendering_kwargs']
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        ws, c = _mapping(G, g, device)
        coords = torch.tensor(g['coords'], device=device)
        res = G.sample_mixed(coords, torch.zeros_like(coords), ws, update_emas=False, noise_mode='const')
        sigma = res['sigma']
        if which == 'sigma':                                                        # loss.py:700-704
            half = sigma.shape[1] // 2
            loss = torch.nn.functional.l1_loss(sigma[:, :half], sigma[:, half:]) * rk['density_reg']
        else:
            loss = res['rgb'].square().mean() + res['sigma'].square().mean() * 1e-3
        loss.backward()
    finally:
        conv2d_gradfix.enabled = prev
    assert np.abs(sigma.detach().cpu().numpy() - g['sigma']).max() < tol * np.abs(g['sigma']).max()
    assert np.abs(res['rgb'].detach()[:, :8].cpu().numpy() - g['rgb_head']).max() < tol
    want = float(g['loss'] if which == 'sigma' else g['rgbloss'])
    assert abs(loss.item() - want) < tol * abs(want)
    return _check_grads(G, g, tol, prefix='' if which == 'sigma' else 'rgbloss_', head_tol=head_tol)


@pytest.mark.parametrize('which', ['sigma', 'rgb'])
def test_density_regularisation_matches_reference_cpu(which):
    """The product's CPU route (tensor ops under autograd) reproduces the reference's Greg phase."""
    _greg('cpu', 5e-4, which)


def test_config3_gradients_match_reference_cpu():
    """The product's CPU route at config 3's real size (batch 2, 128^2 rays x 48+48; ~40 s, ~15 GB): the record is consumable and the host
    logic (mapping -> synthesis wiring, loss, parameter order) is the reference's.  Measured worst gradient-norm error 2e-4."""
    _train_full('cpu', 5e-4)


class _native_training:
    """What the training loop sets (training_loop.py:281) + 'no tensor-op renderer on a device tensor'.  ``bf16x3``: the opt-in arithmetic of the
    fp32 training convolutions (P3D_TRAIN_BF16X3=1) instead of the default exact fp32 products."""

    def __init__(self, bf16x3):
        self.bf16x3 = bf16x3

    def __enter__(self):
        from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix, modconv
        from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
        self.mods = (conv2d_gradfix, modconv, rmod)
        self.prev = (conv2d_gradfix.split_bf16, modconv.split_bf16, rmod.fused_policy)
        rmod.fused_policy = 'require'
        conv2d_gradfix.split_bf16 = bool(self.bf16x3)
        modconv.split_bf16 = bool(self.bf16x3)
        return self

    def __exit__(self, *exc):
        conv2d_gradfix, modconv, rmod = self.mods
        conv2d_gradfix.split_bf16, modconv.split_bf16, rmod.fused_policy = self.prev


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['sigma', 'rgb'])
def test_density_regularisation_on_the_fused_point_kernels(hip_lib, which):
    """Greg on the device: p3d_sample_points forward, p3d_sample_points_backward under autograd (renderer._FusedPointsFn) — no tensor-op
    renderer (``require``), no vendor convolution.  Default (exact fp32) <= 2e-3; the bf16x3 opt-in <= 5e-3."""
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    for bf16x3, tol in ((False, 2e-3), (True, 5e-3)):
        b0, c0 = dict(rmod.backward_calls), dict(conv2d_gradfix.native_calls)
        with _native_training(bf16x3):
            _greg('cuda', tol, which, head_tol=4 * tol if bf16x3 else None)      # (single gradient ENTRIES 14 bf16x3 layers upstream: 1 % measured)
        assert rmod.backward_calls['points'] == b0['points'] + 1 and rmod.backward_calls['replay'] == b0['replay']
        assert conv2d_gradfix.native_calls['aten'] == c0['aten'], conv2d_gradfix.native_calls


@pytest.mark.gpu
def test_config3_gradients_default_exact_fp32(hip_lib):
    """The shipped default of a training run: exact fp32 products everywhere (conv2d_gradfix.split_bf16 is opt-in), fp32 SR heads here."""
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    assert not conv2d_gradfix.split_bf16 or __import__('os').environ.get('P3D_TRAIN_BF16X3') == '1'
    b0, c0 = dict(rmod.backward_calls), dict(conv2d_gradfix.native_calls)
    with _native_training(bf16x3=False):
        worst = _train_full('cuda', 2e-3)
    assert rmod.backward_calls['fused'] == b0['fused'] + 1 and rmod.backward_calls['replay'] == b0['replay']
    assert conv2d_gradfix.native_calls['aten'] == c0['aten'], conv2d_gradfix.native_calls
    print('worst gradient-norm error (exact fp32)', worst)


@pytest.mark.gpu
def test_config3_gradients_bf16x3_opt_in(hip_lib):
    """P3D_TRAIN_BF16X3=1: fp32 convolutions (forward + data gradient) as bf16x3, exact-fp32 weight gradients, exact-fp32 renderer."""
    with _native_training(bf16x3=True):
        worst = _train_full('cuda', 5e-3)
    print('worst gradient-norm error (bf16x3)', worst)


@pytest.mark.gpu
def test_config3_gradients_generator_scoped_bf16x3(hip_lib):
    """P3D_TRAIN_G_BF16X3=1 (training/triplane.py: train_products_bf16x3): the module default stays exact fp32, the generator's own passes ask
    conv2d_gradfix.products(True) for bf16x3 — forward and data gradients inherit it through the op's configuration, whenever the backward runs."""
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training import triplane
    prev, triplane.train_products_bf16x3 = triplane.train_products_bf16x3, True
    try:
        with _native_training(bf16x3=False):
            assert conv2d_gradfix._Cfg(False, (8, 8, 3, 3), 1, 1, 0, 1, 1).split is False            # a discriminator's convolution, called outside the generator
            with conv2d_gradfix.products(True):
                cfg = conv2d_gradfix._Cfg(False, (8, 8, 3, 3), 1, 1, 0, 1, 1)
            assert cfg.split is True and cfg.flipped((1, 8, 4, 4), (1, 8, 4, 4)).split is True      # ... and one called inside: its gradient op inherits the choice
            worst = _train_full('cuda', 5e-3)
    finally:
        triplane.train_products_bf16x3 = prev
    print('worst gradient-norm error (generator-scoped bf16x3)', worst)


@pytest.mark.gpu
def test_config3_gradients_fp16_sr_heads(hip_lib):
    """BASELINE config 3 as train.py configures it on a GPU: fp16 super-resolution heads (sr_num_fp16_res = 4, conv_clamp 256).  The
    reference record is fp32 (its CPU path), so this leg is held to the fp16 class.  The record's loss is a MEAN over 3 x 512^2 pixels, i.e.
    1e-6-sized image gradients, which fp16 tensors cannot carry (6e-5 is the smallest normal): the backward runs on loss x 256 and the
    gradients are divided back — linear, and what any fp16 training setup does when its gradients are that small (the training losses of
    loss.py put ~1e-3..1e-5 on the image through D and do not need it)."""
    with _native_training(bf16x3=False):
        worst = _train_full('cuda', 3e-2, force_fp32=False, loss_scale=256.0)
    print('worst gradient-norm error (fp16 SR heads)', worst)
