"""Training-mode forward + backward of the generator (unfused modulation, differentiable tensor-op renderer, custom
autograd Functions of bias_act / upfirdn2d / conv2d_gradfix / fma) against gradients recorded from the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from model_cases import weights


def _run(device, golden='train_seg2cat'):
    from pix2pix3d_amd import configs, dnnlib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    g = load_golden(golden)
    with_depth = golden == 'train_seg2cat'
    kw = configs.generator_kwargs('seg2cat')
    kw['rendering_kwargs'] = dict(kw['rendering_kwargs'], depth_resolution=8, depth_resolution_importance=8)
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
    weights.seed_module(G, seed=1)
    G = G.to(device)
    ws, c = torch.tensor(g['ws'], device=device), torch.tensor(g['c'], device=device)
    # the reference drew its two uniform tensors from the CPU generator; replay the same values on any device
    torch.manual_seed(int(g['render_seed']))
    u_c = torch.rand([1, 256, 8, 1]); u_f = torch.rand([256, 8])
    from model_cases import replay_uniforms
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        with replay_uniforms(u_c, u_f):
            out = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const', force_fp32=True)
        loss = out['image'].mean() + out['image_raw'].square().mean() + out['semantic'].square().mean() * 0.1
        if with_depth:
            loss = loss + out['image_depth'].mean()
        loss.backward()
    finally:
        conv2d_gradfix.enabled = prev
    return g, G, loss


def _check(g, G, loss, tol):
    assert abs(loss.item() - float(g['loss'])) < tol * max(abs(float(g['loss'])), 1.0)
    params = dict(G.named_parameters())
    for i, name in enumerate(g['names'].tolist()):
        grad = params[name].grad
        assert grad is not None, name
        assert abs(grad.norm().item() - float(g[f'g{i}.norm'])) < tol * max(float(g[f'g{i}.norm']), 1e-6), name
        head = grad.reshape(-1)[:64].float().cpu().numpy()
        assert np.abs(head - g[f'g{i}.head']).max() < tol * max(np.abs(g[f'g{i}.head']).max(), float(g[f'g{i}.norm']) * 1e-3, 1e-12), name


@pytest.mark.parametrize('golden', ['train_seg2cat', 'train_seg2cat_nodepth'])
def test_training_gradients_match_reference_on_cpu(golden):
    g, G, loss = _run('cpu', golden)
    _check(g, G, loss, 2e-4)


@pytest.mark.gpu
def test_training_gradients_match_reference_on_gpu(hip_lib):
    """Same step on the device: bias_act / upfirdn2d gradient kernels are HIP, convolutions' gradients ATen."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    n0 = (_lib.launch_count('bias_act'), _lib.launch_count('upfirdn2d'), _lib.launch_count('render'))
    assert rmod.fused_training
    g, G, loss = _run('cuda')
    assert _lib.launch_count('bias_act') > n0[0] and _lib.launch_count('upfirdn2d') > n0[1]
    assert _lib.launch_count('render') == n0[2] + 1, 'training forward did not go through the fused ray-marcher'
    _check(g, G, loss, 2e-3)
    # and the all-tensor-op route gives the same gradients
    rmod.fused_training = False
    try:
        g2, G2, loss2 = _run('cuda')
    finally:
        rmod.fused_training = True
    _check(g2, G2, loss2, 2e-3)


@pytest.mark.gpu
def test_training_step_goes_through_the_fused_renderer_backward(hip_lib):
    """A loss over images only (what training/loss.py differentiates): forward AND backward of the renderer are the fused
    kernels (p3d_render_forward, p3d_render_backward), gradients as the reference's autograd gives them."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    n0, b0 = _lib.launch_count('render'), dict(rmod.backward_calls)
    g, G, loss = _run('cuda', 'train_seg2cat_nodepth')
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    assert rmod.backward_calls['fused'] == b0['fused'] + 1 and rmod.backward_calls['replay'] == b0['replay'], rmod.backward_calls
    assert conv2d_gradfix.native_calls['aten'] == 0 or conv2d_gradfix.native_calls['forward'] > 0
    assert _lib.launch_count('render') >= n0 + 2          # the forward entry point + the backward entry point (two launches, counted once)
    _check(g, G, loss, 2e-3)
