"""Conv2dLayer's one-launch training form (torch_utils/ops/conv_layer.py) against the unfused formulation it replaces — the reference's own sequence of
operators (networks_stylegan2.py:177-188: weight * gain, conv2d_resample, bias_act) on the same native kernels: output, every first-order gradient, and the
R1 pattern of loss.py:873-879 (data gradient recorded under no_weight_gradients, penalty differentiated to the parameters)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (name, Ci, Co, k, down, activation, bias, conv_clamp, gain, H, N)
CASES = [
    ('conv0', 64, 64, 3, 1, 'lrelu', True, 256, 1.0, 24, 2),
    ('conv1_down', 64, 128, 3, 2, 'lrelu', True, 256, float(np.sqrt(0.5)), 24, 2),
    ('skip', 64, 128, 1, 2, 'linear', False, None, float(np.sqrt(0.5)), 24, 2),
    ('pointwise', 128, 64, 1, 1, 'lrelu', True, None, 1.0, 12, 3),
    ('linear_clamped', 64, 64, 3, 1, 'linear', True, 4.0, 1.0, 12, 2),
    ('low_res', 512, 512, 3, 1, 'lrelu', True, 256, 1.0, 8, 4),
]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def run(layer, x0, fused, second_order):
    from pix2pix3d_amd.torch_utils.ops import conv_layer, conv2d_gradfix
    prev = conv_layer.enabled
    conv_layer.enabled = fused
    try:
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = layer(x, gain=layer._test_gain)
        out = dict(y=y.detach().clone())
        if second_order:                                                         # loss.py:873-879
            with conv2d_gradfix.no_weight_gradients():
                gx, = torch.autograd.grad(outputs=[y.float().sum()], inputs=[x], create_graph=True, only_inputs=True)
            (gx.float().square().sum() * 0.5 + y.float().mean()).backward()
            out['r1_field'] = gx.detach().clone()
        else:
            g = torch.Generator(device='cuda').manual_seed(3)
            (y.float() * torch.randn(y.shape, device='cuda', generator=g)).sum().backward()
            out['gx'] = x.grad.detach().clone()
        out['gw'] = layer.weight.grad.detach().clone()
        if layer.bias is not None:
            out['gb'] = layer.bias.grad.detach().clone()
        return out
    finally:
        conv_layer.enabled = prev


@pytest.mark.parametrize('second_order', [False, True], ids=['first_order', 'r1'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_one_launch_layer_matches_the_unfused_formulation(hip_lib, case, dtype, second_order):
    from pix2pix3d_amd.training.networks_stylegan2 import Conv2dLayer
    from pix2pix3d_amd.torch_utils.ops import conv_layer, conv2d_gradfix
    name, ci, co, k, down, act, bias, clamp, gain, h, n = case
    torch.manual_seed(7)
    layer = Conv2dLayer(ci, co, k, bias=bias, activation=act, down=down, conv_clamp=clamp).cuda()
    if bias:
        with torch.no_grad():
            layer.bias.copy_(torch.randn(co) * 0.3)
    layer._test_gain = gain
    x0 = torch.randn(n, ci, h, h, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        c0 = dict(conv_layer.calls)
        a = run(layer, x0, True, second_order)
        assert conv_layer.calls['forward'] == c0['forward'] + 1 and conv_layer.calls['backward'] >= c0['backward'] + 1
        b = run(layer, x0, False, second_order)
        assert conv_layer.calls['forward'] == c0['forward'] + 1
    finally:
        conv2d_gradfix.enabled = prev
    # fp32: the same products summed in another order.  fp16: the unfused form rounds the convolution's result to fp16 BEFORE the bias / activation pass, so
    # a few pre-activations per ten thousand sit on the other side of zero (the output moves by <= 1e-3 of its range, a gradient entry fed by a flipped
    # leaky-ReLU by a few per cent of the largest one — ten per cent where only 64 pixels feed a weight): the gradients are held to 3 % in L2
    assert set(a) == set(b)
    errs = {key: rel(a[key], b[key]) for key in a}
    if dtype == torch.float32:
        assert all(e < 2e-5 for e in errs.values()), errs
    else:
        l2 = {key: float((a[key].double() - b[key].double()).norm() / b[key].double().norm().clamp_min(1e-12)) for key in a}
        assert errs['y'] < 2e-3 and all(e < 0.2 for e in errs.values()) and all(e < 3e-2 for e in l2.values()), (errs, l2)


# (name, in, out, activation, bias, lr_multiplier, N)
FC_CASES = [('affine', 512, 512, 'linear', True, 1.0, 4), ('mapping', 512, 512, 'lrelu', True, 0.01, 4), ('camera', 25, 512, 'linear', True, 1.0, 4),
            ('logit', 512, 1, 'linear', True, 1.0, 4), ('narrow', 512, 96, 'lrelu', False, 1.0, 3)]


@pytest.mark.parametrize('second_order', [False, True], ids=['first_order', 'second_order'])
@pytest.mark.parametrize('case', FC_CASES, ids=[c[0] for c in FC_CASES])
def test_one_launch_fully_connected_layer_matches_the_unfused_formulation(hip_lib, case, second_order):
    """FullyConnectedLayer in training passes (conv_layer.fc_layer) against the route it replaces (weight * gain, 1x1 convolution of a 1x1 image, bias_act): output,
    gradients, and a gradient-of-gradient pattern (d/dparams of |d sum(y) / dx|^2 — what R1 asks of the discriminator epilogue's layers)."""
    from pix2pix3d_amd.training.networks_stylegan2 import FullyConnectedLayer
    from pix2pix3d_amd.torch_utils.ops import conv_layer, conv2d_gradfix
    name, fin, fout, act, bias, lrm, n = case
    torch.manual_seed(11)
    layer = FullyConnectedLayer(fin, fout, bias=bias, activation=act, lr_multiplier=lrm, bias_init=0.3).cuda()
    x0 = torch.randn(n, fin, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(5)
    probe = torch.randn(n, fout, device='cuda', generator=g)

    def run(fused):
        prev = conv_layer.enabled
        conv_layer.enabled = fused
        try:
            for p in layer.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y = layer(x)
            out = dict(y=y.detach().clone())
            if second_order:
                gx, = torch.autograd.grad(outputs=[(y * probe).sum()], inputs=[x], create_graph=True, only_inputs=True)
                (gx.square().sum() * 0.5 + y.mean()).backward()
                out['field'] = gx.detach().clone()
            else:
                (y * probe).sum().backward()
                out['gx'] = x.grad.detach().clone()
            out['gw'] = layer.weight.grad.detach().clone()
            if bias:
                out['gb'] = layer.bias.grad.detach().clone()
            return out
        finally:
            conv_layer.enabled = prev
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        c0 = dict(conv_layer.calls)
        a = run(True)
        assert conv_layer.calls['forward'] > c0['forward'] and conv_layer.calls['backward'] > c0['backward']
        c1 = dict(conv_layer.calls)
        b = run(False)
        assert conv_layer.calls == c1
    finally:
        conv2d_gradfix.enabled = prev
    assert set(a) == set(b)
    errs = {key: rel(a[key], b[key]) for key in a}
    assert all(e < 2e-5 for e in errs.values()), errs


@pytest.mark.parametrize('shape', [(512, 512, 3, 4), (96, 256, 1, 4), (64, 128, 3, 2), (33, 40, 3, 3)], ids=['backbone', 'torgb_like', 'sr', 'odd'])
def test_demodulation_coefficients_and_their_gradient(hip_lib, shape):
    """conv_layer.demod (one kernel; two for the gradient) against the tensor-operator formulation of networks_stylegan2.py:57-63, and — with create_graph — its
    second-order path against the same formulation differentiated twice."""
    from pix2pix3d_amd.torch_utils.ops import conv_layer, conv2d_gradfix
    co, ci, k, n = shape
    g = torch.Generator(device='cuda').manual_seed(co + ci)
    weight = torch.nn.Parameter(torch.randn(co, ci, k, k, device='cuda', generator=g))
    s0 = torch.randn(n, ci, device='cuda', generator=g) + 1.0
    probe = torch.randn(n, co, device='cuda', generator=g)
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    try:
        assert conv_layer.demod_supported(weight, s0.requires_grad_(True))
        out = {}
        for name, fn in (('native', conv_layer.demod), ('ref', conv_layer.demod_reference)):
            weight.grad = None
            s = s0.detach().clone().requires_grad_(True)
            d = fn(weight, s)
            (d * probe).sum().backward()
            out[name] = (d.detach(), s.grad.clone(), weight.grad.clone())
            weight.grad = None
            s = s0.detach().clone().requires_grad_(True)
            gs, = torch.autograd.grad((fn(weight, s) * probe).sum(), [s], create_graph=True)
            gs.square().sum().backward()
            out[name] += (gs.detach(), s.grad.clone(), weight.grad.clone())
    finally:
        conv2d_gradfix.enabled = prev
    errs = [rel(a, b) for a, b in zip(out['native'], out['ref'])]
    assert all(e < 2e-5 for e in errs), errs
