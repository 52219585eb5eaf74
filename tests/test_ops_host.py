"""Host-side operator mirror on CPU tensors: API surface, CPU dispatch path, autograd structure.
Checked against vectors recorded from the reference (tests/golden)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from pix2pix3d_amd.torch_utils.ops import bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix, fma

ACTS = list(bias_act.activation_funcs.keys())


def _opt(v):
    v = float(v)
    return None if v < 0 else v


def test_activation_table_matches_reference_contract():
    assert ACTS == ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
    assert [bias_act.activation_funcs[a].cuda_idx for a in ACTS] == list(range(1, 10))
    assert bias_act.activation_funcs['lrelu'].def_alpha == 0.2
    assert abs(bias_act.activation_funcs['lrelu'].def_gain - 2 ** 0.5) < 1e-12
    assert bias_act.activation_funcs['swish'].ref == 'x' and bias_act.activation_funcs['relu'].has_2nd_grad is False


@pytest.mark.parametrize('act', ACTS)
def test_bias_act_cpu_path(act):
    g = load_golden('ops_bias_act')
    x = torch.tensor(g[f'{act}.x'], requires_grad=True)
    b = torch.tensor(g[f'{act}.b'], requires_grad=True)
    y = bias_act.bias_act(x, b, dim=1, act=act, alpha=_opt(g[f'{act}.alpha']), gain=_opt(g[f'{act}.gain']), clamp=_opt(g[f'{act}.clamp']))
    assert rel_err(y.detach().numpy(), g[f'{act}.y']) < 1e-12
    dx, db = torch.autograd.grad(y, [x, b], torch.tensor(g[f'{act}.dy']))
    assert rel_err(dx.numpy(), g[f'{act}.dx']) < 1e-12 and rel_err(db.numpy(), g[f'{act}.db']) < 1e-12


def test_upfirdn2d_cpu_path_and_helpers():
    g = load_golden('ops_upfirdn2d')
    for i in range(int(g['num_cases'])):
        f = g[f'{i}.f']
        f = None if f.size == 0 else torch.tensor(f)
        y = upfirdn2d.upfirdn2d(torch.tensor(g[f'{i}.x']), f, up=g[f'{i}.up'].tolist(), down=g[f'{i}.down'].tolist(),
                                padding=g[f'{i}.pad'].tolist(), flip_filter=bool(g[f'{i}.flip']), gain=float(g[f'{i}.gain']))
        assert tuple(y.shape) == g[f'{i}.y'].shape and rel_err(y.numpy(), g[f'{i}.y']) < 2e-6, i
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    assert rel_err(f.numpy(), g['h.f']) < 1e-7 and f.shape == (4, 4)
    x = torch.tensor(g['h.x'])
    assert rel_err(upfirdn2d.upsample2d(x, f).numpy(), g['h.up']) < 2e-6
    assert rel_err(upfirdn2d.downsample2d(x, f).numpy(), g['h.down']) < 2e-6
    assert rel_err(upfirdn2d.filter2d(x, f).numpy(), g['h.filt']) < 2e-6
    assert rel_err(upfirdn2d.setup_filter([1, 4, 6, 4, 1, 2, 3, 5], gain=2.0, flip_filter=True).numpy(), g['h.f_sep']) < 1e-6
    assert upfirdn2d._parse_padding([1, 2]) == (1, 1, 2, 2) and upfirdn2d._get_filter_size(None) == (1, 1)


def test_conv2d_resample_routes_on_cpu():
    g = load_golden('ops_conv')
    f = torch.tensor(g['f'])
    for i in range(int(g['num_resample'])):
        k, up, down, flipw = g[f'r{i}.cfg'].tolist()
        y = conv2d_resample.conv2d_resample(torch.tensor(g[f'r{i}.x']), torch.tensor(g[f'r{i}.w']), f=f, up=up, down=down,
                                            padding=g[f'r{i}.pad'].tolist(), flip_weight=bool(flipw))
        assert rel_err(y.numpy(), g[f'r{i}.y']) < 5e-6, i


def test_conv2d_gradfix_custom_function_on_cpu_tensors():
    """The custom autograd Functions are device-agnostic: drive them directly on CPU and compare all
    gradient orders with torch's own autograd (the contract of conv2d_gradfix.py:107-194)."""
    torch.manual_seed(0)
    for transpose, stride, pad in [(False, 1, 1), (False, 2, 1), (True, 2, 0), (True, 2, 1), (False, 1, 0)]:
        cin, cout, k = 3, 4, (1 if pad == 0 and not transpose else 3)
        x = torch.randn(2, cin, 7, 7, dtype=torch.float64, requires_grad=True)
        w = torch.randn(*((cin, cout, k, k) if transpose else (cout, cin, k, k)), dtype=torch.float64, requires_grad=True)
        cfg = conv2d_gradfix._Cfg(transpose, w.shape, stride, pad, 0, 1, 1)
        y = conv2d_gradfix._Conv.apply(x, w, None, cfg)
        fn = torch.nn.functional.conv_transpose2d if transpose else torch.nn.functional.conv2d
        yr = fn(x, w, None, stride=stride, padding=pad)
        assert torch.allclose(y, yr, atol=1e-10)
        gy = torch.randn_like(y)
        gx, gw = torch.autograd.grad(y, [x, w], gy, create_graph=True)
        gxr, gwr = torch.autograd.grad(yr, [x, w], gy, create_graph=True)
        assert torch.allclose(gx, gxr, atol=1e-10) and torch.allclose(gw, gwr, atol=1e-10)
        # R1-style second order: d/dw and d/dx of |gx|^2
        s, sr = gx.square().sum() + gw.square().sum(), gxr.square().sum() + gwr.square().sum()
        h = torch.autograd.grad(s, [x, w])
        hr = torch.autograd.grad(sr, [x, w])
        assert torch.allclose(h[0], hr[0], atol=1e-8) and torch.allclose(h[1], hr[1], atol=1e-8)
    with conv2d_gradfix.no_weight_gradients():
        assert conv2d_gradfix.weight_gradients_disabled
        x = torch.randn(1, 2, 5, 5, requires_grad=True)
        w = torch.randn(3, 2, 3, 3, requires_grad=True)
        y = conv2d_gradfix._Conv.apply(x, w, None, conv2d_gradfix._Cfg(False, w.shape, 1, 1, 0, 1, 1))
        gx, gw = torch.autograd.grad(y.sum(), [x, w], allow_unused=True)
        assert gw is None and gx is not None
    assert not conv2d_gradfix.weight_gradients_disabled


def test_fma_backward_unbroadcasts():
    a = torch.randn(2, 3, 4, 4, dtype=torch.float64, requires_grad=True)
    b = torch.randn(2, 3, 1, 1, dtype=torch.float64, requires_grad=True)
    c = torch.randn(4, 4, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(fma.fma, (a, b, c))


def test_filtered_lrelu_cpu_path_matches_reference():
    from pix2pix3d_amd.torch_utils.ops import filtered_lrelu
    g = load_golden('ops_filtered_lrelu')
    for i in range(int(g['num'])):
        up, down, flip = g[f'{i}.cfg'].tolist()
        fu, fd = g[f'{i}.fu'], g[f'{i}.fd']
        x = torch.tensor(g[f'{i}.x'], requires_grad=True)
        clamp = float(g[f'{i}.clamp'])
        y = filtered_lrelu.filtered_lrelu(x, fu=None if fu.size == 0 else torch.tensor(fu), fd=None if fd.size == 0 else torch.tensor(fd), b=torch.tensor(g[f'{i}.b']),
                                          up=up, down=down, padding=g[f'{i}.pad'].tolist(), gain=1.3, slope=0.15, clamp=None if clamp < 0 else clamp, flip_filter=bool(flip))
        assert rel_err(y.detach().numpy(), g[f'{i}.y']) < 1e-6, i
        gx, = torch.autograd.grad(y, x, torch.tensor(g[f'{i}.gy']))
        assert rel_err(gx.numpy(), g[f'{i}.gx']) < 1e-6, i


def test_dual_discriminator_forward_and_r1_on_cpu():
    """D forward (north-star API) and the R1 double-backward path through conv2d_gradfix.no_weight_gradients."""
    from pix2pix3d_amd.training.dual_discriminator import DualDiscriminator, SingleDiscriminator, filtered_resizing
    torch.manual_seed(0)
    D = DualDiscriminator(c_dim=25, img_resolution=32, img_channels=3, channel_base=512, channel_max=32, num_fp16_res=0, conv_clamp=None,
                          block_kwargs=dict(freeze_layers=0), mapping_kwargs=dict(), epilogue_kwargs=dict(mbstd_group_size=2))
    names = {n for n, _ in D.named_parameters()}
    assert {'b32.fromrgb.weight', 'b8.skip.weight', 'b4.out.bias', 'mapping.fc7.weight'} <= names and 'resample_filter' in dict(D.named_buffers())
    img = {'image': torch.randn(2, 3, 32, 32, requires_grad=True), 'image_raw': torch.randn(2, 3, 8, 8, requires_grad=True)}
    c = torch.randn(2, 25)
    logits = D(img, c)
    assert logits.shape == (2, 1) and logits.dtype == torch.float32
    with conv2d_gradfix.no_weight_gradients():
        g_img, g_raw = torch.autograd.grad(logits.sum(), [img['image'], img['image_raw']], create_graph=True)
    r1 = g_img.square().sum([1, 2, 3]) + g_raw.square().sum([1, 2, 3])
    r1.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in D.named_parameters() if 'b32.conv0.weight' in n)
    assert filtered_resizing(img['image_raw'], 32, D.resample_filter, 'classic').shape == (2, 3, 32, 32)
    S = SingleDiscriminator(c_dim=0, img_resolution=16, img_channels=3, channel_base=256, channel_max=16, num_fp16_res=0, conv_clamp=None)
    assert S({'image': torch.randn(2, 3, 16, 16)}, None).shape == (2, 1)
