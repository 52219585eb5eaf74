"""Parity of the HIP operators (through the C ABI) with the oracle and with vectors recorded from the
reference.  Tolerances: fp64 2e-7 (scalar parameters are C floats, as in the reference plugin), fp32 2e-5 relative-to-max (the kernels use the hardware exp/log
approximations, as the reference's --use_fast_math build does), fp16 3e-3."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ops_oracle as O

pytestmark = pytest.mark.gpu

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
TOL = {torch.float64: 2e-7, torch.float32: 2e-5, torch.float16: 3e-3}   # fp64: alpha/gain/clamp cross the ABI as C floats (bias_act.h:24-26)


def _opt(v):
    v = float(v)
    return None if v < 0 else v


def _ops():
    from pix2pix3d_amd.torch_utils.ops import bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix
    return bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix


@pytest.mark.parametrize('act', ACTS)
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32, torch.float16])
def test_bias_act_forward_and_gradients(hip_lib, act, dtype):
    bias_act = _ops()[0]
    from pix2pix3d_amd import _lib
    g = load_golden('ops_bias_act')
    kw = dict(dim=1, act=act, alpha=_opt(g[f'{act}.alpha']), gain=_opt(g[f'{act}.gain']), clamp=_opt(g[f'{act}.clamp']))
    x = torch.tensor(g[f'{act}.x'], dtype=dtype, device='cuda', requires_grad=True)
    b = torch.tensor(g[f'{act}.b'], dtype=dtype, device='cuda', requires_grad=True)
    dy = torch.tensor(g[f'{act}.dy'], dtype=dtype, device='cuda')
    ddx = torch.tensor(g[f'{act}.ddx'], dtype=dtype, device='cuda')
    n0 = _lib.launch_count('bias_act')
    y = bias_act.bias_act(x, b, **kw)
    assert _lib.launch_count('bias_act') > n0, 'the HIP kernel did not run'
    xs, bs = x.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert rel_err(y.detach().cpu().numpy(), O.bias_act(xs, bs, **kw)) < TOL[dtype]
    if dtype == torch.float64:                      # and against the reference record itself
        assert rel_err(y.detach().cpu().numpy(), g[f'{act}.y']) < 2e-7
    dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
    dys = dy.cpu().double().numpy()
    dxo, dbo = O.bias_act_grads(xs, bs, dys, **kw)
    near_kink = None
    if dtype == torch.float16:                      # fp16 rounding can put y on the other side of the clamp edge
        yo = O.bias_act(xs, bs, **kw)
        c = kw['clamp']
        near_kink = (np.abs(np.abs(yo) - c) < 5e-3 * max(c, 1)) if c is not None else np.zeros_like(yo, bool)
        near_kink |= np.abs(xs + bs.reshape(1, -1, 1, 1)) < 5e-3
    d = np.abs(dx.detach().cpu().double().numpy() - dxo)
    if near_kink is not None:
        d[near_kink] = 0
    assert d.max() / max(np.abs(dxo).max(), 1e-30) < TOL[dtype] * 2
    if dtype != torch.float16:
        assert rel_err(db.detach().cpu().numpy(), dbo) < TOL[dtype] * 4
    if bias_act.activation_funcs[act].has_2nd_grad and dtype != torch.float16:
        d2 = torch.autograd.grad(dx, x, ddx)[0]
        d2o = O.bias_act_second(xs, bs, dys, ddx.cpu().double().numpy(), **kw)
        assert np.abs(d2.cpu().double().numpy() - d2o).max() < TOL[dtype] * 10 * max(np.abs(d2o).max(), 1.0)


@pytest.mark.parametrize('shape,dim', [((3, 7), 1), ((1, 5, 33, 17), 1), ((2, 16, 8, 8), 1), ((5,), 0), ((0, 4), 1)])
def test_bias_act_shapes_layouts_and_unaligned_views(hip_lib, shape, dim):
    bias_act = _ops()[0]
    torch.manual_seed(3)
    x = torch.randn(shape, device='cuda')
    b = torch.randn(shape[dim], device='cuda')
    y = bias_act.bias_act(x, b, dim=dim, act='lrelu', clamp=0.9)
    assert y.shape == x.shape
    if x.numel():
        assert rel_err(y.cpu().numpy(), O.bias_act(x.cpu().numpy(), b.cpu().numpy(), dim=dim, act='lrelu', clamp=0.9)) < 2e-6
    if x.ndim == 4 and x.numel():
        xc = x.to(memory_format=torch.channels_last)
        yc = bias_act.bias_act(xc, b, dim=1, act='lrelu', clamp=0.9)
        assert yc.is_contiguous(memory_format=torch.channels_last) and torch.equal(yc, y)
        flat = torch.randn(x.numel() + 3, device='cuda')           # storage offset 1 element: 4-byte aligned only
        xv = flat[1:1 + x.numel()].view(shape)
        yv = bias_act.bias_act(xv, b, dim=1, act='lrelu', clamp=0.9)
        assert torch.equal(yv, bias_act.bias_act(xv.clone(), b, dim=1, act='lrelu', clamp=0.9))


def test_bias_act_large_streaming_matches_torch(hip_lib):
    bias_act = _ops()[0]
    x = torch.randn(4, 128, 256, 256, device='cuda', dtype=torch.float16)   # one SR-sized activation
    b = torch.randn(128, device='cuda', dtype=torch.float16)
    y = bias_act.bias_act(x, b, act='lrelu', gain=2 ** 0.5, clamp=256)
    ref = torch.nn.functional.leaky_relu(x.float() + b.float().view(1, -1, 1, 1), 0.2).mul(2 ** 0.5).clamp(-256, 256)
    assert (y.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.float64])
def test_upfirdn2d_cases(hip_lib, dtype):
    upfirdn2d = _ops()[1]
    g = load_golden('ops_upfirdn2d')
    for i in range(int(g['num_cases'])):
        f = g[f'{i}.f']
        ft = None if f.size == 0 else torch.tensor(f, device='cuda')
        x = torch.tensor(g[f'{i}.x'], device='cuda', dtype=dtype)
        kw = dict(up=g[f'{i}.up'].tolist(), down=g[f'{i}.down'].tolist(), padding=g[f'{i}.pad'].tolist(),
                  flip_filter=bool(g[f'{i}.flip']), gain=float(g[f'{i}.gain']))
        y = upfirdn2d.upfirdn2d(x, ft, **kw)
        yo = O.upfirdn2d(x.cpu().double().numpy(), None if f.size == 0 else f, **kw)
        assert tuple(y.shape) == yo.shape, i
        assert rel_err(y.cpu().numpy(), yo) < {torch.float32: 2e-6, torch.float16: 2e-3, torch.float64: 1e-7}[dtype], i
        if dtype == torch.float32:
            assert rel_err(y.cpu().numpy(), g[f'{i}.y']) < 3e-6, i
        yc = upfirdn2d.upfirdn2d(x.to(memory_format=torch.channels_last), ft, **kw)      # channels_last in -> out
        assert yc.is_contiguous(memory_format=torch.channels_last) or yc.shape[1] == 1
        assert torch.allclose(yc.float(), y.float(), atol=1e-6 if dtype != torch.float16 else 2e-3)


def test_upfirdn2d_gradient_is_the_adjoint(hip_lib):
    upfirdn2d = _ops()[1]
    torch.manual_seed(5)
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    for kw in [dict(up=2, padding=[2, 1, 2, 1], gain=4), dict(down=2, padding=[1, 1, 1, 1]), dict(padding=[1, 1, 1, 1], gain=4),
               dict(up=2, down=2, padding=[3, 0, 1, 2], flip_filter=True)]:
        x = torch.randn(2, 3, 10, 9, device='cuda', dtype=torch.float64, requires_grad=True)
        y = upfirdn2d.upfirdn2d(x, f, **kw)
        gy = torch.randn_like(y)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        xr = x.detach().cpu().requires_grad_(True)
        yr = upfirdn2d.upfirdn2d(xr, f.cpu(), impl='ref', **kw)
        gxr, = torch.autograd.grad(yr, xr, gy.cpu())
        assert torch.allclose(gx.detach().cpu(), gxr, atol=1e-9)
        # <A x, gy> == <x, A^T gy>
        assert abs((y.detach() * gy).sum().item() - (x.detach() * gx.detach()).sum().item()) < 1e-8 * y.numel()


def test_conv2d_resample_and_gradfix_on_gpu(hip_lib):
    _, upfirdn2d, conv2d_resample, conv2d_gradfix = _ops()
    g = load_golden('ops_conv')
    f = torch.tensor(g['f'], device='cuda')
    prev = conv2d_gradfix.enabled
    conv2d_gradfix.enabled = True
    try:
        for i in range(int(g['num_resample'])):
            k, up, down, flipw = g[f'r{i}.cfg'].tolist()
            x = torch.tensor(g[f'r{i}.x'], device='cuda', requires_grad=True)
            w = torch.tensor(g[f'r{i}.w'], device='cuda', requires_grad=True)
            y = conv2d_resample.conv2d_resample(x, w, f=f, up=up, down=down, padding=g[f'r{i}.pad'].tolist(), flip_weight=bool(flipw))
            assert rel_err(y.detach().cpu().numpy(), g[f'r{i}.y']) < 2e-5, i
            gx, gw = torch.autograd.grad(y.square().sum(), [x, w], create_graph=True)
            (gx.square().sum() + gw.square().sum()).backward()          # second order runs end to end
            assert torch.isfinite(x.grad).all() and torch.isfinite(w.grad).all()
    finally:
        conv2d_gradfix.enabled = prev


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_filtered_lrelu_on_gpu(hip_lib, dtype):
    from pix2pix3d_amd.torch_utils.ops import filtered_lrelu
    from pix2pix3d_amd import _lib
    g = load_golden('ops_filtered_lrelu')
    for i in range(int(g['num'])):
        up, down, flip = g[f'{i}.cfg'].tolist()
        fu, fd = g[f'{i}.fu'], g[f'{i}.fd']
        x = torch.tensor(g[f'{i}.x'], device='cuda', dtype=dtype, requires_grad=True)
        clamp = float(g[f'{i}.clamp'])
        n0 = _lib.launch_count('upfirdn2d') + _lib.launch_count('bias_act') + _lib.launch_count('filtered_lrelu')
        y = filtered_lrelu.filtered_lrelu(x, fu=None if fu.size == 0 else torch.tensor(fu, device='cuda'), fd=None if fd.size == 0 else torch.tensor(fd, device='cuda'),
                                          b=torch.tensor(g[f'{i}.b'], device='cuda', dtype=dtype), up=up, down=down, padding=g[f'{i}.pad'].tolist(),
                                          gain=1.3, slope=0.15, clamp=None if clamp < 0 else clamp, flip_filter=bool(flip))
        assert _lib.launch_count('upfirdn2d') + _lib.launch_count('bias_act') + _lib.launch_count('filtered_lrelu') > n0
        tol = 2e-5 if dtype == torch.float32 else 2e-7
        assert rel_err(y.detach().cpu().numpy(), g[f'{i}.y']) < tol, i
        gx, = torch.autograd.grad(y, x, torch.tensor(g[f'{i}.gy'], device='cuda', dtype=dtype))
        assert rel_err(gx.cpu().numpy(), g[f'{i}.gx']) < tol * 5, i
