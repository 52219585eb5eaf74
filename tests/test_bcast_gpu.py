"""csrc/bcast_ops.hip behind ``x * styles``, ``fma.fma`` and the bias-gradient sums (torch_utils/ops/bcast.py): values and gradients
against the tensor-op formulation in fp64 on the same (dtype-rounded) operands, both dense layouts, both dtypes.
Tolerances: one rounding of the result (fp16 1e-3 of the range, fp32 1e-6); reductions accumulate in fp32 (2e-3 / 2e-5)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 16, 24), (3, 96, 7, 8), (4, 8, 32, 32), (1, 512, 4, 4), (2, 128, 33, 16)]


def _fmt(layout):
    return torch.channels_last if layout == 'nhwc' else torch.contiguous_format


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('shape', SHAPES, ids=[str(s) for s in SHAPES])
def test_scale_channels_and_its_gradients(hip_lib, shape, layout, dtype):
    from pix2pix3d_amd.torch_utils.ops import bcast
    n, c, h, w = shape
    g = torch.Generator().manual_seed(n * c + h)
    x = torch.randn(shape, generator=g).to(dtype)
    s = (torch.randn(n, c, generator=g) + 1)
    gy = torch.randn(shape, generator=g).to(dtype)
    xd = x.cuda().contiguous(memory_format=_fmt(layout)).requires_grad_(True)
    sd = s.cuda().requires_grad_(True)
    if bcast.layout(xd) is None:
        pytest.skip('inner extent not a whole vector: the tensor-op route takes it')
    c0 = dict(bcast.calls)
    y = bcast.scale_channels(xd, sd)
    gx, gs = torch.autograd.grad(y, [xd, sd], gy.cuda().contiguous(memory_format=_fmt(layout)))
    assert bcast.calls['fma'] == c0['fma'] + 2 and bcast.calls['dot'] == c0['dot'] + 1
    assert y.stride() == xd.stride() and gx.stride() == xd.stride()
    xr, sr = x.double().requires_grad_(True), s.to(dtype).double().requires_grad_(True)
    yr = xr * sr.reshape(n, c, 1, 1)
    gxr, gsr = torch.autograd.grad(yr, [xr, sr], gy.double())
    tol, rtol = (1e-3, 2e-3) if dtype == torch.float16 else (1e-6, 2e-5)
    assert rel_err(y.detach().double().cpu(), yr.detach()) < tol
    assert rel_err(gx.double().cpu(), gxr) < tol and rel_err(gs.double().cpu(), gsr) < rtol


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('shared_noise', [True, False])
@pytest.mark.parametrize('shape', SHAPES[:3] + SHAPES[4:], ids=[str(s) for s in SHAPES[:3] + SHAPES[4:]])
def test_fma_with_noise_and_its_gradients(hip_lib, shape, shared_noise, layout, dtype):
    from pix2pix3d_amd.torch_utils.ops import bcast, fma
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + w)
    a = torch.randn(shape, generator=g).to(dtype)
    b = (torch.rand(n, c, 1, 1, generator=g) + 0.5).to(dtype)
    z = torch.randn(1 if shared_noise else n, 1, h, w, generator=g).to(dtype)
    gy = torch.randn(shape, generator=g).to(dtype)
    ad = a.cuda().contiguous(memory_format=_fmt(layout)).requires_grad_(True)
    bd, zd = b.cuda().requires_grad_(True), z.cuda().requires_grad_(True)
    if bcast.layout(ad) is None:
        pytest.skip('inner extent not a whole vector: the tensor-op route takes it')
    c0 = dict(bcast.calls)
    y = fma.fma(ad, bd, zd)
    assert bcast.calls['fma'] == c0['fma'] + 1                                     # fma.fma took the native route
    ga, gb, gz = torch.autograd.grad(y, [ad, bd, zd], gy.cuda().contiguous(memory_format=_fmt(layout)))
    ar, br, zr = (t.double().requires_grad_(True) for t in (a, b, z))
    yr = ar * br + zr
    gar, gbr, gzr = torch.autograd.grad(yr, [ar, br, zr], gy.double())
    tol, rtol = (1e-3, 2e-3) if dtype == torch.float16 else (1e-6, 2e-5)
    assert rel_err(y.detach().double().cpu(), yr.detach()) < tol and rel_err(ga.double().cpu(), gar) < tol
    assert gb.shape == bd.shape and gz.shape == zd.shape
    assert rel_err(gb.double().cpu(), gbr) < rtol and rel_err(gz.double().cpu(), gzr) < rtol


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_bias_gradient_sum_and_double_backward(hip_lib, layout, dtype):
    """bias_act's db through the native reduction, and its derivative (R1 differentiates through it)."""
    from pix2pix3d_amd.torch_utils.ops import bcast, bias_act
    torch.manual_seed(0)
    x = torch.randn(3, 64, 20, 12).to(dtype)
    b = torch.randn(64).to(dtype)
    xd = x.cuda().contiguous(memory_format=_fmt(layout)).requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    c0 = dict(bcast.calls)
    y = bias_act.bias_act(xd, bd, act='lrelu')
    gy = torch.randn_like(y)
    gx, gb = torch.autograd.grad(y, [xd, bd], gy, create_graph=True)
    assert bcast.calls['dot'] == c0['dot'] + 1
    xr, br = x.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(xr + br.reshape(1, -1, 1, 1), 0.2) * 2 ** 0.5
    gxr, gbr = torch.autograd.grad(yr, [xr, br], gy.double().cpu())
    assert rel_err(gb.detach().double().cpu(), gbr) < (2e-3 if dtype == torch.float16 else 2e-5)
    # derivative of sum(gb * v) w.r.t. the upstream gradient: v broadcast over the pixels, through the activation's slope
    gyv = gy.clone().requires_grad_(True)
    gx2, gb2 = torch.autograd.grad(y, [xd, bd], gyv, create_graph=True)
    v = torch.randn_like(gb2)
    d_gy, = torch.autograd.grad((gb2 * v).sum(), [gyv])
    slope = torch.where(xr + br.reshape(1, -1, 1, 1) > 0, 1.0, 0.2) * 2 ** 0.5
    ref = slope * v.double().cpu().reshape(1, -1, 1, 1)
    assert rel_err(d_gy.double().cpu(), ref.detach()) < (2e-3 if dtype == torch.float16 else 1e-5)
