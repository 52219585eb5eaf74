"""conv2d_gradfix on the device: forward, data gradient, weight gradient and the second-order terms of every geometry
conv2d_resample asks for, through libp3d_hip.so (p3d_conv2d_forward / p3d_conv2d_bwd_weight), against torch's own operators
evaluated on the CPU in float64 on the same values (the fp16 cases on the fp16-rounded inputs).

Tolerances (max |a - b| / max |b|): fp32 2e-5 (f32-input MFMA is an exact fma chain: summation order only), fp16 4e-3 (one
rounding of the result to fp16; accumulation is fp32)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.float16: 4e-3}      # fp32: forward / data gradient run as bf16x3 by default (measured <= 6e-6), the weight gradient exact

# (name, transposed, k, stride, Ci, Co, H, W, N, output_padding)
GEOM = [
    ('same3x3', False, 3, 1, 64, 96, 20, 18, 2, 0),
    ('same3x3_big', False, 3, 1, 128, 128, 40, 40, 2, 0),            # fp16: the two-blocks-per-CU halo kernel
    ('same1x1', False, 1, 1, 64, 160, 12, 20, 3, 0),
    ('down3x3_odd', False, 3, 2, 64, 64, 33, 33, 2, 0),
    ('down3x3_even', False, 3, 2, 64, 32, 34, 36, 2, 0),             # its data gradient needs output_padding = 1
    ('up3x3', True, 3, 2, 64, 128, 17, 19, 2, 0),
    ('up3x3_big', True, 3, 2, 64, 128, 32, 32, 2, 0),                # fp16: convT_h2 kernel
    ('up3x3_big_op1', True, 3, 2, 64, 128, 32, 32, 1, 1),
    ('sameT3x3', True, 3, 1, 64, 64, 16, 16, 2, 0),                  # the data gradient of 'same3x3' as a forward op
    ('fold4x4', False, 3, 1, 64, 64, 4, 4, 4, 0),                    # batch folded into the GEMM rows
    ('fold8x8', False, 3, 1, 128, 64, 8, 8, 5, 0),
    ('torgb', False, 1, 1, 128, 3, 24, 24, 2, 0),                    # skinny 1x1, many -> few
    ('fromrgb', False, 1, 1, 6, 64, 24, 24, 2, 0),                   # skinny 1x1, few -> many
    ('odd_channels', False, 3, 1, 33, 40, 4, 4, 2, 0),               # channel padding route (the 513-channel epilogue conv of D)
    ('small64_mid', False, 3, 1, 64, 64, 40, 44, 3, 0),              # weight gradient: the 64-channel form over several double chunks, an odd count per split, a half-filled last one
    ('half_tile_b', False, 3, 1, 128, 64, 40, 44, 3, 0),             # weight gradient: waves pair up on pixels when one side holds <= 64 channels ...
    ('half_tile_s', False, 3, 1, 64, 128, 40, 44, 3, 0),             # ... either side
    ('row64', False, 3, 1, 128, 128, 64, 64, 2, 0),                  # fp16 weight gradient: transposing LDS reads, rows of 64 pixels (one position per thread)
    ('row64_down', False, 3, 2, 64, 128, 129, 129, 2, 0),            # ... against the big image at stride 2 (gradient image 64 x 64)
    ('row64_up', True, 3, 2, 128, 64, 64, 64, 2, 0),                 # ... and with the roles swapped (transposed op)
    ('small64_row128', False, 3, 1, 64, 64, 24, 128, 2, 0),          # ... the 64-channel form of it on rows of 128 pixels (double chunks, one position per thread)
    ('small64_narrow', False, 3, 1, 32, 64, 20, 36, 3, 0),           # ... and with a 32-channel side, rows that are no multiple of anything
]


def _make(geom, dtype, seed=0):
    name, tr, k, stride, ci, co, h, w, n, op = geom
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(*((ci, co) if tr else (co, ci)), k, k, generator=g) / np.sqrt(ci * k * k)
    x, wt = x.to(dtype).double(), wt.to(dtype).double()               # the values the device sees, exactly
    return x, wt, g


def _ref_op(x, wt, geom):
    name, tr, k, stride, ci, co, h, w, n, op = geom
    pad = k // 2 if stride == 1 else 0
    if tr:
        return F.conv_transpose2d(x, wt, stride=stride, padding=pad, output_padding=op)
    return F.conv2d(x, wt, stride=stride, padding=pad)


def _dev_op(x, wt, geom):
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    name, tr, k, stride, ci, co, h, w, n, op = geom
    pad = k // 2 if stride == 1 else 0
    if tr:
        return conv2d_gradfix.conv_transpose2d(x, wt, stride=stride, padding=pad, output_padding=op)
    return conv2d_gradfix.conv2d(x, wt, stride=stride, padding=pad)


@pytest.fixture()
def gradfix():
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    prev = conv2d_gradfix.enabled
    conv2d_gradfix.enabled = True
    yield conv2d_gradfix
    conv2d_gradfix.enabled = prev


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('geom', GEOM, ids=[g[0] for g in GEOM])
def test_forward_and_first_order_gradients(hip_lib, gradfix, geom, dtype, layout):
    x64, w64, g = _make(geom, dtype)
    xr, wr = x64.clone().requires_grad_(True), w64.clone().requires_grad_(True)
    yr = _ref_op(xr, wr, geom)
    gy64 = torch.randn(yr.shape, generator=g).to(dtype).double()
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy64)

    fmt = torch.channels_last if layout == 'nhwc' else torch.contiguous_format
    xd = x64.to('cuda', dtype).contiguous(memory_format=fmt).requires_grad_(True)
    wd = w64.to('cuda', dtype).requires_grad_(True)
    c0 = dict(gradfix.native_calls)
    yd = _dev_op(xd, wd, geom)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], gy64.to('cuda', dtype).contiguous(memory_format=fmt))
    torch.cuda.synchronize()
    assert gradfix.native_calls['forward'] == c0['forward'] + 2 and gradfix.native_calls['weight_grad'] == c0['weight_grad'] + 1, gradfix.native_calls
    assert gradfix.native_calls['aten'] == c0['aten']
    assert yd.shape == yr.shape and gxd.shape == xr.shape and gwd.shape == wr.shape
    tol = TOL[dtype]
    e = dict(y=rel_err(yd.detach().double().cpu(), yr.detach()), gx=rel_err(gxd.double().cpu(), gxr), gw=rel_err(gwd.double().cpu(), gwr))
    print(geom[0], dtype, layout, e)
    assert e['y'] < tol and e['gx'] < tol and e['gw'] < tol, e


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('geom', [GEOM[0], GEOM[3], GEOM[5], GEOM[11]], ids=['same3x3', 'down3x3', 'up3x3', 'torgb'])
def test_second_order_terms(hip_lib, gradfix, geom, dtype):
    """R1-style double backward: L2 = <d(sum(y * gy))/dx, v> differentiated w.r.t. w and gy — the weight gradient of the flipped op
    and the forward op again (conv2d_gradfix.py:139-194)."""
    x64, w64, g = _make(geom, dtype, seed=3)

    def run(x, w, gy, v, op):
        y = op(x, w, geom)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        return torch.autograd.grad((gx * v).sum(), [w, gy])

    xr, wr = x64.clone().requires_grad_(True), w64.clone().requires_grad_(True)
    shape = _ref_op(xr, wr, geom).shape
    gy64 = torch.randn(shape, generator=g).to(dtype).double()
    v64 = torch.randn(x64.shape, generator=g).to(dtype).double()
    gwr, ggr = run(xr, wr, gy64.clone().requires_grad_(True), v64, _ref_op)
    xd, wd = x64.to('cuda', dtype).requires_grad_(True), w64.to('cuda', dtype).requires_grad_(True)
    c0 = dict(gradfix.native_calls)
    gwd, ggd = run(xd, wd, gy64.to('cuda', dtype).requires_grad_(True), v64.to('cuda', dtype), _dev_op)
    torch.cuda.synchronize()
    assert gradfix.native_calls['aten'] == c0['aten'] and gradfix.native_calls['weight_grad'] >= c0['weight_grad'] + 1
    tol = TOL[dtype]
    e = dict(gw=rel_err(gwd.double().cpu(), gwr), ggy=rel_err(ggd.double().cpu(), ggr))
    print(geom[0], dtype, e)
    assert e['gw'] < tol and e['ggy'] < tol, e


def test_no_weight_gradients_and_bias(hip_lib, gradfix):
    x = torch.randn(2, 64, 16, 16, device='cuda', requires_grad=True)
    w = torch.randn(32, 64, 3, 3, device='cuda', requires_grad=True)
    b = torch.randn(32, device='cuda', requires_grad=True)
    y = gradfix.conv2d(x, w, b, padding=1)
    yr = F.conv2d(x.detach().cpu().double(), w.detach().cpu().double(), b.detach().cpu().double(), padding=1)
    assert rel_err(y.detach().double().cpu(), yr) < 2e-5
    with gradfix.no_weight_gradients():
        gx, gb = torch.autograd.grad(y.sum(), [x, b], retain_graph=True)
        with pytest.raises(RuntimeError):
            torch.autograd.grad(y.sum(), [w], retain_graph=True)      # no gradient flows to the weight in this mode
    assert rel_err(gb.cpu(), torch.full([32], 2.0 * 16 * 16)) < 1e-6


def test_weight_gradient_at_training_sizes(hip_lib, gradfix):
    """The split-K weight-gradient kernel at the pixel counts of the SR heads / discriminator (batch 4 x 256^2, fp16) against an
    fp32 einsum over the same fp16 values on the device, and determinism (no atomics: two runs are bit-identical)."""
    from pix2pix3d_amd.torch_utils.ops.conv2d_gradfix import _Cfg, _weight_grad_impl
    torch.manual_seed(0)
    n, ci, co, h = 4, 128, 128, 256
    x = torch.randn(n, ci, h, h, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(n, co, h, h, device='cuda', dtype=torch.float16) / 64).contiguous(memory_format=torch.channels_last)
    cfg = _Cfg(False, (co, ci, 3, 3), 1, 1, 0, 1, 1)
    gw = _weight_grad_impl(gy, x, cfg)
    gw2 = _weight_grad_impl(gy, x, cfg)
    assert torch.equal(gw, gw2)
    xp = F.pad(x.float(), (1, 1, 1, 1))
    ref = torch.stack([torch.stack([torch.einsum('nohw,nihw->oi', gy.float(), xp[:, :, ky:ky + h, kx:kx + h]) for kx in range(3)], -1) for ky in range(3)], -2)
    assert rel_err(gw.float().cpu(), ref.cpu()) < 4e-3


X6_GEOM = [g for g in GEOM if g[0] in ('same3x3', 'same3x3_big', 'same1x1', 'down3x3_even', 'up3x3', 'up3x3_big_op1', 'sameT3x3', 'fold8x8', 'odd_channels', 'row64', 'half_tile_b')] + [
    ('wide_down', False, 3, 2, 128, 256, 65, 65, 2, 0),              # weight gradient as bf16x6: whole 128 x 128 tiles at stride 2 ...
    ('wide_up', True, 3, 2, 256, 128, 32, 32, 2, 0),                 # ... with the roles swapped (transposed op) ...
    ('wide_1x1', False, 1, 1, 128, 160, 40, 36, 3, 0),               # ... one tap, a ragged tile on the big side
    ('wide_ragged', False, 3, 1, 132, 200, 21, 23, 3, 0),            # ... channel counts that are multiples of 4 only, a pixel count that is no multiple of the chunk
]


@pytest.fixture(params=['auto', 'presplit'])
def x6_kernel_choice(request):
    """'presplit': conv3x3_halo_x6p_kernel on every 3x3 'same' forward / data-gradient launch of the halo-slab route, whatever its grid (P3D_X6_PRESPLIT=2, read by the
    library at every call); 'auto': launches of at least 256 work-groups only."""
    prev = os.environ.get('P3D_X6_PRESPLIT')
    if request.param == 'presplit':
        os.environ['P3D_X6_PRESPLIT'] = '2'
    yield request.param
    if prev is None:
        os.environ.pop('P3D_X6_PRESPLIT', None)
    else:
        os.environ['P3D_X6_PRESPLIT'] = prev


@pytest.mark.parametrize('geom', X6_GEOM, ids=lambda g: g[0])
def test_forward_and_data_gradient_as_bf16x6(hip_lib, gradfix, x6_kernel_choice, geom):
    """modconv.f32_x6 (P3D_F32_BF16X6, the default): the fp32 forward and data-gradient convolutions — and the weight gradients of whole 128 x 128 tiles (both sides
    > 64 channels; modconv.wgrad_x6) — as six bf16 MFMAs per product of three-piece splits: the error class of the exact fp32 kernels (bar 4e-6 of the range
    against fp64, and within 2x of what the exact kernels measure on the same tensors)."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    x64, w64, g = _make(geom, torch.float32)
    xr, wr = x64.clone().requires_grad_(True), w64.clone().requires_grad_(True)
    yr = _ref_op(xr, wr, geom)
    gy64 = torch.randn(yr.shape, generator=g).float().double()
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy64)
    errs = {}
    prev = modconv.f32_x6
    try:
        for x6 in (True, False):
            modconv.f32_x6 = x6
            xd = x64.to('cuda', torch.float32).requires_grad_(True)
            wd = w64.to('cuda', torch.float32).requires_grad_(True)
            c0 = dict(gradfix.native_calls)
            yd = _dev_op(xd, wd, geom)
            gxd, gwd = torch.autograd.grad(yd, [xd, wd], gy64.to('cuda', torch.float32))
            torch.cuda.synchronize()
            assert gradfix.native_calls['forward'] == c0['forward'] + 2 and gradfix.native_calls['aten'] == c0['aten']
            assert gradfix.native_calls['weight_grad'] == c0['weight_grad'] + 1
            errs[x6] = dict(y=rel_err(yd.detach().double().cpu(), yr.detach()), gx=rel_err(gxd.double().cpu(), gxr), gw=rel_err(gwd.double().cpu(), gwr))
    finally:
        modconv.f32_x6 = prev
    print(geom[0], 'bf16x6', errs[True], 'exact', errs[False])
    for k in ('y', 'gx', 'gw'):
        assert errs[True][k] < 4e-6 and errs[True][k] < 2 * errs[False][k] + 1e-7, (k, errs)
    name, tr, k, stride, ci, co = geom[:6]
    if min(ci, co) > 64 and ci % 4 == 0 and co % 4 == 0:
        assert errs[True]['gw'] != errs[False]['gw']                # (the other arithmetic did run)
