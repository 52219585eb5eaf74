"""filtered_lrelu on the device: the fused kernel (p3d_filtered_lrelu) and the generic route (p3d_filtered_lrelu_act) against the
reference's records (tests/golden/ops_filtered_lrelu.npz: outputs and input gradients of its ``_filtered_lrelu_ref``) and against the
numpy oracle, INCLUDING the bit-packed sign tensor: byte-for-byte equal to the oracle's packing of the same signs wherever the
up-sampled value is not within rounding distance of zero or of the clamp (index / mask class: exact)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ops_oracle as O

pytestmark = pytest.mark.gpu


def _case(g, i):
    up, down, flip = g[f'{i}.cfg'].tolist()
    fu, fd = g[f'{i}.fu'], g[f'{i}.fd']
    clamp = float(g[f'{i}.clamp'])
    return dict(up=up, down=down, flip_filter=bool(flip), fu=None if fu.size == 0 else fu, fd=None if fd.size == 0 else fd,
                clamp=None if clamp < 0 else clamp, padding=g[f'{i}.pad'].tolist(), gain=1.3, slope=0.15)


def _t(a, dtype=torch.float32):
    return None if a is None else torch.tensor(a, device='cuda', dtype=dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('route', ['fused', 'generic'])
def test_forward_backward_match_reference_records(hip_lib, dtype, route, monkeypatch):
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import filtered_lrelu as F
    if route == 'generic':                                       # what the op does when the plugin reports "no specialised kernel"
        monkeypatch.setattr(F._Plugin, 'filtered_lrelu', staticmethod(lambda x, *a: (torch.empty([0], device=x.device), torch.empty([0], device=x.device), -1)))
        F._op_cache.clear()
    g = load_golden('ops_filtered_lrelu')
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    for i in range(int(g['num'])):
        cs = _case(g, i)
        x = torch.tensor(g[f'{i}.x'], device='cuda', dtype=dtype, requires_grad=True)
        n0 = _lib.launch_count('filtered_lrelu')
        y = F.filtered_lrelu(x, fu=_t(cs['fu']), fd=_t(cs['fd']), b=_t(g[f'{i}.b'], dtype), up=cs['up'], down=cs['down'], padding=cs['padding'],
                             gain=cs['gain'], slope=cs['slope'], clamp=cs['clamp'], flip_filter=cs['flip_filter'])
        assert _lib.launch_count('filtered_lrelu') > n0
        assert rel_err(y.detach().float().cpu().numpy(), g[f'{i}.y']) < tol, (i, route)
        gx, = torch.autograd.grad(y, x, _t(g[f'{i}.gy'], dtype))
        if dtype == torch.float16 and route == 'generic':
            # the generic route rounds the up-sampled tensor to fp16 BEFORE the sign / clamp decision: an element within fp16 rounding of the
            # clamp can land on the other side than in the fp64 record and then owns a whole input gradient tap — bound the bulk instead
            d = np.abs(gx.float().cpu().numpy() - g[f'{i}.gx']) / np.abs(g[f'{i}.gx']).max()
            assert np.mean(d > tol * 5) < 0.1 and np.median(d) < tol, (i, route, np.mean(d > tol * 5), np.median(d))
        else:
            assert rel_err(gx.float().cpu().numpy(), g[f'{i}.gx']) < tol * 5, (i, route)
    F._op_cache.clear()


@pytest.mark.parametrize('cfg', [dict(up=2, down=2, fu=12, fd=12, pad=[9, 10, 9, 10], clamp=0.9, flip=False, shape=(2, 5, 37, 41)),
                                 dict(up=4, down=2, fu=16, fd=8, pad=[13, 12, 13, 12], clamp=None, flip=True, shape=(1, 3, 20, 23)),
                                 dict(up=1, down=1, fu=1, fd=1, pad=[0, 0, 0, 0], clamp=0.5, flip=False, shape=(2, 4, 33, 70)),
                                 dict(up=2, down=4, fu=8, fd=16, pad=[10, 11, 10, 11], clamp=None, flip=False, shape=(1, 2, 40, 40))],
                         ids=['2x2', '4x2', '1x1', '2x4'])
def test_sign_tensor_is_bit_exact_and_round_trips(hip_lib, cfg):
    """Plugin protocol directly: (y, so, 0) in write mode; ``so`` equals the oracle's packed signs byte for byte on every byte none of
    whose four elements is a near-tie; feeding ``so`` back in read mode (the backward configuration) reproduces the oracle's gradient."""
    from pix2pix3d_amd.torch_utils.ops import filtered_lrelu as F
    from pix2pix3d_amd.torch_utils.ops import upfirdn2d
    rng = np.random.RandomState(7)
    n, c, h, w = cfg['shape']
    x = rng.randn(n, c, h, w).astype(np.float32)
    b = rng.randn(c).astype(np.float32)
    fu = upfirdn2d.setup_filter(rng.randn(cfg['fu']).tolist()).numpy() if cfg['fu'] > 1 else None
    fd = upfirdn2d.setup_filter(rng.randn(cfg['fd']).tolist()).numpy() if cfg['fd'] > 1 else None
    px0, px1, py0, py1 = cfg['pad']
    yo, so_o = O.filtered_lrelu(x, fu, fd, b, cfg['up'], cfg['down'], cfg['pad'], 1.3, 0.2, cfg['clamp'], cfg['flip'], write_signs=True)
    clamp = float('inf') if cfg['clamp'] is None else cfg['clamp']
    y, so, rc = F._Plugin.filtered_lrelu(_t(x), _t(fu), _t(fd), _t(b), None, cfg['up'], cfg['down'], px0, px1, py0, py1, 0, 0, 1.3, 0.2, clamp, cfg['flip'], True)
    assert rc == 0 and so.dtype == torch.uint8 and tuple(so.shape) == so_o.shape
    assert rel_err(y.cpu().numpy(), yo) < 2e-5
    # near-ties: recompute the up-sampled values in float64 and mask the bytes that hold one
    sz = O.filtered_lrelu_sizes(x.shape, fu, fd, cfg['up'], cfg['down'], cfg['pad'])
    ext_w = (sz['sw_active'] + 3) & ~3
    u = O.upfirdn2d(x.astype(np.float64) + b.reshape(1, -1, 1, 1), O._filter2d(fu), up=cfg['up'],
                    padding=[px0, px1 + max(ext_w - sz['cw'], 0), py0, py1 + max(sz['sh'] - sz['ch'], 0)], flip_filter=cfg['flip']) * (cfg['up'] ** 2 * 1.3)
    u = u[..., :sz['sh'], :ext_w]
    scale = np.abs(u).max()
    tie = np.abs(u) < 1e-5 * scale
    if cfg['clamp'] is not None:
        act = np.where(u < 0, u * 0.2, u)
        tie |= np.abs(np.abs(act) - cfg['clamp']) < 1e-5 * scale
    tie_bytes = np.zeros(so_o.shape, bool)
    tie4 = tie.reshape(tie.shape[:-1] + (ext_w // 4, 4)).any(-1)
    tie_bytes[..., :tie4.shape[-1]] = tie4
    got = so.cpu().numpy()
    valid = np.zeros(so_o.shape, bool)
    valid[..., :(sz['sw_active'] + 3) >> 2] = True                   # bytes beyond the active width are padding
    cmp = valid & ~tie_bytes
    assert cmp.mean() > 0.5 and np.array_equal(got[cmp], so_o[cmp]), (np.mean(got[cmp] != so_o[cmp]), cmp.mean())
    # backward configuration: read the signs the kernel itself wrote
    gy = rng.randn(*yo.shape).astype(np.float32)
    gxo = O.filtered_lrelu_backward(gy, fu, fd, x.shape, got, cfg['up'], cfg['down'], cfg['pad'], 1.3, 0.2, cfg['flip'])
    fuw = 1 if fu is None else fu.shape[-1]
    fdw = 1 if fd is None else fd.shape[-1]
    pp = [(fuw - 1) + (fdw - 1) - px0, w * cfg['up'] - yo.shape[3] * cfg['down'] + px0 - (cfg['up'] - 1),
          (fuw - 1) + (fdw - 1) - py0, h * cfg['up'] - yo.shape[2] * cfg['down'] + py0 - (cfg['up'] - 1)]
    gx, so2, rc2 = F._Plugin.filtered_lrelu(_t(gy), _t(fd), _t(fu), None, so, cfg['down'], cfg['up'], pp[0], pp[1], pp[2], pp[3], -(fuw - 1) + px0, -(fuw - 1) + py0,
                                            1.3 * cfg['up'] ** 2 / cfg['down'] ** 2, 0.2, float('inf'), not cfg['flip'], False)
    assert rc2 == 0 and so2.numel() == 0 and tuple(gx.shape) == x.shape
    assert rel_err(gx.cpu().numpy(), gxo) < 2e-5


def test_act_kernel_and_unsupported_geometry(hip_lib):
    """filtered_lrelu_act_: in place, sign tensor [N,C,H,ceil16(W)/4]; read mode with offsets; and the -1 protocol for a geometry
    whose tiles do not fit LDS."""
    from pix2pix3d_amd.torch_utils.ops import filtered_lrelu as F
    torch.manual_seed(0)
    for dtype in (torch.float32, torch.float16, torch.float64):
        x = torch.randn(2, 3, 9, 21, device='cuda', dtype=dtype)
        x0 = x.clone()
        so = F._Plugin.filtered_lrelu_act_(x, None, 0, 0, 1.5, 0.1, 1.0, True)
        v = x0.double() * 1.5
        code = (v < 0).to(torch.uint8)
        act = torch.where(v < 0, v * 0.1, v)
        code = torch.where(act.abs() > 1.0, torch.full_like(code, 2), code)
        assert torch.allclose(x.double(), act.clamp(-1, 1), atol=2e-3 if dtype == torch.float16 else 1e-6)
        assert tuple(so.shape) == (2, 3, 9, 8) and np.array_equal(so.cpu().numpy(), O.pack_signs(code.cpu().numpy()))
        # read mode at an offset: gradient-like tensor through the saved signs
        gsrc = torch.randn(2, 3, 7, 18, device='cuda', dtype=dtype)
        gref = gsrc.double() * 0.7
        cd = torch.tensor(O.unpack_signs(so.cpu().numpy(), *np.meshgrid(np.arange(18) + 2, np.arange(7) + 1, indexing='xy')), device='cuda')
        gref = torch.where((cd & 1) > 0, gref * 0.1, gref)
        gref = torch.where((cd & 2) > 0, torch.zeros_like(gref), gref)
        ret = F._Plugin.filtered_lrelu_act_(gsrc, so, 2, 1, 0.7, 0.1, float('inf'), False)
        assert ret.numel() == 0 and torch.allclose(gsrc.double(), gref, atol=2e-3 if dtype == torch.float16 else 1e-6)
    # 64-tap filters at up = down = 4 need 103 KB of tiles: above the 64 KB default, inside gfx950's 160 KB (opt-in taken by the entry point)
    xb = torch.randn(1, 2, 32, 32, device='cuda')
    f64 = torch.randn(64, device='cuda').abs() + 0.1
    f64 = f64 / f64.sum()
    y, so, rc = F._Plugin.filtered_lrelu(xb, f64, f64, None, None, 4, 4, 30, 30, 30, 30, 0, 0, 1.0, 0.2, float('inf'), False, False)
    assert rc == 0 and y.numel() > 0
    want = F.filtered_lrelu(xb, fu=f64, fd=f64, up=4, down=4, padding=30, gain=1.0, slope=0.2, clamp=None, impl='ref')
    assert y.shape == want.shape and (y - want).abs().max().item() < 2e-5 * max(want.abs().max().item(), 1e-3)
    # ... and 96 taps (187 KB) do not fit: the plugin's -1 protocol
    big = torch.ones(96, device='cuda')
    y, so, rc = F._Plugin.filtered_lrelu(torch.randn(1, 1, 32, 32, device='cuda'), big, big, None, None, 4, 4, 46, 46, 46, 46, 0, 0, 1.0, 0.2, float('inf'), False, False)
    assert rc == -1 and y.numel() == 0 and so.numel() == 0
