"""``filtered_lrelu``: bias -> upsampling FIR -> leaky ReLU (gain, clamp) -> downsampling FIR (StyleGAN3).

Mirror of the reference operator (torch_utils/ops/filtered_lrelu.py:58-274) and of its plugin seam
(``filtered_lrelu_plugin.filtered_lrelu`` / ``filtered_lrelu_act_``, filtered_lrelu.cpp:20-23, :217).  No pix2pix3D configuration
calls this op (its only caller, networks_stylegan3.SynthesisLayer, is imported but never instantiated — SURVEY §2 note N1); it is
here because the operator API and its saved sign tensor are part of the drop-in boundary.

Device tensors run ONE fused kernel of libp3d_hip.so (``p3d_filtered_lrelu``, csrc/filtered_lrelu.hip) that also writes the
bit-packed sign tensor; the backward pass is the same kernel with up/down and the filters swapped, reading those signs instead of
comparing (filtered_lrelu.py:240-270) — so every gradient order is native and only 2 bits per up-sampled element are kept for
autograd.  Geometries whose tiles do not fit LDS come back with the plugin's "no specialised kernel" code (-1) and take the generic
route the reference takes in that case: ``upfirdn2d`` -> ``p3d_filtered_lrelu_act`` (in place, same sign tensor) -> ``upfirdn2d``.
CPU tensors use the plain four-step composition (:123-148).
"""
import ctypes

import numpy as np
import torch

from ... import _lib
from . import bias_act
from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_p32, _p64 = ctypes.POINTER(_i32), ctypes.POINTER(_i64)
_lib.register('p3d_filtered_lrelu', ctypes.c_int, [_vp] * 6 + [ctypes.c_int, _p32, _p64, _p32, _p64, _i64] + [_i32] * 13 + [_f32] * 3 + [_i32, _i32, _vp])
_lib.register('p3d_filtered_lrelu_act', ctypes.c_int, [_vp, _vp, ctypes.c_int, _p32, _p64] + [_i32] * 4 + [_f32] * 3 + [_i32, _vp])


def _ref_composition(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter, impl):
    px0, px1, py0, py1 = padding
    y = bias_act.bias_act(x=x, b=b, impl=impl)
    y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, impl=impl)
    y = bias_act.bias_act(x=y, act='lrelu', alpha=slope, gain=gain, clamp=clamp, impl=impl)
    return upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter, impl=impl)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl='cuda'):
    """x [N,C,H,W]; fu/fd FIR filters from ``upfirdn2d.setup_filter``; output size
    ``(in*up + pad0 + pad1 - (fu-1) - (fd-1) + (down-1)) // down`` per axis (filtered_lrelu.py:58-118)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ['ref', 'cuda']
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype and tuple(b.shape) == (x.shape[1],)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    px0, px1, py0, py1 = _parse_padding(padding)
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    n, c, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    if impl == 'cuda' and x.device.type == 'cuda' and x.dtype in (torch.float16, torch.float32):
        y = _native_op(up, down, (px0, px1, py0, py1), float(gain), float(slope), None if clamp is None else float(clamp), bool(flip_filter)).apply(x, fu, fd, b, None, 0, 0)
    else:
        y = _ref_composition(x, fu, fd, b, up, down, (px0, px1, py0, py1), gain, slope, clamp, flip_filter, impl if x.device.type == 'cuda' else 'ref')
    assert tuple(y.shape) == (n, c, out_h, out_w) and y.dtype == x.dtype
    return y


def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    return filtered_lrelu(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter, impl='ref')


# ---- plugin seam ---------------------------------------------------------------------------------------------------------------
def _table(f, device):
    """Dense fp32 [h, w] table of a filter as the kernel wants it (None -> {1}; separable -> outer product)."""
    if f is None:
        return torch.ones([1, 1], dtype=torch.float32, device=device)
    f = f.to(device=device, dtype=torch.float32)
    return (torch.outer(f, f) if f.ndim == 1 else f).contiguous()


def _sizes4(t):
    return _lib.i32x4(t.shape[3], t.shape[2], t.shape[1], t.shape[0]), _lib.i64x4(t.stride(3), t.stride(2), t.stride(1), t.stride(0))


class _Plugin:
    """``filtered_lrelu_plugin`` (filtered_lrelu.cpp:20-23, :217): same arguments, same returns — ``(y, so, return_code)`` with
    return_code -1 and empty tensors for "no specialised kernel" — backed by libp3d_hip.so."""

    @staticmethod
    def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filters, write_signs):
        assert x.is_cuda and x.ndim == 4 and x.dtype in (torch.float16, torch.float32)
        fu2, fd2 = _table(fu, x.device), _table(fd, x.device)
        n, c, xh, xw = x.shape
        cw, ch = xw * up + px0 + px1 - (fu2.shape[1] - 1), xh * up + py0 + py1 - (fu2.shape[0] - 1)
        if not (cw > fd2.shape[1] - 1 and ch > fd2.shape[0] - 1):
            raise RuntimeError('upsampled buffer must be at least the size of downsampling filter')
        yw, yh = (cw - (fd2.shape[1] - 1) + down - 1) // down, (ch - (fd2.shape[0] - 1) + down - 1) // down
        fmt = torch.channels_last if (x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()) else torch.contiguous_format
        y = torch.empty([n, c, yh, yw], dtype=x.dtype, device=x.device, memory_format=fmt)
        read = si is not None and si.numel() > 0
        s, so, sw_active = (si if read else None), None, 0
        if write_signs:                                           # sizes as filtered_lrelu.cpp:89-97
            sw_active = yw * down - (down - 1) + fd2.shape[1] - 1
            sh = yh * down - (down - 1) + fd2.shape[0] - 1
            s = so = torch.zeros([n, c, sh, ((sw_active + 15) & ~15) >> 2], dtype=torch.uint8, device=x.device)
        elif read:
            assert s.dtype == torch.uint8 and s.is_contiguous() and s.ndim == 4 and tuple(s.shape[:2]) == (n, c)
            sw_active = s.shape[3] << 2
        mode = 1 if write_signs else (2 if read else 0)
        xs, xst = _sizes4(x)
        ys, yst = _sizes4(y)
        bb = None if b is None else b.to(x.dtype)
        code = _lib.lib().p3d_filtered_lrelu(_lib.ptr(x), _lib.ptr(fu2), _lib.ptr(fd2), _lib.ptr(bb), _lib.ptr(s), _lib.ptr(y), _lib.DTYPE_CODE[x.dtype],
                                             xs, xst, ys, yst, 0 if bb is None else bb.stride(0), fu2.shape[1], fu2.shape[0], fd2.shape[1], fd2.shape[0],
                                             up, down, px0, py0, 0 if s is None else s.shape[3], 0 if s is None else s.shape[2], sx, sy, (sw_active + 3) >> 2,
                                             float(gain), float(slope), float(clamp), int(bool(flip_filters)), mode, _lib.stream_of(x))
        if code == _lib.P3D_ERR_UNSUPPORTED:
            return torch.empty([0], device=x.device), torch.empty([0], device=x.device), -1
        _lib.check(code, 'filtered_lrelu')
        return y, (so if so is not None else torch.empty([0], device=x.device)), 0

    @staticmethod
    def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, write_signs):
        assert x.is_cuda and x.ndim == 4 and x.dtype in (torch.float16, torch.float32, torch.float64)
        n, c, h, w = x.shape
        read = si is not None and si.numel() > 0
        s, so = (si if read else None), None
        if write_signs:
            s = so = torch.zeros([n, c, h, ((w + 15) & ~15) >> 2], dtype=torch.uint8, device=x.device)
        mode = 1 if write_signs else (2 if read else 0)
        xs, xst = _sizes4(x)
        code = _lib.lib().p3d_filtered_lrelu_act(_lib.ptr(x), _lib.ptr(s), _lib.DTYPE_CODE[x.dtype], xs, xst, 0 if s is None else s.shape[3] << 2,
                                                 0 if s is None else s.shape[2], sx, sy, float(gain), float(slope), float(clamp), mode, _lib.stream_of(x))
        _lib.check(code, 'filtered_lrelu_act')
        return so if so is not None else torch.empty([0], device=x.device)


_plugin = _Plugin

_op_cache = {}


def _native_op(up, down, padding, gain, slope, clamp, flip_filter):
    """autograd Function of one static configuration (the reference caches them the same way, filtered_lrelu.py:164-168)."""
    key = (up, down, padding, gain, slope, clamp, flip_filter)
    if key in _op_cache:
        return _op_cache[key]
    px0, px1, py0, py1 = padding
    clamp_f = float('inf') if clamp is None else clamp

    class FilteredLRelu(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fu, fd, b, si, sx, sy):
            have_signs = si is not None and si.numel() > 0
            keep_signs = (not have_signs) and (x.requires_grad or (b is not None and b.requires_grad))
            y, so, rc = _Plugin.filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp_f, flip_filter, keep_signs)
            if rc < 0:                                            # generic route, still with the 2-bit sign tensor as the only saved state
                t = x if b is None else x + b.to(x.dtype).reshape(1, -1, 1, 1)
                t = upfirdn2d.upfirdn2d(x=t, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
                so = _Plugin.filtered_lrelu_act_(t, si, sx, sy, gain, slope, clamp_f, keep_signs)
                y = upfirdn2d.upfirdn2d(x=t, f=fd, down=down, flip_filter=flip_filter)
            ctx.save_for_backward(fu, fd, si if have_signs else so)
            ctx.geometry = (x.shape, y.shape, sx, sy)
            return y

        @staticmethod
        def backward(ctx, dy):
            fu, fd, signs = ctx.saved_tensors
            (_, _, xh, xw), (_, _, yh, yw), sx, sy = ctx.geometry
            assert not any(ctx.needs_input_grad[i] for i in (1, 2, 4, 5, 6)), 'filtered_lrelu: only x and b are differentiable'
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
                fuw, fuh = _get_filter_size(fu)
                fdw, fdh = _get_filter_size(fd)
                # the adjoint: the same op with up <-> down, fu <-> fd, mirrored filters, no clamp, and the saved signs, whose origin moves by the
                # up-sampling filter's reach (filtered_lrelu.py:254-266)
                adj_pad = ((fuw - 1) + (fdw - 1) - px0, xw * up - yw * down + px0 - (up - 1),
                           (fuh - 1) + (fdh - 1) - py0, xh * up - yh * down + py0 - (up - 1))
                adjoint = _native_op(down, up, adj_pad, gain * (up ** 2) / (down ** 2), slope, None, not flip_filter)
                dx = adjoint.apply(dy, fd, fu, None, signs, sx - (fuw - 1) + px0, sy - (fuh - 1) + py0)
                if ctx.needs_input_grad[3]:
                    db = dx.sum([0, 2, 3])
            return dx, None, None, db, None, None, None

    _op_cache[key] = FilteredLRelu
    return FilteredLRelu
