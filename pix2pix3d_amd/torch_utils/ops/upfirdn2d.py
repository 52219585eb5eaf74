"""``upfirdn2d``: pad -> zero-insert upsample -> 2-D FIR -> decimate, and the filter helpers around it.

Mirror of the reference operator API (torch_utils/ops/upfirdn2d.py:72-389): ``setup_filter``,
``upfirdn2d``, ``filter2d``, ``upsample2d``, ``downsample2d`` plus the private parsers
``conv2d_resample`` imports.  Device tensors run ``p3d_upfirdn2d`` (csrc/upfirdn2d.hip); CPU tensors
run a plain-torch restatement.  Backward is the same op with up/down exchanged and the filter
mirrored (upfirdn2d.py:252-271), so gradients of every order stay on the native kernel.
"""
import numpy as np
import torch

from ... import _lib


def _pair(v, name):
    if isinstance(v, int):
        v = [v, v]
    assert isinstance(v, (list, tuple)) and len(v) == 2 and all(isinstance(e, int) for e in v), name
    return int(v[0]), int(v[1])


def _parse_scaling(scaling):
    sx, sy = _pair(scaling, 'scaling')
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    """int | [x, y] | [x0, x1, y0, y1] -> (x0, x1, y0, y1)."""
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(e, (int, np.integer)) for e in padding)
    padding = [int(e) for e in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    assert len(padding) == 4
    return tuple(padding)


def _get_filter_size(f):
    """(fw, fh) of a filter tensor; None is the 1x1 identity."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Build the fp32 FIR tensor ``upfirdn2d`` expects (reference: upfirdn2d.py:72-116).

    1-D input with < 8 taps becomes its outer product (2-D, non-separable); longer 1-D inputs stay
    separable.  ``normalize`` divides by the tap sum; ``gain`` is applied as gain**(ndim/2)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Resample a batch of 2-D images (reference: upfirdn2d.py:120-165).  x: [N, C, H, W]."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda':
        return _Upfirdn2d.apply(x, f, _Geom(up, down, padding, flip_filter, gain))
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Plain-torch path for CPU tensors (and impl='ref'): explicit zero stuffing + depthwise correlation."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32 and not f.requires_grad
    n, c, h, w = x.shape
    upx, upy = _parse_scaling(up)
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    assert w * upx + px0 + px1 >= f.shape[-1] and h * upy + py0 + py1 >= f.shape[0]

    # zero-stuffed grid: sample (i, j) lands on (i*upy, j*upx)
    z = x.new_zeros([n, c, h, upy, w, upx])
    z[:, :, :, 0, :, 0] = x
    z = z.reshape(n, c, h * upy, w * upx)
    # positive padding adds zeros, negative padding crops
    z = torch.nn.functional.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]

    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:                                   # conv2d correlates; mirror to convolve
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        z = torch.nn.functional.conv2d(z, k[None, None].expand(c, 1, -1, -1), groups=c)
    else:
        z = torch.nn.functional.conv2d(z, k[None, None, None, :].expand(c, 1, -1, -1), groups=c)
        z = torch.nn.functional.conv2d(z, k[None, None, :, None].expand(c, 1, -1, -1), groups=c)
    return z[:, :, ::dny, ::dnx]


class _Geom:
    __slots__ = ('upx', 'upy', 'dnx', 'dny', 'px0', 'px1', 'py0', 'py1', 'flip', 'gain')

    def __init__(self, up, down, padding, flip_filter, gain):
        self.upx, self.upy = _parse_scaling(up)
        self.dnx, self.dny = _parse_scaling(down)
        self.px0, self.px1, self.py0, self.py1 = _parse_padding(padding)
        self.flip, self.gain = bool(flip_filter), gain


class _Plugin:
    """``upfirdn2d_plugin`` equivalent (upfirdn2d.cpp:20): one 2-D pass, output allocated here."""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        if not (x.is_cuda and f.device == x.device):
            raise RuntimeError('upfirdn2d: x and f must live on the same GPU')
        if f.dtype != torch.float32 or f.ndim != 2 or x.ndim != 4:
            raise RuntimeError('upfirdn2d: f must be float32 rank 2 and x rank 4')
        if x.dtype not in _lib.DTYPE_CODE:
            raise RuntimeError(f'upfirdn2d: unsupported dtype {x.dtype}')
        if x.numel() > 2**31 - 1 or f.numel() > 2**31 - 1:
            raise RuntimeError('upfirdn2d: tensor too large')                 # upfirdn2d.cpp:26-32
        if min(upx, upy, downx, downy) < 1:
            raise RuntimeError('upfirdn2d: up/down factors must be at least 1')
        n, c, h, w = x.shape
        fh, fw = f.shape
        out_w = (w * upx + padx0 + padx1 - fw + downx) // downx                # upfirdn2d.cpp:39-40
        out_h = (h * upy + pady0 + pady1 - fh + downy) // downy
        if out_w < 1 or out_h < 1:
            raise RuntimeError('upfirdn2d: output must be at least 1x1')
        cl = x.ndim == 4 and x.stride(1) == 1 and c > 1 and not x.is_contiguous()
        y = torch.empty([n, c, out_h, out_w], dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        if y.numel() == 0:
            return y
        code = _lib.lib().p3d_upfirdn2d(
            _lib.ptr(x), _lib.ptr(f), _lib.ptr(y), _lib.DTYPE_CODE[x.dtype],
            _lib.i32x4(w, h, c, n), _lib.i64x4(x.stride(3), x.stride(2), x.stride(1), x.stride(0)),
            _lib.i32x2(fw, fh), _lib.i64x2(f.stride(1), f.stride(0)),
            _lib.i32x4(out_w, out_h, c, n), _lib.i64x4(y.stride(3), y.stride(2), y.stride(1), y.stride(0)),
            upx, upy, downx, downy, padx0, pady0, int(bool(flip)), float(gain), _lib.stream_of(x))
        _lib.check(code, 'upfirdn2d')
        return y


def upsample2d_add_(y, x, f):
    """y += upsample2d(x, f, up=2) in one launch where the native kernel can (channels-last fp32 / fp16, 4 x 4 filter: the skip image of the
    synthesis blocks, networks_stylegan2.py:453-459); the two-step form otherwise.  Returns y."""
    n, c, h, w = x.shape
    ok = (x.is_cuda and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4) and f.dtype == torch.float32 and x.dtype == y.dtype and x.dtype in (torch.float16, torch.float32)
          and tuple(y.shape) == (n, c, 2 * h, 2 * w) and c > 1 and c % (16 // x.element_size()) == 0 and x.stride(1) == 1 and y.stride(1) == 1
          and x.is_contiguous(memory_format=torch.channels_last) and y.is_contiguous(memory_format=torch.channels_last)
          and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0 and not (x.requires_grad or y.requires_grad))
    if not ok:
        return y.add_(upsample2d(x, f))
    code = _lib.lib().p3d_upfirdn2d_acc(
        _lib.ptr(x), _lib.ptr(f), _lib.ptr(y), _lib.DTYPE_CODE[x.dtype],
        _lib.i32x4(w, h, c, n), _lib.i64x4(x.stride(3), x.stride(2), x.stride(1), x.stride(0)),
        _lib.i32x2(4, 4), _lib.i64x2(f.stride(1), f.stride(0)),
        _lib.i32x4(2 * w, 2 * h, c, n), _lib.i64x4(y.stride(3), y.stride(2), y.stride(1), y.stride(0)),
        2, 2, 1, 1, 2, 2, 0, 4.0, _lib.stream_of(x))                             # upsample2d: padding (fw + up - 1) // 2 = 2 in front, gain up^2
    if code == _lib.P3D_ERR_UNSUPPORTED:                                         # the channels-last kernel declined this geometry: two-step form
        return y.add_(upsample2d(x, f))
    _lib.check(code, 'upfirdn2d_acc')
    return y


def _plugin():
    from .. import custom_ops
    return custom_ops.get_plugin('upfirdn2d_plugin')


def _native(x, f, g):
    """Full op on the kernel: one 2-D pass, or two 1-D passes for a separable (rank-1) filter."""
    p = _plugin()
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)                      # separable 1-tap == full 1x1 (upfirdn2d.py:241-242)
    if f.ndim == 2:
        return p.upfirdn2d(x, f, g.upx, g.upy, g.dnx, g.dny, g.px0, g.px1, g.py0, g.py1, g.flip, g.gain)
    y = p.upfirdn2d(x, f.unsqueeze(0), g.upx, 1, g.dnx, 1, g.px0, g.px1, 0, 0, g.flip, 1.0)
    return p.upfirdn2d(y, f.unsqueeze(1), 1, g.upy, 1, g.dny, 0, 0, g.py0, g.py1, g.flip, g.gain)


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, g):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2])
        y = _native(x, f, g)
        ctx.save_for_backward(f)
        ctx.g, ctx.x_shape = g, x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        g = ctx.g
        dx = None
        if ctx.needs_input_grad[0]:
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            pad = [fw - g.px0 - 1, iw * g.upx - ow * g.dnx + g.px0 - g.upx + 1,
                   fh - g.py0 - 1, ih * g.upy - oh * g.dny + g.py0 - g.upy + 1]
            gt = _Geom([g.dnx, g.dny], [g.upx, g.upy], pad, not g.flip, g.gain)
            dx = _Upfirdn2d.apply(dy, f, gt)
        assert not ctx.needs_input_grad[1], 'upfirdn2d: no gradient is defined for the filter'
        return dx, None, None


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Same-size FIR filtering when padding == 0 (reference: upfirdn2d.py:279-311)."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Integer-factor upsampling; output is ``up`` times the input when padding == 0 (upfirdn2d.py:315-350)."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Integer-factor downsampling; output is 1/``down`` of the input when padding == 0 (upfirdn2d.py:354-389)."""
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2, py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
