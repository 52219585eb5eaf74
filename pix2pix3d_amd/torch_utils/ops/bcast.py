"""Per-(image, channel) scaling of an activation tensor, ``fma`` with a per-pixel addend, and their gradient reductions on the device
(csrc/bcast_ops.hip): the element-wise half of the unfused modulated convolution the training passes run
(reference: training/networks_stylegan2.py:70-79, torch_utils/ops/fma.py:17-60, bias_act.py:190-193).

No new reference API — these are the kernels behind ``x * styles``, ``fma.fma`` and the bias-gradient sums where the operands are dense
device tensors; anything else stays on the tensor-op formulation.  Arithmetic is fp32 with ONE rounding to the tensor dtype, which is
what the reference's fp16 multiplies / addcmul produce."""
import ctypes

import torch

from ... import _lib

_vp, _i32 = ctypes.c_void_p, ctypes.c_int32
_lib.register('p3d_bcast_fma', ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, _i32, _i32, _i32, _i32, _i32, _vp])
_lib.register('p3d_channel_dot_workspace', ctypes.c_int64, [_i32, _i32, _i32, _i32])
_lib.register('p3d_channel_dot', ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, _i32, _i32, _i32, _i32, _vp])
_lib.register('p3d_pixel_sum', ctypes.c_int, [_vp, _vp, ctypes.c_int, _i32, _i32, _i32, _i32, _vp])

enabled = True
calls = {'fma': 0, 'dot': 0, 'sum': 0}          # launches through the native route (tests / census)


def layout(x):
    """(channels_last, N, A, B) of a dense 4-D device tensor as [N][A][B] with B contiguous, or None when the native kernels do not apply."""
    if not (enabled and isinstance(x, torch.Tensor) and x.ndim == 4 and x.is_cuda and x.dtype in (torch.float16, torch.float32)) or x.numel() == 0:
        return None
    n, c, h, w = x.shape
    vec = 8 if x.dtype == torch.float16 else 4
    if x.is_contiguous():
        geo = (0, n, c, h * w)
    elif x.is_contiguous(memory_format=torch.channels_last):
        geo = (1, n, h * w, c)
    else:
        return None
    if not cl_ok(geo):                                          # csrc/bcast_ops.hip: the NCHW kernels index (n, c) rows with 16 bits
        return None
    return geo if geo[3] % vec == 0 and x.data_ptr() % 16 == 0 else None


def cl_ok(geo):
    return bool(geo[0]) or geo[1] * geo[2] < 65536


def _aligned(*ts):
    """Every operand the C side takes by 16-byte vectors (p3d_bcast_fma / p3d_channel_dot check it and refuse otherwise)."""
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def _scale_fma(x, geo, s, z):
    """y = x * s[n, c] (+ z[n or 0, 0, h, w]); s float32 [N, C] contiguous, z None or x.dtype [Nz, 1, H, W] contiguous."""
    cl, n, a, b = geo
    y = torch.empty_like(x)                                                   # preserves the dense layout
    code = _lib.lib().p3d_bcast_fma(_lib.ptr(x), _lib.ptr(s), _lib.ptr(z), _lib.ptr(y), _lib.DTYPE_CODE[x.dtype], cl, n, a, b,
                                    int(z is not None and z.shape[0] > 1), _lib.stream_of(x))
    _lib.check(code, 'bcast_fma')
    calls['fma'] += 1
    return y


def _channel_dot(p, q, geo, merge_batch=False):
    """float32 [N, C]: sum over the pixels of p * q (q None: of p).  merge_batch (channels-last only): one [1, C] sum over all images."""
    cl, n, a, b = geo
    if merge_batch and cl:
        n, a = 1, n * a
    c = b if cl else a
    out = torch.empty([n, c], dtype=torch.float32, device=p.device)
    nbytes = int(_lib.lib().p3d_channel_dot_workspace(cl, n, a, b))
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=p.device) if nbytes else None
    code = _lib.lib().p3d_channel_dot(_lib.ptr(p), _lib.ptr(q), _lib.ptr(out), _lib.ptr(work), nbytes, _lib.DTYPE_CODE[p.dtype], cl, n, a, b, _lib.stream_of(p))
    _lib.check(code, 'channel_dot')
    calls['dot'] += 1
    return out


def _pixel_sum(p, geo):
    """float32 [N, 1, H, W]: sum over the channels."""
    cl, n, a, b = geo
    out = torch.empty([n, 1, p.shape[2], p.shape[3]], dtype=torch.float32, device=p.device)
    code = _lib.lib().p3d_pixel_sum(_lib.ptr(p), _lib.ptr(out), _lib.DTYPE_CODE[p.dtype], cl, n, a, b, _lib.stream_of(p))
    _lib.check(code, 'pixel_sum')
    calls['sum'] += 1
    return out


def _same_layout(t, like):
    """``t`` in the dense layout of ``like`` (gradients arrive in whatever layout autograd produced)."""
    fmt = torch.contiguous_format if like.is_contiguous() else torch.channels_last
    return t.contiguous(memory_format=fmt)


class _ScaleChannels(torch.autograd.Function):
    """y = x * s[:, :, None, None]."""

    @staticmethod
    def forward(ctx, x, s):
        geo = layout(x)
        s_x = s.to(x.dtype)                                                   # the factor the reference multiplies by (rounded to x's dtype)
        ctx.save_for_backward(x, s_x)
        ctx.s_dtype = s.dtype
        return _scale_fma(x, geo, s_x.float().contiguous(), None)

    @staticmethod
    def backward(ctx, gy):
        x, s_x = ctx.saved_tensors
        need_x, need_s = ctx.needs_input_grad
        if torch.is_grad_enabled():                                           # higher-order: stay differentiable
            n, c = s_x.shape
            return (gy * s_x.reshape(n, c, 1, 1) if need_x else None, (gy * x).sum([2, 3]).to(ctx.s_dtype) if need_s else None)
        gy = _same_layout(gy.to(x.dtype), x)
        geo = layout(x)
        if not _aligned(gy):                                                  # (a view into a larger buffer): the tensor-op formulation
            n, c = s_x.shape
            return (gy * s_x.reshape(n, c, 1, 1) if need_x else None, (gy.float() * x.float()).sum([2, 3]).to(ctx.s_dtype) if need_s else None)
        gx = _scale_fma(gy, geo, s_x.float().contiguous(), None) if need_x else None
        gs = _channel_dot(gy, x, geo).to(ctx.s_dtype) if need_s else None
        return gx, gs


def scale_channels_supported(x, s):
    return layout(x) is not None and s.ndim == 2 and s.shape == x.shape[:2] and s.is_cuda


def scale_channels(x, s):
    """x [N,C,H,W] * s [N,C] broadcast over the pixels, with the fused backward (gx = gy * s, gs = sum_hw gy * x)."""
    return _ScaleChannels.apply(x, s)


class _FmaNative(torch.autograd.Function):
    """a * b + c with b [N,C,1,1] and c [N or 1, 1, H, W] (the demodulation coefficients and the noise image)."""

    @staticmethod
    def forward(ctx, a, b, c):
        geo = layout(a)
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        ctx.c_dtype = c.dtype
        return _scale_fma(a, geo, b.reshape(b.shape[0], b.shape[1]).float().contiguous(), c.to(a.dtype).contiguous())

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        need_a, need_b, need_c = ctx.needs_input_grad
        if torch.is_grad_enabled():
            return (dout * b if need_a else None, (dout * a).sum([2, 3], keepdim=True) if need_b else None,
                    dout.sum([0, 1] if ctx.c_shape[0] == 1 else [1], keepdim=True).to(ctx.c_dtype) if need_c else None)
        dout = _same_layout(dout.to(a.dtype), a)
        geo = layout(a)
        if not _aligned(dout):
            return (dout * b if need_a else None, (dout.float() * a.float()).sum([2, 3], keepdim=True).to(b.dtype) if need_b else None,
                    dout.sum([0, 1] if ctx.c_shape[0] == 1 else [1], keepdim=True).to(ctx.c_dtype) if need_c else None)
        da = _scale_fma(dout, geo, b.reshape(b.shape[0], b.shape[1]).float().contiguous(), None) if need_a else None
        db = _channel_dot(dout, a, geo).to(b.dtype).reshape(b.shape) if need_b else None
        dc = None
        if need_c:
            dc = _pixel_sum(dout, geo)
            if ctx.c_shape[0] == 1 and dc.shape[0] > 1:
                dc = dc.sum(0, keepdim=True)
            dc = dc.to(ctx.c_dtype)
        return da, db, dc


def fma_supported(a, b, c):
    if layout(a) is None or not (isinstance(b, torch.Tensor) and isinstance(c, torch.Tensor)) or b.dtype != a.dtype:
        return False
    n, ch, h, w = a.shape
    if not (tuple(b.shape) == (n, ch, 1, 1) and tuple(c.shape) in ((n, 1, h, w), (1, 1, h, w)) and b.is_cuda and c.is_cuda):
        return False
    # the addend is passed as a dense [Nz,1,H,W] tensor of a's dtype: a contiguous() / to() copy is freshly allocated (aligned); an operand
    # used as is must be aligned itself
    return c.dtype != a.dtype or not c.is_contiguous() or c.data_ptr() % 16 == 0


def fma(a, b, c):
    return _FmaNative.apply(a, b, c)


class _BiasSum(torch.autograd.Function):
    """t.sum over every axis but the channel axis (dim 1) of a dense 4-D tensor."""

    @staticmethod
    def forward(ctx, t):
        geo = layout(t)
        ctx.shape = t.shape
        if geo[0]:
            return _channel_dot(t, None, geo, merge_batch=True)[0].to(t.dtype)
        return _channel_dot(t, None, geo).sum(0).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.reshape(1, -1, 1, 1).expand(ctx.shape)


def bias_sum_supported(t, dim):
    return dim == 1 and layout(t) is not None


def bias_sum(t):
    return _BiasSum.apply(t)
