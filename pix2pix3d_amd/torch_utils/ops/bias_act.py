"""``bias_act``: y = clamp(act(x + b) * gain) with first- and second-order gradients.

Mirror of the reference operator API (torch_utils/ops/bias_act.py:54-209): same function signature,
same ``activation_funcs`` table (read by networks_stylegan2.py:159, 303), same dispatch rule — device
tensors go to the native kernel, CPU tensors to plain torch ops (bias_act.py:86).  The native side is
``p3d_bias_act`` in libp3d_hip.so (csrc/bias_act.hip); there is no fallback for device tensors.
"""
import numpy as np
import torch

from ... import _lib
from ...dnnlib import EasyDict

# name -> forward function, defaults, kernel index, which forward tensor backward needs, 2nd-order support
activation_funcs = {
    'linear':   EasyDict(func=lambda x, **_: x,                                         def_alpha=0,   def_gain=1,          cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     EasyDict(func=lambda x, **_: torch.nn.functional.relu(x),               def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    EasyDict(func=lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), def_alpha=0.2, def_gain=np.sqrt(2), cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     EasyDict(func=lambda x, **_: torch.tanh(x),                             def_alpha=0,   def_gain=1,          cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  EasyDict(func=lambda x, **_: torch.sigmoid(x),                          def_alpha=0,   def_gain=1,          cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      EasyDict(func=lambda x, **_: torch.nn.functional.elu(x),                def_alpha=0,   def_gain=1,          cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     EasyDict(func=lambda x, **_: torch.nn.functional.selu(x),               def_alpha=0,   def_gain=1,          cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': EasyDict(func=lambda x, **_: torch.nn.functional.softplus(x),           def_alpha=0,   def_gain=1,          cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    EasyDict(func=lambda x, **_: torch.sigmoid(x) * x,                      def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=9, ref='x', has_2nd_grad=True),
}

_null = None  # "absent tensor" marker of the plugin protocol (an empty tensor in the reference, bias_act.py:38)


def _dense_like(x):
    """Kernel precondition (bias_act.cpp:51-55): dense, non-overlapping; keep channels_last if it is that."""
    if x.is_contiguous() or (x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return x
    return x.contiguous()


def _same_layout(t, like):
    if t is None:
        return None
    if t.stride() != like.stride() or t.dtype != like.dtype:
        t = torch.empty_like(like).copy_(t)
    return t


class _Plugin:
    """``bias_act_plugin`` equivalent (bias_act.cpp:36): tensors in, freshly allocated tensor out."""

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        if x.dtype not in _lib.DTYPE_CODE:
            raise RuntimeError(f'bias_act: unsupported dtype {x.dtype}')
        absent = lambda t: t is None or t.numel() == 0
        x = _dense_like(x)
        b = None if absent(b) else b.to(x.dtype).contiguous()
        xref = None if absent(xref) else _same_layout(xref, x)
        yref = None if absent(yref) else _same_layout(yref, x)
        dy = None if absent(dy) else _same_layout(dy, x)
        if x.numel() > 2**31 - 1:
            raise RuntimeError('bias_act: x is too large')              # bias_act.cpp:44
        if b is not None:
            if b.ndim != 1 or not (0 <= dim < x.ndim) or b.shape[0] != x.shape[dim]:
                raise RuntimeError('bias_act: b must be a vector matching x.shape[dim]')
        y = torch.empty_like(x)
        if x.numel() == 0:
            return y
        step_b = x.stride(dim) if b is not None else 1
        size_b = b.shape[0] if b is not None else 0
        code = _lib.lib().p3d_bias_act(_lib.ptr(x), _lib.ptr(b), _lib.ptr(xref), _lib.ptr(yref), _lib.ptr(dy), _lib.ptr(y),
                                       _lib.DTYPE_CODE[x.dtype], int(grad), int(act), float(alpha), float(gain), float(clamp),
                                       x.numel(), size_b, step_b, _lib.stream_of(x))
        _lib.check(code, 'bias_act')
        return y


def _plugin():
    from .. import custom_ops
    return custom_ops.get_plugin('bias_act_plugin')


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """Fused bias + activation (+ gain, + clamp).  Arguments as in the reference (bias_act.py:54-91)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda':
        return _BiasAct.apply(x, b, _Spec.make(x, b, dim, act, alpha, gain, clamp))
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Plain-torch path for CPU tensors (and impl='ref'); differentiable to any order through autograd."""
    assert isinstance(x, torch.Tensor) and (clamp is None or clamp >= 0)
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        x = x + b.reshape([-1 if d == dim else 1 for d in range(x.ndim)])
    y = spec.func(x, alpha=alpha)
    y = y if gain == 1 else y * gain
    return y if clamp is None or clamp < 0 else y.clamp(-float(clamp), float(clamp))


class _Spec:
    """Hashable bundle of the non-tensor arguments."""
    __slots__ = ('dim', 'act', 'idx', 'alpha', 'gain', 'clamp', 'ref', 'second', 'identity')

    @staticmethod
    def make(x, b, dim, act, alpha, gain, clamp):
        assert clamp is None or clamp >= 0
        table = activation_funcs[act]
        s = _Spec()
        s.dim, s.act, s.idx = int(dim), act, table.cuda_idx
        s.alpha = float(alpha if alpha is not None else table.def_alpha)
        s.gain = float(gain if gain is not None else table.def_gain)
        s.clamp = float(clamp if clamp is not None else -1)
        s.ref, s.second = table.ref, table.has_2nd_grad
        s.identity = (act == 'linear' and s.gain == 1 and s.clamp < 0)       # bias_act.py:151, 168 shortcuts
        if b is not None:
            assert isinstance(b, torch.Tensor) and b.ndim == 1 and 0 <= s.dim < x.ndim and b.shape[0] == x.shape[s.dim]
        return s


def _run(spec, grad, x, b, xref, yref, dy):
    return _plugin().bias_act(x, b, xref, yref, dy, grad, spec.dim, spec.idx, spec.alpha, spec.gain, spec.clamp)


def _sum_to_bias(t, dim):
    from . import bcast
    if bcast.bias_sum_supported(t, dim):                  # one read of the tensor (csrc/bcast_ops.hip) instead of a generic reduction
        return bcast.bias_sum(t)
    return t.sum([i for i in range(t.ndim) if i != dim])


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, spec):
        x = _dense_like(x)
        b_ = b.contiguous() if b is not None else None
        y = x
        if not (spec.identity and b_ is None):
            y = _run(spec, 0, x, b_, None, None, None)
        keep_x = ('x' in spec.ref) or spec.second
        ctx.save_for_backward(x if keep_x else _null, b_ if keep_x else _null, y if 'y' in spec.ref else _null)
        ctx.spec, ctx.has_b = spec, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        spec = ctx.spec
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy if spec.identity else _BiasActGrad.apply(dy, x, b, y, spec)
            if ctx.needs_input_grad[1] and ctx.has_b:
                db = _sum_to_bias(dx, spec.dim)
        return dx, db, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, spec):
        dy = _dense_like(dy)
        dx = _run(spec, 1, dy, b, x, y, None)
        ctx.save_for_backward(dy if spec.second else _null, x, b, y)
        ctx.spec = spec
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        spec = ctx.spec
        d_dy = d_x = d_b = None
        d_dx = _dense_like(d_dx)
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, spec)
        if spec.second and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _run(spec, 2, d_dx, b, x, y, dy)
            if ctx.needs_input_grad[2] and b is not None:
                d_b = _sum_to_bias(d_x, spec.dim)
        return d_dy, d_x, d_b, None, None
