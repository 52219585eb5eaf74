"""``conv2d`` / ``conv_transpose2d`` with gradients of arbitrary order and a global switch that skips
weight gradients (used for R1, loss.py:873).  Mirror of torch_utils/ops/conv2d_gradfix.py:26-197.

Every derivative of a convolution is again a convolution of the same family, so one autograd
Function (``_Conv``) expresses data gradients as the transposed op and a second one
(``_ConvWeightGrad``) the weight gradient; both route their dense arithmetic through
``_conv_impl`` / ``_weight_grad_impl`` — the single place where the MFMA implicit-GEMM kernels of
libp3d_hip.so (csrc/conv2d.hip) take over from the vendor library for the shapes they cover.
"""
import contextlib

import torch

enabled = False                     # set True by training_loop.py:281
weight_gradients_disabled = False   # toggled by no_weight_gradients()


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    prev = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = prev


def _tup(v, n=2):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _should_use_custom_op(input):
    assert isinstance(input, torch.Tensor)
    if (not enabled) or (not torch.backends.cudnn.enabled):
        return False
    return input.device.type == 'cuda'


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _should_use_custom_op(input):
        return _Conv.apply(input, weight, bias, _Cfg(False, weight.shape, stride, padding, 0, dilation, groups))
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _should_use_custom_op(input):
        return _Conv.apply(input, weight, bias, _Cfg(True, weight.shape, stride, padding, output_padding, dilation, groups))
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


class _Cfg:
    """Static description of one convolution (the cache key of the reference, conv2d_gradfix.py:78-80)."""
    __slots__ = ('transpose', 'wshape', 'stride', 'padding', 'output_padding', 'dilation', 'groups')

    def __init__(self, transpose, wshape, stride, padding, output_padding, dilation, groups):
        self.transpose, self.wshape, self.groups = bool(transpose), tuple(wshape), int(groups)
        self.stride, self.padding = _tup(stride), _tup(padding)
        self.output_padding, self.dilation = _tup(output_padding), _tup(dilation)
        assert self.groups >= 1 and len(self.wshape) == 4
        assert all(s >= 1 for s in self.stride) and all(p >= 0 for p in self.padding) and all(d >= 0 for d in self.dilation)
        if not self.transpose:
            assert all(o == 0 for o in self.output_padding)
        else:
            assert all(0 <= o < max(s, d) for o, s, d in zip(self.output_padding, self.stride, self.dilation))

    def flipped(self, out_shape, in_shape):
        """Config of the op computing d(input) from d(output): the opposite direction, with the output_padding
        that makes the shapes round-trip (conv2d_gradfix.py:95-104)."""
        op = (0, 0)
        if not self.transpose:
            kh, kw = self.wshape[2:]
            op = tuple(in_shape[i + 2] - (out_shape[i + 2] - 1) * self.stride[i] - (1 - 2 * self.padding[i]) - self.dilation[i] * (k - 1)
                       for i, k in enumerate((kh, kw)))
        return _Cfg(not self.transpose, self.wshape, self.stride, self.padding, op, self.dilation, self.groups)


def _conv_impl(x, w, b, cfg):
    """Dense arithmetic of one (possibly transposed) convolution."""
    if not cfg.transpose:
        return torch.nn.functional.conv2d(x, w, b, stride=cfg.stride, padding=cfg.padding, dilation=cfg.dilation, groups=cfg.groups)
    return torch.nn.functional.conv_transpose2d(x, w, b, stride=cfg.stride, padding=cfg.padding, output_padding=cfg.output_padding,
                                                groups=cfg.groups, dilation=cfg.dilation)


def _is_pointwise(cfg):
    return cfg.wshape[2:] == (1, 1) and cfg.stride == (1, 1) and cfg.dilation == (1, 1) and cfg.padding == (0, 0)


def _weight_grad_impl(grad_output, x, cfg):
    """d(weight) for y = conv(x, w): for the transposed op the roles of x and grad_output swap."""
    if _is_pointwise(cfg) and not cfg.transpose:          # 1x1: a batched matmul over pixels (conv2d_gradfix.py:165-170)
        g = cfg.groups
        a = grad_output.reshape(grad_output.shape[0], g, grad_output.shape[1] // g, -1).permute(1, 2, 0, 3).flatten(2)
        bmat = x.reshape(x.shape[0], g, x.shape[1] // g, -1).permute(1, 2, 0, 3).flatten(2)
        return (a @ bmat.transpose(1, 2)).reshape(cfg.wshape)
    _, gw, _ = torch.ops.aten.convolution_backward(
        grad_output, x, torch.empty(cfg.wshape, dtype=x.dtype, device=x.device), None,
        list(cfg.stride), list(cfg.padding), list(cfg.dilation), cfg.transpose, list(cfg.output_padding), cfg.groups,
        [False, True, False])
    return gw


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, cfg):
        assert w.shape == cfg.wshape
        ctx.save_for_backward(x if w.requires_grad else None, w if x.requires_grad else None)
        ctx.cfg, ctx.x_shape = cfg, x.shape
        if _is_pointwise(cfg) and not cfg.transpose and cfg.groups == 1 and x.stride(1) != 1:
            # 1x1 as a matmul keeps NCHW layout work off the conv path (conv2d_gradfix.py:117-124)
            y = (w.reshape(w.shape[0], -1) @ x.reshape(x.shape[0], x.shape[1], -1)).reshape(x.shape[0], w.shape[0], *x.shape[2:])
            return y if b is None else y + b.reshape(1, -1, 1, 1)
        return _conv_impl(x, w, b, cfg)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _Conv.apply(gy, w, None, cfg.flipped(gy.shape, ctx.x_shape))
            assert gx.shape == ctx.x_shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            gw = _ConvWeightGrad.apply(gy, x, cfg)
        if ctx.needs_input_grad[2]:
            gb = gy.sum([0, 2, 3])
        return gx, gw, gb, None


class _ConvWeightGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x, cfg):
        ctx.save_for_backward(gy if x.requires_grad else None, x if gy.requires_grad else None)
        ctx.cfg, ctx.gy_shape, ctx.x_shape = cfg, gy.shape, x.shape
        return _weight_grad_impl(gy, x, cfg)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        cfg = ctx.cfg
        ggy = gx = None
        if ctx.needs_input_grad[0]:
            ggy = _Conv.apply(x, ggw, None, cfg)
            assert ggy.shape == ctx.gy_shape
        if ctx.needs_input_grad[1]:
            gx = _Conv.apply(gy, ggw, None, cfg.flipped(ctx.gy_shape, ctx.x_shape))
            assert gx.shape == ctx.x_shape
        return ggy, gx, None
