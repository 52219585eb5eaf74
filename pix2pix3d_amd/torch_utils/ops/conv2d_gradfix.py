"""``conv2d`` / ``conv_transpose2d`` with gradients of arbitrary order and a global switch that skips
weight gradients (used for R1, loss.py:873).  Mirror of torch_utils/ops/conv2d_gradfix.py:26-197.

Every derivative of a convolution is again a convolution of the same family, so one autograd
Function (``_Conv``) expresses data gradients as the transposed op and a second one
(``_ConvWeightGrad``) the weight gradient; both route their dense arithmetic through
``_conv_impl`` / ``_weight_grad_impl``.  For device tensors those two call libp3d_hip.so
(``p3d_conv2d_forward`` / ``p3d_conv2d_bwd_weight``, csrc/conv2d_grad.hip: MFMA implicit GEMM,
channels-last, fp16 / fp32) for every geometry conv2d_resample ever asks for — 1x1 and 3x3 at
stride 1 with "same" padding, 3x3 at stride 2 without padding, plain and transposed — so neither
the forward nor any gradient order of a training step enters the vendor convolution library.
Anything else (groups, dilation, other strides; CPU tensors) takes torch's own operators.
"""
import contextlib
import threading
import ctypes
import os

import torch

from ... import _lib

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


_scope = threading.local()          # .split: None = the module default (split_bf16), True / False = what the enclosing products() asked for


@contextlib.contextmanager
def products(bf16x3):
    """Arithmetic of the fp32 forward / data-gradient convolutions CALLED inside this block (and of their gradients, whenever those run: the choice
    travels with the op's configuration): True = three bf16 MFMAs per product ("bf16x3"), False = exact fp32 products, None = the module default.
    The generator wraps its own passes in it (training/triplane.py: ``train_products``) — the discriminators' convolutions, called outside, keep the
    default (exact fp32: their R1 gradient fields are the ones that react to 5e-6, see split_bf16 below)."""
    prev = getattr(_scope, 'split', None)
    _scope.split = bf16x3
    try:
        yield
    finally:
        _scope.split = prev


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
    __slots__ = ('transpose', 'wshape', 'stride', 'padding', 'output_padding', 'dilation', 'groups', 'split')

    def __init__(self, transpose, wshape, stride, padding, output_padding, dilation, groups, split=None):
        self.transpose, self.wshape, self.groups = bool(transpose), tuple(wshape), int(groups)
        scoped = getattr(_scope, 'split', None)
        self.split = split if split is not None else (split_bf16 if scoped is None else bool(scoped))      # fixed when the op is first called; its gradients inherit it
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
        return _Cfg(not self.transpose, self.wshape, self.stride, self.padding, op, self.dilation, self.groups, split=self.split)


native = True              # device tensors of the covered family go to libp3d_hip.so; False = torch's operators everywhere
split_bf16 = os.environ.get('P3D_TRAIN_BF16X3', '0') == '1'      # opt-in: fp32 forward / data-gradient convolutions as three bf16 MFMAs per product (csrc/conv2d.hip,
                                                                 # "bf16x3": ~5e-6 of the output range, 2.5x the fp32 matrix rate; weight gradients stay exact fp32 either
                                                                 # way).  Training defaults to EXACT fp32 products — the reference's stance (training_loop.py:278-280 turns
                                                                 # TF32 off): a 5e-6 perturbation carries a handful of leaky-ReLU pre-activations across zero per pass,
                                                                 # which moves R1 gradient fields by ~1 % in L2 (tests/test_discriminator.py) — fine for inference, not a
                                                                 # default for a training run that is meant to reproduce the reference's
native_calls = {'forward': 0, 'weight_grad': 0, 'aten': 0}      # which route the dense arithmetic took (tests)
aten_log = []                # the first few calls that went to torch's operators: (what, transpose, weight shape, stride, padding, output_padding, dtypes, input shape)


def _note_aten(what, cfg, x, other):
    native_calls['aten'] += 1
    if len(aten_log) < 16:
        aten_log.append((what, cfg.transpose, cfg.wshape, cfg.stride, cfg.padding, cfg.output_padding, cfg.groups, str(x.dtype), str(getattr(other, 'dtype', None)), tuple(x.shape), str(x.device)))

_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
_lib.register('p3d_conv2d_forward', ctypes.c_int, [_vp] * 5 + [ctypes.c_int] + [_i32] * 10 + [_vp, _i64, _vp])
_lib.register('p3d_conv2d_bwd_data', ctypes.c_int, [_vp] * 5 + [ctypes.c_int] + [_i32] * 10 + [_vp, _i64, _vp])
_lib.register('p3d_conv2d_forward_workspace', _i64, [ctypes.c_int] + [_i32] * 8)
_lib.register('p3d_conv2d_bwd_weight_workspace', _i64, [ctypes.c_int] + [_i32] * 6)
_lib.register('p3d_conv2d_bwd_weight', ctypes.c_int, [_vp] * 4 + [_i64, ctypes.c_int] + [_i32] * 10 + [_vp])
_lib.register('p3d_conv2d_bwd_weight_scaled', ctypes.c_int, [_vp] * 4 + [_i64, ctypes.c_int] + [_i32] * 10 + [ctypes.c_float, _vp])

_zero_pages = {}


def _zeros_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(256, dtype=torch.float32, device=device)
    return z


def _native_geometry(x, w, cfg):
    """(k, stride) when libp3d_hip.so covers this op for these tensors, else None."""
    if not native or not x.is_cuda or x.dtype not in (torch.float16, torch.float32) or w.dtype != x.dtype or x.ndim != 4:
        return None
    kh, kw = cfg.wshape[2:]
    if cfg.groups != 1 or cfg.dilation != (1, 1) or kh != kw or kh not in (1, 3) or cfg.stride[0] != cfg.stride[1] or cfg.padding[0] != cfg.padding[1]:
        return None
    if cfg.stride == (1, 1) and cfg.padding == (kh // 2, kh // 2) and cfg.output_padding == (0, 0):
        return kh, 1
    if kh == 3 and cfg.stride == (2, 2) and cfg.padding == (0, 0) and all(o in (0, 1) for o in cfg.output_padding):
        return 3, 2
    return None


def _channels_last(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _native_conv(x, w, cfg, k, stride):
    """One launch of p3d_conv2d_forward (+ the weight re-layout it does): x any dense layout -> y channels-last."""
    n, ci, h, wd = x.shape
    tr = cfg.transpose
    co = w.shape[1] if tr else w.shape[0]
    assert w.shape[0 if tr else 1] == ci, 'conv2d_gradfix: weight / input channel mismatch'
    if not tr:
        oh, ow = (h, wd) if stride == 1 else ((h - 3) // 2 + 1, (wd - 3) // 2 + 1)
    else:
        oh, ow = (h, wd) if stride == 1 else (2 * h + 1 + cfg.output_padding[0], 2 * wd + 1 + cfg.output_padding[1])
    if k == 1 and h == 1 and wd == 1 and x.dtype == torch.float32 and n <= 16 and n * ((ci + 3) // 4 * 4) <= 16384:
        # a 1x1 convolution of 1x1 images IS a fully connected layer on a few rows — what FullyConnectedLayer routes here in training passes (style affines,
        # mapping MLPs: ~210 forward / data-gradient calls per six-phase iteration, 28 us each as weight re-layout + GEMM-tile kernel + epilogue): one launch of
        # the fc kernel instead; the transposed op (the data gradient) reads the weight matrix transposed
        from . import modconv
        wm = w.reshape(w.shape[0], w.shape[1])
        if tr:
            wm = wm.t()
        y = modconv.fc(x.reshape(n, ci), wm.contiguous(), None, 1.0, 1.0)
        native_calls['forward'] += 1
        return y.reshape(n, co, 1, 1)
    mult = 64 if x.dtype == torch.float16 else 32
    skinny = k == 1 and (ci % mult != 0 or co < 32)
    x = _channels_last(x)
    w = w.contiguous()
    if not skinny and ci % mult != 0 and not (tr and stride == 2 and x.dtype == torch.float16 and ci % 32 == 0 and co % 128 == 0 and h >= 32 and wd >= 32):
        cip = (ci + mult - 1) // mult * mult            # whole K rows for the matrix-core kernel: zero channels change nothing
        xp = torch.empty([n, cip, h, wd], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        xp[:, :ci] = x
        xp[:, ci:] = 0
        wshape = list(w.shape)
        wshape[0 if tr else 1] = cip - ci
        w = torch.cat([w, w.new_zeros(wshape)], dim=0 if tr else 1).contiguous()
        x, ci = xp, cip
    y = torch.empty([n, co, oh, ow], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    scratch = None if skinny else torch.empty([co * ci * k * k], dtype=x.dtype, device=x.device)
    code_dtype = 3 if (cfg.split and x.dtype == torch.float32 and not skinny and k == 3 and ci % 32 == 0) else _lib.DTYPE_CODE[x.dtype]     # 3 = P3D_F32_BF16X3
    if code_dtype == 0 and not skinny and ci % 32 == 0:
        from . import modconv
        if modconv.f32_x6:
            code_dtype = 4                                     # P3D_F32_BF16X6: fp32-accurate products on the bf16 matrix pipe (opt-in, modconv.f32_x6)
    nbytes = 0 if skinny else int(_lib.lib().p3d_conv2d_forward_workspace(code_dtype, n, h, wd, ci, co, k, stride, int(tr)))
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device) if nbytes > 0 else None      # split-K partial tiles (low-resolution layers)
    code = _lib.lib().p3d_conv2d_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(scratch), _lib.ptr(_zeros_page(x.device)), code_dtype,
                                         n, h, wd, ci, co, k, stride, int(tr), oh if tr and stride == 2 else 0, ow if tr and stride == 2 else 0,
                                         _lib.ptr(work), nbytes, _lib.stream_of(x))
    _lib.check(code, 'conv2d_forward')
    native_calls['forward'] += 1
    log = _lib.kernel_events.get('conv_flops')               # bench.py's arithmetic floor: (arithmetic class, multiply-add FLOPs) of every native convolution
    if log is not None:
        log.append(('bf16x3' if code_dtype == 3 else ('bf16x6' if code_dtype == 4 else str(x.dtype)), 2.0 * n * ci * co * k * k * (h * wd if (tr and stride == 2) else oh * ow)))
    return y


def _conv_impl(x, w, b, cfg):
    """Dense arithmetic of one (possibly transposed) convolution."""
    geo = _native_geometry(x, w, cfg)
    if geo is not None:
        y = _native_conv(x, w, cfg, *geo)
        return y if b is None else y + b.to(y.dtype).reshape(1, -1, 1, 1)
    _note_aten('conv', cfg, x, w)
    if not cfg.transpose:
        return torch.nn.functional.conv2d(x, w, b, stride=cfg.stride, padding=cfg.padding, dilation=cfg.dilation, groups=cfg.groups)
    return torch.nn.functional.conv_transpose2d(x, w, b, stride=cfg.stride, padding=cfg.padding, output_padding=cfg.output_padding,
                                                groups=cfg.groups, dilation=cfg.dilation)


def _is_pointwise(cfg):
    return cfg.wshape[2:] == (1, 1) and cfg.stride == (1, 1) and cfg.dilation == (1, 1) and cfg.padding == (0, 0)


def _native_weight_grad(grad_output, x, cfg, k, stride, scale=None):
    """p3d_conv2d_bwd_weight: the SMALL image of (x, grad_output) is correlated against the big one, pixels are the contraction.
    ``scale``: the gradient of the fp32 PARAMETER behind a (weight * scale).to(dtype) operand — fp32, scaled (p3d_conv2d_bwd_weight_scaled)."""
    small, big = (x, grad_output) if cfg.transpose else (grad_output, x)
    small, big = _channels_last(small), _channels_last(big)
    n, cs, hs, ws_ = small.shape
    _, cb, hb, wb = big.shape
    assert (cs, cb) == tuple(cfg.wshape[:2]) and big.shape[0] == n
    gw = torch.empty(cfg.wshape, dtype=x.dtype if scale is None else torch.float32, device=x.device)
    code_dtype = _lib.DTYPE_CODE[x.dtype]
    x6 = False
    if x.dtype == torch.float32:
        from . import modconv
        # P3D_F32_BF16X6 in the weight gradient: the whole-tile kernel's arithmetic only (both sides > 64 channels, whole aligned channel groups) — the library
        # takes the exact kernels for every other geometry, whatever the code says; mirrored here for the FLOP log's arithmetic class
        x6 = modconv.f32_x6 and modconv.wgrad_x6 and cs > 64 and cb > 64 and cs % 4 == 0 and cb % 4 == 0
        if x6:
            code_dtype = 4
    nbytes = int(_lib.lib().p3d_conv2d_bwd_weight_workspace(code_dtype, n, hs, ws_, cs, cb, k))
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device)
    if scale is None:
        code = _lib.lib().p3d_conv2d_bwd_weight(_lib.ptr(small), _lib.ptr(big), _lib.ptr(gw), _lib.ptr(work), nbytes, code_dtype, n, hs, ws_, cs, hb, wb, cb,
                                                k, stride, cfg.padding[0], _lib.stream_of(gw))
    else:
        code = _lib.lib().p3d_conv2d_bwd_weight_scaled(_lib.ptr(small), _lib.ptr(big), _lib.ptr(gw), _lib.ptr(work), nbytes, code_dtype, n, hs, ws_, cs, hb, wb, cb,
                                                       k, stride, cfg.padding[0], float(scale), _lib.stream_of(gw))
    _lib.check(code, 'conv2d_bwd_weight')
    native_calls['weight_grad'] += 1
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append(('bf16x6' if x6 else str(x.dtype), 2.0 * n * cs * cb * k * k * hs * ws_))
    return gw


def _weight_grad_impl(grad_output, x, cfg, scale=None):
    """d(weight) for y = conv(x, w): for the transposed op the roles of x and grad_output swap."""
    geo = None if grad_output.dtype != x.dtype else _native_geometry(x, torch.empty(0, dtype=x.dtype), cfg)
    if geo is not None:
        return _native_weight_grad(grad_output, x, cfg, *geo, scale=scale)
    if scale is not None:
        return _weight_grad_impl(grad_output, x, cfg).to(torch.float32) * scale
    _note_aten('weight_grad', cfg, x, grad_output)
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
        if _is_pointwise(cfg) and not cfg.transpose and cfg.groups == 1 and x.stride(1) != 1 and _native_geometry(x, w, cfg) is None:
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
            from . import bcast
            gb = bcast.bias_sum(gy) if bcast.bias_sum_supported(gy, 1) else gy.sum([0, 2, 3])
        return gx, gw, gb, None


class _ConvWeightGrad(torch.autograd.Function):
    """``scale`` (conv_layer.py): the result is (gradient.float() * scale) — what reaches an fp32 parameter whose (weight * scale).to(dtype) was the operand."""

    @staticmethod
    def forward(ctx, gy, x, cfg, scale=None):
        ctx.save_for_backward(gy if x.requires_grad else None, x if gy.requires_grad else None)
        ctx.cfg, ctx.gy_shape, ctx.x_shape, ctx.scale, ctx.dtype = cfg, gy.shape, x.shape, scale, x.dtype
        return _weight_grad_impl(gy, x, cfg, scale)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        cfg = ctx.cfg
        ggy = gx = None
        if ctx.scale is not None:
            ggw = (ggw * ctx.scale).to(ctx.dtype)
        if ctx.needs_input_grad[0]:
            ggy = _Conv.apply(x, ggw, None, cfg)
            assert ggy.shape == ctx.gy_shape
        if ctx.needs_input_grad[1]:
            gx = _Conv.apply(gy, ggw, None, cfg.flipped(ctx.gy_shape, ctx.x_shape))
            assert gx.shape == ctx.x_shape
        return ggy, gx, None, None
