"""``Conv2dLayer`` (networks_stylegan2.py:135-188) in training passes on the device, as ONE convolution launch per call.

The layer is ``bias_act(conv2d_resample(x, weight * weight_gain, f, down), bias, act, gain, clamp)``.  Written out with the library's
operators that is, per call: a multiply for the equalised-learning-rate gain, a cast of the weights to the activations' dtype, a
re-layout of the weights inside the convolution, the convolution, a cast of the bias and a separate bias / activation pass over the
result — for the discriminators' 64-channel layers at 512^2 that last pass moves as many bytes as the convolution itself, and the
tiny launches add up to a third of the 7 000 ATen launches of a six-phase iteration (profiles/round4_x_aten_census.txt).

Here the forward is what inference already runs (torch_utils/ops/modconv.py: the weights scaled, cast and laid tap-major ONCE per
weight version — i.e. once per optimizer step, shared by every pass of a phase — and bias, activation, gain and clamp in the
convolution's epilogue), and the backward is composed of the SAME differentiable pieces the unfused graph consists of:
``bias_act``'s gradient operator on the saved output, ``conv2d_gradfix``'s data-gradient and weight-gradient operators.  Every one of
them is an autograd Function with its own backward, so gradients of any order (R1: loss.py:873-879) come out as before, and
``conv2d_gradfix.no_weight_gradients`` is honoured.  The low-pass filter of the down-sampling layers stays the ``upfirdn2d`` call in
front of the convolution that conv2d_resample.py:108-118 makes it.

Not taken (the caller falls back to the unfused formulation): CPU tensors, up-sampling layers, channel counts that are not whole K
rows of the matrix-core kernels (the 6 / 18-channel fromrgb layers), activations other than linear / lrelu.
"""
import ctypes
import os

import torch

from ... import _lib
from . import bias_act, conv2d_gradfix, modconv, upfirdn2d

_lib.register('p3d_demod_coefs_backward', ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int32] * 4 + [ctypes.c_void_p])

enabled = os.environ.get('P3D_CONV_LAYER', '1') != '0'      # off: Conv2dLayer keeps the unfused formulation in training passes (tests / A-B measurements)
calls = {'forward': 0, 'backward': 0}


def supported(x, weight, bias, up, down, activation):
    if not (enabled and conv2d_gradfix.enabled and conv2d_gradfix.native and modconv.enabled and torch.is_grad_enabled()):
        return False
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.ndim == 4 and x.dtype in (torch.float16, torch.float32) and weight.dtype == torch.float32):
        return False
    if up != 1 or down not in (1, 2) or activation not in ('linear', 'lrelu') or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (1, 3):
        return False
    if not (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False                                       # nothing to differentiate: the inference route takes it
    if x.shape[1] % (64 if x.dtype == torch.float16 else 32) != 0:
        return False
    k = weight.shape[2]
    return not (down == 2 and k == 3 and (x.shape[2] < 3 or x.shape[3] < 3))


class _Geo:
    """Non-tensor arguments of one call."""
    __slots__ = ('k', 'stride', 'wgain', 'act_idx', 'spec', 'split')


def conv_layer(x, weight, bias, weight_gain, resample_filter, down, padding, activation, act_gain, clamp):
    """Conv2dLayer.forward for the calls ``supported`` accepts.  x any dense layout; the result is channels-last."""
    k = weight.shape[2]
    assert padding == k // 2
    geo = _Geo()
    geo.k, geo.stride, geo.wgain = k, 1, float(weight_gain)
    if down == 2:
        fw = resample_filter.shape[-1]
        p0, p1 = padding + (fw - down + 1) // 2, padding + (fw - down) // 2
        if k == 1:                                         # low-pass + decimate, then the 1x1 (conv2d_resample.py:96-100)
            x = upfirdn2d.upfirdn2d(x, resample_filter, down=down, padding=[p0, p1, p0, p1])
        else:                                              # low-pass at full rate, then the valid stride-2 correlation (conv2d_resample.py:108-111)
            x = upfirdn2d.upfirdn2d(x, resample_filter, padding=[p0, p1, p0, p1])
            geo.stride = 2
    geo.act_idx = {'linear': 0, 'lrelu': 1}[activation]
    geo.spec = bias_act._Spec.make(x, None, 1, activation, None, act_gain, clamp)
    scoped = getattr(conv2d_gradfix._scope, 'split', None)
    split = conv2d_gradfix.split_bf16 if scoped is None else bool(scoped)
    geo.split = bool(split and x.dtype == torch.float32 and k == 3)        # (the arithmetic conv2d_gradfix would have chosen for this call: exact fp32 unless opted in)
    return _ConvBiasAct.apply(x, weight, bias, geo)


def _effective_weight(weight, gain, dtype):
    """(weight * gain).to(dtype): a graph node when this backward is itself being recorded (create_graph: the R1 penalty reaches the
    parameter through the data gradient), else the per-version cached constant."""
    if torch.is_grad_enabled() and weight.requires_grad:
        return (weight * gain).to(dtype)
    return modconv._cached_weight(weight, ('effective', dtype, gain), lambda: (weight.detach() * gain).to(dtype))


class _ConvBiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, geo):
        x = conv2d_gradfix._channels_last(x)
        ci = weight.shape[1]
        wdt = modconv.BF16X3 if geo.split else x.dtype
        wmod = modconv._cached_weight(weight, ('mfma', wdt, geo.wgain),
                                      lambda: modconv.modulate_weights(weight, torch.ones([1, ci], dtype=torch.float32, device=weight.device), demodulate=False,
                                                                       pre_scale=geo.wgain, dtype=wdt))
        y = modconv.conv2d(x, wmod, bias=bias, act=geo.act_idx, gain=geo.spec.gain, clamp=geo.spec.clamp, down=geo.stride, split=geo.split)
        ctx.save_for_backward(x, weight, y if 'y' in geo.spec.ref else None)           # what bias_act keeps for its gradient (bias_act.py:143-146)
        ctx.geo, ctx.has_bias = geo, bias is not None
        calls['forward'] += 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        geo = ctx.geo
        calls['backward'] += 1
        g = dy if geo.spec.identity else bias_act._BiasActGrad.apply(dy, None, None, y, geo.spec)      # through clamp, gain and the activation (bias_act.py:187-196)
        gx = gw = gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bias_act._sum_to_bias(g, 1).to(torch.float32)
        pad = geo.k // 2 if geo.stride == 1 else 0
        cfg = conv2d_gradfix._Cfg(False, weight.shape, geo.stride, pad, 0, 1, 1, split=geo.split)
        if ctx.needs_input_grad[0]:
            gx = conv2d_gradfix._Conv.apply(g, _effective_weight(weight, geo.wgain, x.dtype), None, cfg.flipped(g.shape, x.shape))
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            gw = conv2d_gradfix._ConvWeightGrad.apply(g, x, cfg, geo.wgain)                         # fp32, times the gain: the cast's and the gain's own gradients ride in the kernel's last pass
        return gx, gw, gb, None


# ---- FullyConnectedLayer (networks_stylegan2.py:96-130) in training passes: the style affines and mapping MLPs, a few rows each ---------------------------------
# Unfused that is weight * gain, bias * gain, the product, the bias / activation pass and their four gradients — eight launches around a 4 x 512 x 512 product,
# ~210 calls per six-phase iteration.  Forward: the one-launch fc kernel inference uses (gains, bias, activation inside); backward: bias_act's gradient operator on the
# saved output, the data gradient as the same kernel on the transposed matrix (kept per weight version), the weight gradient on conv2d_gradfix's operator with the gain
# in its final pass.  All of them differentiable again (the discriminator epilogue's layers sit on R1's path).

def fc_supported(x, weight, bias, activation):
    if not (enabled and conv2d_gradfix.enabled and conv2d_gradfix.native and modconv.enabled and torch.is_grad_enabled()):
        return False
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.ndim == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and activation in ('linear', 'lrelu')):
        return False
    if not (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False
    n, rows = x.shape[0], max(weight.shape[0], weight.shape[1])
    return 1 <= n <= 16 and n * ((rows + 3) // 4 * 4) <= 16384          # the fc kernel's row budget, for the product and for its data gradient


def fc_layer(x, weight, bias, weight_gain, bias_gain, activation):
    return _Fc.apply(x, weight, bias, float(weight_gain), float(bias_gain), activation)


class _Fc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, wgain, bgain, activation):
        y = modconv.fc(x, weight, bias, wgain, bgain, activation)
        spec = bias_act._Spec.make(y, None, 1, activation, None, None, None)
        ctx.save_for_backward(x, weight, y if 'y' in spec.ref else None)
        ctx.misc = (wgain, bgain, spec, bias is not None)
        calls['forward'] += 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        wgain, bgain, spec, has_bias = ctx.misc
        calls['backward'] += 1
        g = dy if spec.identity else bias_act._BiasActGrad.apply(dy, None, None, y, spec)
        gx = gw = gb = None
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0) * bgain
        if ctx.needs_input_grad[0]:
            if torch.is_grad_enabled() and weight.requires_grad:          # create_graph: the matrix stays a view of the parameter
                wt = weight.t()
            else:
                wt = modconv._cached_weight(weight, ('fc_transposed',), lambda: weight.detach().t().contiguous())
            gx = _Fc.apply(g, wt, None, wgain, 1.0, 'linear')
        if ctx.needs_input_grad[1]:
            cfg = conv2d_gradfix._Cfg(False, (weight.shape[0], weight.shape[1], 1, 1), 1, 0, 0, 1, 1, split=False)
            gw = conv2d_gradfix._ConvWeightGrad.apply(g[:, :, None, None], x[:, :, None, None], cfg, wgain).reshape(weight.shape)
        return gx, gw, gb, None, None, None


# ---- demodulation coefficients of the modulated convolution (networks_stylegan2.py:57-63) in training passes ------------------------------------------------------
# rsqrt(sum_{i,taps} (w * s)^2 + 1e-8) as [N, Co]: written with tensor operators that is square / tap sum / square / broadcast product / sum / add / rsqrt and their
# gradients — ~15 launches for a [4, 512] result, ~55 calls per iteration.  Forward: the kernel inference uses (tap sums of w^2 kept per weight version); backward: two
# kernels (p3d_demod_coefs_backward).  When the backward is itself being recorded (create_graph) the tensor-operator formulation is differentiated instead.

def demod_supported(weight, styles):
    if not (enabled and conv2d_gradfix.enabled and conv2d_gradfix.native and modconv.enabled and torch.is_grad_enabled()):
        return False
    if not (isinstance(weight, torch.nn.Parameter) and weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.is_contiguous()):
        return False                                       # (a derived tensor — the fp16 pre-scaled weights — has no version to key the tap sums on)
    if not (styles.is_cuda and styles.dtype == torch.float32 and styles.ndim == 2 and (weight.requires_grad or styles.requires_grad)):
        return False
    n = styles.shape[0]
    return 1 <= n <= 16 and n * max(weight.shape[0], weight.shape[1]) * 4 <= 64 * 1024


def demod_reference(weight, styles):
    """The tensor-operator formulation (taps summed first: the [N, O, I, k, k] product of the reference is never formed)."""
    energy = weight.square().sum(dim=(2, 3))
    return torch.rsqrt((styles.square().unsqueeze(1) * energy.unsqueeze(0)).sum(dim=2) + 1e-8)


def demod(weight, styles):
    return _Demod.apply(weight, styles)


class _Demod(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, styles):
        d = modconv.demod_coefs(weight, styles)
        ctx.save_for_backward(weight, styles, d)
        calls['forward'] += 1
        return d

    @staticmethod
    def backward(ctx, gd):
        weight, styles, d = ctx.saved_tensors
        calls['backward'] += 1
        if torch.is_grad_enabled():                        # create_graph: differentiate the plain formulation, to any order
            with torch.enable_grad():
                wanted = [t for t, need in ((weight, ctx.needs_input_grad[0]), (styles, ctx.needs_input_grad[1])) if need]
                grads = list(torch.autograd.grad(demod_reference(weight, styles), wanted, gd, create_graph=True))
            return (grads.pop(0) if ctx.needs_input_grad[0] else None), (grads.pop(0) if ctx.needs_input_grad[1] else None)
        co, ci, kh, kw = weight.shape
        n = styles.shape[0]
        w2 = modconv._cached_weight(weight, 'w2_tapsum', lambda: weight.detach().float().square().sum(dim=[2, 3]).contiguous())
        s32, gd32 = styles.detach().contiguous(), gd.detach().float().contiguous()
        gw = torch.empty_like(weight, memory_format=torch.contiguous_format) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(s32) if ctx.needs_input_grad[1] else None
        code = _lib.lib().p3d_demod_coefs_backward(_lib.ptr(gd32), _lib.ptr(d), _lib.ptr(s32), _lib.ptr(w2), _lib.ptr(weight.detach()), _lib.ptr(gs), _lib.ptr(gw),
                                                   n, ci, co, kh * kw, _lib.stream_of(gd32))
        _lib.check(code, 'demod_coefs_backward')
        return gw, gs
