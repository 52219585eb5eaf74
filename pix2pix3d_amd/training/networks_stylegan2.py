"""StyleGAN2 building blocks of the tri-plane backbone, super-resolution heads and discriminator.

Mirror of training/networks_stylegan2.py (line references below are to that file): identical class
names, constructor arguments, parameter/buffer names (so ``copy_params_and_buffers`` and released
checkpoints map 1:1) and forward semantics; the arithmetic runs on this package's operators.
"""
import os

import weakref

import numpy as np
import torch

from ..torch_utils import misc
from ..torch_utils import persistence
from ..torch_utils.ops import conv2d_resample, conv2d_gradfix, upfirdn2d, bias_act, fma, modconv, bcast, conv_layer


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """x / rms(x) along ``dim`` (:28-29)."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def _fp16_prescale(weight, styles):
    """Keep fp16 products in range before demodulation cancels the scale again (:54-56): each output filter and each sample's style
    vector is divided by its largest magnitude (the filters additionally by sqrt(fan_in))."""
    cout, cin, kh, kw = weight.shape
    peak_w = weight.abs().amax(dim=(1, 2, 3), keepdim=True)
    peak_s = styles.abs().amax(dim=1, keepdim=True)
    return weight / (peak_w * np.sqrt(cin * kh * kw)), styles / peak_s


def _demod_coefficients(weight, styles):
    """rsqrt(sum_{i,ky,kx} (w[o,i,ky,kx] * s[n,i])^2 + 1e-8) as [N, O] (:65) — the taps are summed first, so the [N,O,I,k,k] product
    the reference forms for this is never materialised: an [N,O,I] broadcast product."""
    if conv_layer.demod_supported(weight, styles):               # training passes on the device: one kernel, two for its gradient
        return conv_layer.demod(weight, styles)
    return conv_layer.demod_reference(weight, styles)            # [N, O]; broadcast-sum, not a GEMM call


def _per_sample_conv(x, w_each, **resample):
    """One grouped convolution applying sample n's own filter bank w_each[n] to image n (:81-88)."""
    n, cin = x.shape[:2]
    cout, kh, kw = w_each.shape[1], w_each.shape[3], w_each.shape[4]
    flat = conv2d_resample.conv2d_resample(x=x.reshape(1, n * cin, *x.shape[2:]), w=w_each.reshape(n * cout, cin, kh, kw).to(x.dtype), groups=n, **resample)
    return flat.reshape(n, cout, *flat.shape[2:])


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True):
    """Style-modulated convolution (:34-91).  x [N,I,H,W], weight [O,I,k,k], styles [N,I], noise broadcastable to the output.

    fused: every sample gets its own filters w * s (times the demodulation coefficient) and the batch runs as one grouped conv.
    unfused: the styles scale the activations, ONE shared-weight conv runs, and the demodulation coefficient (with the noise) is
    applied to its output — the same function, and the form whose weight gradient is a plain convolution gradient."""
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    misc.assert_shape(x, [n, cin, None, None])
    misc.assert_shape(styles, [n, cin])
    if demodulate and x.dtype == torch.float16:
        weight, styles = _fp16_prescale(weight, styles)
    demod = _demod_coefficients(weight, styles) if demodulate else None
    resample = dict(f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
    if fused_modconv and x.is_cuda and conv2d_gradfix.enabled and conv2d_gradfix.native:
        # A device call that reaches this function with fused_modconv set is one the native fused kernels did not take (e.g. a 16-channel layer).  The
        # grouped convolution below would go to the vendor library; the unfused formulation is the same function on the native shared-weight kernels.
        fused_modconv = False

    if fused_modconv:
        w_each = weight.unsqueeze(0) * styles.reshape(n, 1, cin, 1, 1)
        if demod is not None:
            w_each = w_each * demod.reshape(n, cout, 1, 1, 1)
        y = _per_sample_conv(x, w_each, **resample)
        return y if noise is None else y.add_(noise)

    if bcast.scale_channels_supported(x, styles):         # dense device activations: fused scaling, fused gradient reductions
        x_mod = bcast.scale_channels(x, styles)
    else:
        x_mod = x * styles.to(x.dtype).reshape(n, cin, 1, 1)
    y = conv2d_resample.conv2d_resample(x=x_mod, w=weight.to(x.dtype), **resample)
    scale = None if demod is None else demod.to(y.dtype).reshape(n, cout, 1, 1)
    if noise is not None:
        noise = noise.to(y.dtype)
    if scale is not None and noise is not None:
        return fma.fma(y, scale, noise)
    if scale is not None:
        return bcast.scale_channels(y, demod) if bcast.scale_channels_supported(y, demod) else y * scale
    return y if noise is None else y.add_(noise)


@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    """y = act(x @ (W * lr/sqrt(in)).T + b * lr)   (:96-130)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x, out_scale=1):
        """``out_scale`` multiplies the result (ToRGBLayer folds its weight gain in here instead of a separate launch)."""
        if modconv.fc_supported(x, self.weight, self.bias, self.activation):
            return modconv.fc(x, self.weight, self.bias, self.weight_gain, self.bias_gain, self.activation, out_scale)
        if conv_layer.fc_supported(x, self.weight, self.bias, self.activation):      # training passes on the device: one launch, same gradients
            y = conv_layer.fc_layer(x, self.weight, self.bias, self.weight_gain, self.bias_gain, self.activation)
        else:
            y = self._forward_as_conv(x) if self._conv_route(x) else self._forward(x)
        return y if out_scale == 1 else y * out_scale

    def _conv_route(self, x):
        """Training passes on the device: the affine map runs as a 1x1 convolution over a [N, in, 1, 1] image, i.e. on the native
        conv2d_gradfix kernels (forward, data gradient, weight gradient, any order) instead of a vendor GEMM."""
        return (x.is_cuda and x.ndim == 2 and x.dtype in (torch.float32, torch.float16) and conv2d_gradfix.enabled and conv2d_gradfix.native
                and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))

    def _forward_as_conv(self, x):
        w = (self.weight.to(x.dtype) * self.weight_gain)[:, :, None, None]
        y = conv2d_gradfix.conv2d(x[:, :, None, None], w)[:, :, 0, 0]
        b = None if self.bias is None else self.bias.to(x.dtype) * self.bias_gain
        return bias_act.bias_act(y, b, act=self.activation)

    def _forward(self, x):
        """Generic route: equalised-learning-rate gains applied at run time, the affine map as one library call, anything but the
        identity activation through bias_act."""
        w = self.weight.to(x.dtype) * self.weight_gain
        b = None if self.bias is None else self.bias.to(x.dtype) * self.bias_gain
        if self.activation == 'linear':
            return torch.nn.functional.linear(x, w, b)
        return bias_act.bias_act(torch.nn.functional.linear(x, w), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


@persistence.persistent_class
class Conv2dLayer(torch.nn.Module):
    """Plain (unmodulated) conv + bias_act with optional resampling (:135-188)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.in_channels, self.out_channels, self.activation = in_channels, out_channels, activation
        self.up, self.down, self.conv_clamp = up, down, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        # frozen layers (freeze_layers of the discriminator) keep the same names as buffers, so checkpoints map either way
        self._hold('weight', weight, trainable)
        self._hold('bias', torch.zeros([out_channels]) if bias else None, trainable)

    def _hold(self, name, tensor, trainable):
        if tensor is None:
            setattr(self, name, None)
        elif trainable:
            setattr(self, name, torch.nn.Parameter(tensor))
        else:
            self.register_buffer(name, tensor)

    def forward(self, x, gain=1):
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if modconv.plain_layer_supported(x, self.weight, self.up, self.down, self.activation):
            return modconv.plain_layer(x, self.weight, self.bias, self.weight_gain, self.resample_filter, self.down, self.padding,
                                       self.activation, self.act_gain * gain, clamp)
        if conv_layer.supported(x, self.weight, self.bias, self.up, self.down, self.activation):       # training passes on the device: one launch, same gradients
            return conv_layer.conv_layer(x, self.weight, self.bias, self.weight_gain, self.resample_filter, self.down, self.padding,
                                         self.activation, self.act_gain * gain, clamp)
        w = self.weight * self.weight_gain
        b = self.bias.to(x.dtype) if self.bias is not None else None
        x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, activation={self.activation:s}, up={self.up}, down={self.down}'


def track_w_avg(net, w):
    """Exponential moving average of the mapping output, ``net.w_avg`` (:258-260); no-op for networks that do not track one."""
    if getattr(net, 'w_avg_beta', None) is None:
        return
    with torch.autograd.profiler.record_function('update_w_avg'):
        net.w_avg.copy_(w.detach().mean(dim=0).lerp(net.w_avg, net.w_avg_beta))


def truncate_ws(net, ws, psi, cutoff):
    """Truncation trick (:267-273): pull ws towards ``net.w_avg`` by ``psi``; with a cutoff only the first ``cutoff`` layers (in place,
    as the reference does).  psi == 1 returns ws untouched."""
    if psi == 1:
        return ws
    assert net.w_avg_beta is not None
    with torch.autograd.profiler.record_function('truncate'):
        if net.num_ws is None or cutoff is None:
            return net.w_avg.lerp(ws, psi)
        ws[:, :cutoff] = net.w_avg.lerp(ws[:, :cutoff], psi)
        return ws


@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    """z (and optional label c) -> ws [N, num_ws, w_dim], with w_avg tracking and truncation (:193-272)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.998, **unused_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed_features = 0 if c_dim == 0 else (w_dim if embed_features is None else embed_features)
        layer_features = w_dim if layer_features is None else layer_features
        sizes = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(sizes[idx], sizes[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        feats = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            feats.append(normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            feats.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        x = feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if update_emas:
            track_w_avg(self, x)
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        return truncate_ws(self, x, truncation_psi, truncation_cutoff)

    def extra_repr(self):
        return f'z_dim={self.z_dim:d}, c_dim={self.c_dim:d}, w_dim={self.w_dim:d}, num_ws={self.num_ws:d}'


@persistence.persistent_class
class SynthesisLayer(torch.nn.Module):
    """affine(w) -> modulated 3x3 conv (optionally x2 up) -> noise -> bias + lrelu (:277-337)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, **unused_kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if use_noise:                                # fixed noise image for noise_mode='const' + its learned strength (starts at 0); registration
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))      # order = the reference's (:307-310): parameters() /
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))                       # state_dict / optimizer indices line up with its runs
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, rgb=None, out_split=False, rgb_wide=None):
        """``rgb`` (device inference, from SynthesisBlock): (torgb layer, its latent, skip image) — when the native kernel can, this layer
        also adds the block's ToRGB output into the skip image from its own launch and returns (x, True); otherwise (x, False).
        ``out_split`` (device inference, bf16x3): the caller's consumers read modconv.SplitActs; x may be one.
        ``rgb_wide`` (device inference, bf16x3, from the LAST block of a network whose x nobody reads): (torgb layer, its latent, the skip image of the
        block below or None, the resampling filter) — returns (None, img) when this layer's launch also produced the block's wide image
        (modconv.conv3x3_torgb_wide: the activations are never stored), else (x, None) with x as ``out_split`` asks."""
        if noise_mode not in ('random', 'const', 'none'):
            raise AssertionError(f'unknown noise_mode {noise_mode!r}')
        in_res = self.resolution // self.up
        misc.assert_shape(x, [None, self.in_channels, in_res, in_res])
        if rgb_wide is not None:
            assert rgb is None
            torgb, w_rgb, prev, f = rgb_wide
            if fused_modconv is True and noise_mode != 'random' and x.is_cuda and \
                    modconv.conv3x3_torgb_wide_supported(x, self.weight, torgb.weight, prev, f, self.up, self.activation):
                planned = modconv.take_plan(self) if modconv._plan else None
                styles, pre = planned if planned is not None else (self.affine(w), None)
                wmod = pre[0] if pre is not None and pre[1] == ('mfma', 1, modconv.BF16X3) else modconv.modulate_weights(self.weight, styles, demodulate=True, dtype=modconv.BF16X3)
                planned_rgb = modconv.take_plan(torgb) if modconv._plan else None
                s_rgb, pre_rgb = planned_rgb if planned_rgb is not None else (torgb.affine(w_rgb, out_scale=torgb.weight_gain), None)
                rgb_wmod = pre_rgb[0] if pre_rgb is not None and pre_rgb[1] == ('rgb', modconv.BF16X3) else \
                    modconv.modulate_weights(torgb.weight, s_rgb, demodulate=False, dtype=modconv.BF16X3)
                const_noise = self.use_noise and noise_mode == 'const'
                img = modconv.conv3x3_torgb_wide(x, wmod, self.bias, self.noise_const if const_noise else None, self.noise_strength if const_noise else None,
                                                 {'linear': 0, 'lrelu': 1}[self.activation], self.act_gain * gain,
                                                 -1.0 if self.conv_clamp is None else float(self.conv_clamp * gain), rgb_wmod, torgb.bias, torgb.conv_clamp, prev, f)
                return None, img
            return self.forward(x, w, noise_mode=noise_mode, fused_modconv=fused_modconv, gain=gain, out_split=out_split), None
        if isinstance(x, modconv.SplitActs) and (rgb is not None or not modconv.layer_supported(x, self.weight, None, noise_mode, fused_modconv, self.up)):
            x = x.dense()
        planned = modconv.take_plan(self) if modconv._plan else None
        styles, pre = planned if planned is not None else (self.affine(w), None)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if native_channels_last and fused_modconv is True and x.is_cuda and not torch.is_grad_enabled() and not modconv.is_small(x, self.up) \
                and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)      # e.g. the output of a small generic-route layer feeding a native one
        if noise_mode == 'random' and modconv.layer_supported(x, self.weight, styles, 'none', fused_modconv, self.up) and rgb is None:
            # per-image noise (device, no graph recorded): the native layer computes conv (+ FIR) alone; noise, bias and activation follow as the two
            # element-wise passes of the reference's order (:319-332) — the kernels' fused epilogue takes ONE noise image shared by the batch
            y = modconv.synthesis_layer(x, self.weight, styles, None, self.up, self.resample_filter, noise_const=None, noise_strength=None,
                                        act='linear', act_gain=1.0, clamp=None, pre=pre, rgb=None, out_split=False)
            if noise is not None:
                y = y.add_(noise.to(y.dtype))
            return bias_act.bias_act(y, self.bias.to(y.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)
        if modconv.layer_supported(x, self.weight, styles, noise_mode, fused_modconv, self.up):
            # fp16 channels-last inference: weight modulation, MFMA conv, noise, bias, activation in native kernels
            const_noise = self.use_noise and noise_mode == 'const'
            fused_rgb = None
            if rgb is not None:
                torgb, w_rgb, img = rgb[:3]
                if modconv.torgb_fusable(x, self.weight, torgb.weight, img, self.up, self.noise_const if const_noise else None, self.activation):
                    planned_rgb = modconv.take_plan(torgb) if modconv._plan else None
                    s_rgb = planned_rgb[0] if planned_rgb is not None else torgb.affine(w_rgb, out_scale=torgb.weight_gain)
                    fused_rgb = (torgb.weight, s_rgb, torgb.bias, torgb.conv_clamp, img, len(rgb) > 3 and bool(rgb[3]),      # [5]: x has no other reader
                                 planned_rgb[1] if planned_rgb is not None else None)                                        # [6]: its weights, modulated ahead
            y = modconv.synthesis_layer(x, self.weight, styles, self.bias, self.up, self.resample_filter,
                                        noise_const=self.noise_const if const_noise else None,
                                        noise_strength=self.noise_strength if const_noise else None,
                                        act=self.activation, act_gain=self.act_gain * gain, clamp=clamp, pre=pre, rgb=fused_rgb, out_split=out_split)
            return y if rgb is None else (y, fused_rgb is not None)
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1), fused_modconv=fused_modconv)
        y = bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)
        return y if rgb is None else (y, False)

    def extra_repr(self):
        return (f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}, '
                f'resolution={self.resolution:d}, up={self.up}, activation={self.activation:s}')


@persistence.persistent_class
class ToRGBLayer(torch.nn.Module):
    """1x1 modulated conv without demodulation; the weight gain rides on the styles (:342-362)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True, accumulate_into=None):
        """``accumulate_into`` (device inference): an fp32 image the native kernel may add its result to in place — the skip-image sum of
        SynthesisBlock without a separate launch.  The return value is then that image; callers check identity."""
        planned = modconv.take_plan(self) if modconv._plan else None
        styles = planned[0] if planned is not None else self.affine(w, out_scale=self.weight_gain)
        if modconv.torgb_supported(x, self.weight, styles, fused_modconv):
            out = accumulate_into if accumulate_into is not None and modconv.torgb_accumulates(x, self.weight, accumulate_into) else None
            return modconv.torgb(x, self.weight, styles, self.bias, clamp=self.conv_clamp, out=out, pre=planned[1] if planned is not None else None)      # fp32 NCHW, bias + clamp fused
        if isinstance(x, modconv.SplitActs):
            x = x.dense()
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}'


native_channels_last = True     # run device inference channels-last so every 3x3 / 1x1 layer takes the native kernels (csrc/conv2d.hip)


def _block_mode(block, ws, force_fp32, fused_modconv):
    """Shared dtype / layout / modconv-mode decision of the synthesis blocks (:421-430)."""
    if ws.device.type != 'cuda':
        force_fp32 = True
    dtype = torch.float16 if block.use_fp16 and not force_fp32 else torch.float32
    fmt = torch.channels_last if block.channels_last and not force_fp32 else torch.contiguous_format
    if fused_modconv is None:
        fused_modconv = block.fused_modconv_default
    if fused_modconv == 'inference_only':
        # the reference: fused exactly when not training (:428-429) — its grouped convolution is the slow one under autograd.  The two forms are the same
        # function; on the device a pass that records no graph (the generator passes of the D phases and of the cross-view block, loss.py:657-675, 834-836)
        # takes the fused inference kernels in training mode too
        fused_modconv = (not block.training) or native_no_grad_fused(ws)
    if native_channels_last and modconv.enabled and ws.device.type == 'cuda' and fused_modconv is True and not torch.is_grad_enabled():
        # inference on the device: channels-last is the layout the MFMA conv kernels consume (any dtype); the low-resolution
        # blocks run as batched GEMMs on plain NCHW and are left alone (a layout round trip per layer is pure launch latency)
        fmt = torch.channels_last if block.resolution ** 2 > modconv.gemm_max_pixels else torch.contiguous_format
    elif _native_training(ws):
        fmt = torch.channels_last
    return dtype, fmt, fused_modconv


no_grad_fused_in_training = os.environ.get('P3D_NO_GRAD_FUSED', '1') != '0'


def native_no_grad_fused(t):
    return no_grad_fused_in_training and modconv.enabled and native_channels_last and t.is_cuda and not torch.is_grad_enabled()


def _native_training(t):
    """Training-mode pass on the device with conv2d_gradfix's native route on: every convolution (and each of its gradients)
    consumes and produces channels-last tensors, so the blocks keep their activations that way and no layout copy sits between
    layers (the reference only does this for fp16 under ``fp16_channels_last``)."""
    return native_channels_last and t.device.type == 'cuda' and conv2d_gradfix.enabled and conv2d_gradfix.native and torch.is_grad_enabled()


@persistence.persistent_class
class SynthesisBlock(torch.nn.Module):
    """One resolution level: (up-)conv0, conv1, ToRGB with skip-image accumulation (:367-466)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2, resample_filter=resample_filter,
                                        conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    channels_last=self.channels_last, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)

    _in_div = 2          # input resolution = resolution // _in_div (the NoUp variant in superresolution.py uses 1)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, _split_ok=False, _x_dead=False, **layer_kwargs):
        """``_split_ok`` (private, set by SynthesisNetwork.forward only): the returned ``x`` may be a modconv.SplitActs that only the next block of the
        same network reads.  Every other caller — a feature extractor stepping through the blocks, a forward hook — gets the reference's contract: a
        tensor."""
        del update_emas
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        dtype, fmt, fused_modconv = _block_mode(self, ws, force_fp32, fused_modconv)
        per_layer = list(ws.unbind(dim=1))                       # one w per conv layer, then one for ToRGB
        conv_kwargs = dict(layer_kwargs, fused_modconv=fused_modconv)

        x = self._entry_features(x, ws.shape[0], dtype, fmt)
        if self.in_channels == 0:
            x = self.conv1(x, per_layer[0], **conv_kwargs)
        elif self.architecture == 'resnet':
            shortcut = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, per_layer[0], **conv_kwargs)
            x = self.conv1(x, per_layer[1], gain=np.sqrt(0.5), **conv_kwargs)
            x = shortcut.add_(x)
        rgb_done = False
        wants_rgb = self.is_last or self.architecture == 'skip'
        if self.in_channels != 0 and self.architecture != 'resnet':
            # bf16x3 device inference: x stays in the matrix cores' (hi, lo) layout between the layers whose kernels read it (modconv.SplitActs):
            # conv0's result feeds conv1; conv1's feeds this block's ToRGB and the next block's conv0 (a consumer that cannot read it calls dense())
            keep_split = (_split_ok and dtype == torch.float32 and fmt == torch.channels_last and fused_modconv is True and ws.is_cuda and not torch.is_grad_enabled()
                          and self.conv1.out_channels % 32 == 0 and layer_kwargs.get('noise_mode', 'random') != 'random' and self.img_channels > 8
                          and modconv.accepts_split_input(ws.shape[0], self.conv1.out_channels, self.resolution ** 2, 1))
            x = self.conv0(x, per_layer[0], out_split=keep_split, **conv_kwargs)
            if keep_split and _x_dead and wants_rgb and self._in_div == 2 and (img is None or img.is_contiguous(memory_format=torch.channels_last)):
                # the network's LAST block (SynthesisNetwork.forward: nobody reads its x): conv1, the wide ToRGB and the skip-image sum in one launch where
                # the kernel takes the sizes — the layer's activations are then never written (csrc/conv2d.hip: conv3x3_r2_bf16x3_kernel<TR>)
                if img is not None:
                    misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
                x, fused_img = self.conv1(x, per_layer[1], out_split=True, rgb_wide=(self.torgb, per_layer[self.num_conv], img, self.resample_filter), **conv_kwargs)
                if fused_img is not None:
                    return None, fused_img
                img_carried = False
            elif keep_split:
                x = self.conv1(x, per_layer[1], out_split=True, **conv_kwargs)
                img_carried = False
            elif wants_rgb and img is not None and self.img_channels <= 8 and x.is_cuda and not torch.is_grad_enabled():
                img = self._carry_image(img)                     # (independent of the convolutions: done first so that conv1 can add into it)
                # _x_dead (private, set by the super-resolution heads for their last block): the caller drops the returned x, so when conv1 also produces
                # the ToRGB contribution its activations are never stored and x comes back as None
                x, rgb_done = self.conv1(x, per_layer[1], rgb=(self.torgb, per_layer[self.num_conv], img, _x_dead), **conv_kwargs)
                img_carried = True
            else:
                x = self.conv1(x, per_layer[1], **conv_kwargs)
                img_carried = False
        else:
            img_carried = False

        if (wants_rgb and not rgb_done and img is not None and not img_carried and self._in_div == 2 and self.img_channels > 8 and self.img_channels % 4 == 0
                and img.is_cuda and not torch.is_grad_enabled() and fmt == torch.channels_last and img.is_contiguous(memory_format=torch.channels_last)):
            # wide skip image (the 96 tri-plane channels), device inference: ToRGB first, then its upsampled predecessor is added INTO it by the
            # upsampling launch — one pass over the image instead of three (upsample, ToRGB, add)
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            if fused_modconv is True and modconv.torgb_wide_skip_supported(x, self.torgb.weight, img, self.resample_filter):
                # split activations in: ToRGB and the skip-image sum in ONE pass over the image (csrc/torgb_split.hip)
                planned = modconv.take_plan(self.torgb) if modconv._plan else None
                s_rgb = planned[0] if planned is not None else self.torgb.affine(per_layer[self.num_conv], out_scale=self.torgb.weight_gain)
                img = modconv.torgb_wide_skip(x, self.torgb.weight, s_rgb, self.torgb.bias, self.torgb.conv_clamp, img, self.resample_filter,
                                              pre=planned[1] if planned is not None else None)
                return x, img
            y = self.torgb(x, per_layer[self.num_conv], fused_modconv=fused_modconv)
            if y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last):
                img = upfirdn2d.upsample2d_add_(y, img, self.resample_filter)
            else:
                img = self._accumulate_image(self._carry_image(img), y, fmt)
            assert x.dtype == dtype and img.dtype == torch.float32
            return x, img
        if img is not None and not img_carried:
            img = self._carry_image(img)
        if wants_rgb and not rgb_done:
            y = self.torgb(x, per_layer[self.num_conv], fused_modconv=fused_modconv, accumulate_into=img)
            img = img if y is img else self._accumulate_image(img, y, fmt)

        assert (x is None and _x_dead and rgb_done) or x.dtype == dtype
        assert img is None or img.dtype == torch.float32
        return x, img

    def _carry_image(self, img):
        """The running skip image at this block's resolution (:453-456)."""
        if self._in_div != 2:
            return img
        misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
        if img.shape[1] <= 8 and img.is_cuda and not img.is_contiguous():
            img = img.contiguous()                               # a narrow image stays NCHW: the layout the ToRGB kernels accumulate into
        return upfirdn2d.upsample2d(img, self.resample_filter)

    def _entry_features(self, x, batch, dtype, fmt):
        """The block's input activations in its working dtype / layout: the learned constant for b4, else the previous block's x."""
        if self.in_channels == 0:
            if self.const.is_cuda and not torch.is_grad_enabled():
                # device inference: the batch of constants is the same tensor every pass — built once per (batch, dtype, layout, version of the parameter);
                # nothing writes into a block's input in place (conv1 returns a new tensor), so the copy can be handed out again
                # every (batch, dtype, layout) keeps its own tensor: a captured hipGraph reads the one it was captured with, so a later call
                # with another batch must not free it.  modconv.invalidate_caches() drops them all (writes through .data bump no version).
                slot, stamp = (batch, dtype, fmt), (self.const._version, self.const.data_ptr())
                held = _const_batches.get(self)                  # (kept outside the module: nothing of it is pickled, deep-copied or moved with the module)
                if held is None or held[0] != stamp:
                    held = (stamp, {})
                    _const_batches[self] = held
                hit = held[1].get(slot)
                if hit is None:
                    if len(held[1]) >= 8:
                        held[1].clear()
                    hit = self.const.detach().to(dtype=dtype).unsqueeze(0).repeat([batch, 1, 1, 1]).contiguous(memory_format=fmt)
                    held[1][slot] = hit
                return hit
            return self.const.to(dtype=dtype).unsqueeze(0).repeat([batch, 1, 1, 1]).contiguous(memory_format=fmt)
        in_res = self.resolution // self._in_div
        misc.assert_shape(x, [None, self.in_channels, in_res, in_res])
        if isinstance(x, modconv.SplitActs):
            if dtype == torch.float32 and fmt == torch.channels_last:
                return x                                          # the previous block's conv1 left it in the layout this block's conv0 reads
            x = x.dense()
        if fmt == torch.channels_last and native_channels_last and modconv.is_small(x, self._in_div) and x.is_cuda and not torch.is_grad_enabled():
            return x.to(dtype=dtype)      # first MFMA-sized block: its x2 layer still takes the GEMM route on NCHW; conv1 converts
        return x.to(dtype=dtype, memory_format=fmt)

    def _accumulate_image(self, img, y, fmt):
        """Skip-connection image: fp32 sum of every block's ToRGB output.  A wide image (the 96-channel tri-planes) stays
        channels-last where the blocks are, so the ray-marcher reads it in place."""
        wide_cl = fmt == torch.channels_last and y.shape[1] > 8 and y.shape[1] % 4 == 0
        y = y.to(dtype=torch.float32, memory_format=torch.channels_last if wide_cl else torch.contiguous_format)
        if img is None:
            return y
        if wide_cl and not img.is_contiguous(memory_format=torch.channels_last):
            img = img.contiguous(memory_format=torch.channels_last)
        return img.add_(y)

    def extra_repr(self):
        return f'resolution={self.resolution:d}, architecture={self.architecture:s}'


_const_batches = weakref.WeakKeyDictionary()      # SynthesisBlock (b4) -> (stamp of the parameter, {(batch, dtype, layout): its learned constant repeated over the batch}): device inference only
modconv.on_invalidate(_const_batches.clear)


def prefetch_styles(blocks, block_ws, block_kwargs, ahead=False):
    """Device inference: every layer's style affine and weight modulation depend on ``ws`` alone, so they are issued up front on a
    second stream and run under the convolutions of the layers before them (they are memory-bound, the convolutions are not);
    a layer waits for its entry's event (modconv.take_plan).  Anything the plan gets wrong (a layer that ends up on another route) is
    simply recomputed.  Returns the plan's keys when a plan was made — the caller joins the side stream and drops them afterwards (finish_prefetch) — else None.
    ``ahead``: issued for a network that runs LATER in the step (the super-resolution heads, from inside the backbone's forward right after its own plan,
    ``modconv.after_prefetch``): the side stream is already ordered behind whatever made ``ws`` and nothing of the earlier plan has been released yet, so
    it neither waits for the calling stream again nor touches other networks' entries."""
    ws0 = block_ws[0]
    fused = block_kwargs.get('fused_modconv')
    own = [id(l) for b in blocks for l in (getattr(b, 'conv0', None), b.conv1, getattr(b, 'torgb', None)) if l is not None]
    for k in own:
        modconv._plan.pop(k, None)    # entries an interrupted forward left behind must never reach a layer of this one
    if not (modconv.prefetch_styles and modconv.enabled and native_channels_last and ws0.is_cuda and not torch.is_grad_enabled()
            and (fused is None or fused is True)):
        return None
    force_fp32 = bool(block_kwargs.get('force_fp32', False))
    main, side = torch.cuda.current_stream(), modconv.side_stream(ws0.device)
    if not ahead:
        side.wait_stream(main)
    keys = []
    with torch.cuda.stream(side):
        todo = []                                          # (layer, latent row, out_scale, input pixels or None for ToRGB, dtype)
        for block, cur in zip(blocks, block_ws):
            if block.fused_modconv_default is not True and fused is None and block.training and not native_no_grad_fused(ws0):
                continue
            res = block.resolution
            dtype = torch.float16 if block.use_fp16 and not force_fp32 else torch.float32
            ws_iter = iter(cur.unbind(dim=1))
            layers = [(block.conv1, res)] if block.in_channels == 0 else [(block.conv0, res // block._in_div), (block.conv1, res)]
            for layer, in_res in layers:
                todo.append((layer, next(ws_iter), 1, in_res * in_res, dtype))
            if block.is_last or block.architecture == 'skip':
                # (its modulated weights too — ('rgb', pixels) — where the layer would otherwise launch that modulation in line; premodulate_torgb)
                todo.append((block.torgb, next(ws_iter), block.torgb.weight_gain, ('rgb', res * res) if modconv.premodulate_rgb else None, dtype))
        # every style affine of the network in one launch (they are ~6 us of launch latency each on their own)
        batched = (0 < len(todo) <= modconv.FC_MAX_JOBS and len({t[1].shape[0] for t in todo}) == 1 and all(t[1].shape[1] % 4 == 0 for t in todo)   # (fc_multi takes whole float4 rows; fc() pads)
                   and all(modconv.fc_supported(w, l.affine.weight, l.affine.bias, l.affine.activation) for l, w, *_ in todo))
        all_styles = (modconv.fc_multi([(w, l.affine, sc) for l, w, sc, _, _ in todo]) if batched
                      else [l.affine(w) if sc == 1 else l.affine(w, out_scale=sc) for l, w, sc, _, _ in todo])
        # (ToRGB entries carry ('rgb', pixels): premodulate_many itself launches nothing for them — their modulations are issued together, below)
        items = [(layer.weight, styles, getattr(layer, 'up', 1), None if isinstance(in_pixels, tuple) else in_pixels, dtype) for (layer, _, _, in_pixels, dtype), styles in zip(todo, all_styles)]
        pres = modconv.premodulate_many(items)
        # one event per layer that LAUNCHED something; a layer that did not (a shared-weight layer after the first) shares the event of the last one that did, and
        # take_plan skips a position its stream already waits behind — every wait is an edge between two branches of the captured graph.  Positions number the side
        # stream's events for the life of the process (one side stream per device: a later position implies every earlier one).
        # Right behind the FIRST layer's event: every ToRGB layer's weight modulation (3-6 us each, eleven per step, otherwise in line in front of their layers) and
        # one event for all of them — the network's first layer does not stand behind them, its first ToRGB (tens of microseconds later) waits for that one event,
        # and from there on a wait is for everything issued (modconv.take_plan).
        ev, ev_seq = None, -1
        rgb_entry = {}
        for k, ((layer, _, _, in_pixels, dtype), styles, pre, fresh) in enumerate(zip(todo, all_styles, pres, modconv.premodulate_launches(items))):
            if fresh or ev is None:
                first = ev is None
                ev = torch.cuda.Event()
                ev.record(side)
                modconv._plan_seq[0] += 1
                ev_seq = modconv._plan_seq[0]
                modconv._plan_latest[:] = [ev, ev_seq]
                if first and not ahead:
                    modconv._plan_own_until[0] = modconv._plan_seq[0]
                if first:
                    rgb = [j for j, t in enumerate(todo) if isinstance(t[3], tuple)]
                    if rgb:
                        rgb_pre = [modconv.premodulate_torgb(todo[j][0].weight, all_styles[j], todo[j][3][1], todo[j][4]) for j in rgb]
                        ev_rgb = torch.cuda.Event()
                        ev_rgb.record(side)
                        modconv._plan_seq[0] += 1
                        modconv._plan_latest[:] = [ev_rgb, modconv._plan_seq[0]]
                        if not ahead:
                            modconv._plan_own_until[0] = modconv._plan_seq[0]
                        rgb_entry = {j: (p, ev_rgb, modconv._plan_seq[0]) for j, p in zip(rgb, rgb_pre)}
            if k in rgb_entry:
                modconv._plan[id(layer)] = (styles,) + rgb_entry[k]
            else:
                modconv._plan[id(layer)] = (styles, pre, ev, ev_seq)
            keys.append(id(layer))
    if not ahead:
        hooks, modconv.after_prefetch[:] = list(modconv.after_prefetch), []
        for hook in hooks:             # (networks later in the step: their plans follow this one on the side stream, before any of its tensors is released)
            hook()
    return keys


def finish_prefetch(device, keys=None):
    main = torch.cuda.current_stream()
    if not modconv.plan_joined(main):                      # (a stream that already waits behind the prefetch stream's newest event IS joined: no further edge)
        main.wait_stream(modconv.side_stream(device))
    if keys is None:
        modconv._plan.clear()
    else:
        for k in keys:
            modconv._plan.pop(k, None)


@persistence.persistent_class
class SynthesisNetwork(torch.nn.Module):
    """Stack of SynthesisBlocks b4..b{img_resolution}; ws are dealt out block by block (:471-526)."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res], w_dim=w_dim, resolution=res, img_channels=img_channels,
                                   is_last=(res == img_resolution), use_fp16=(res >= fp16_resolution), **block_kwargs)
            self.num_ws += block.num_conv
            if res == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, **block_kwargs):
        with torch.autograd.profiler.record_function('split_ws'):
            misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
            ws = ws.to(torch.float32)
            block_ws, idx = [], 0
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                block_ws.append(ws.narrow(1, idx, block.num_conv + block.num_torgb))     # ToRGB shares the next block's first w
                idx += block.num_conv
        planned = self._prefetch(block_ws, block_kwargs)
        x = img = None
        try:
            _m = torch.nn.modules.module
            blocks = [getattr(self, f'b{res}') for res in self.block_resolutions]
            for i, (block, cur) in enumerate(zip(blocks, block_ws)):
                # x is private to this loop unless somebody can observe it: a forward hook on the producing block (its output), a forward PRE-hook on
                # the consuming block (its input), or a global hook of either kind — then it stays a tensor.  Sampled per block, not per pass.
                nxt = blocks[i + 1] if i + 1 < len(blocks) else None
                hooked = bool(_m._global_forward_hooks or _m._global_forward_pre_hooks or block._forward_hooks or (nxt is not None and nxt._forward_pre_hooks))
                # the last block's x is dropped here (only img is returned): unless a hook on the block, on its conv1 or on its ToRGB could see the activations,
                # the block may leave them unwritten (SynthesisBlock.forward: _x_dead)
                dead = (nxt is None and not hooked and getattr(block, 'conv1', None) is not None and getattr(block, 'torgb', None) is not None
                        and not (block.conv1._forward_hooks or block.conv1._forward_pre_hooks or block.torgb._forward_hooks or block.torgb._forward_pre_hooks))
                x, img = block(x, img, cur, _split_ok=not hooked, _x_dead=dead, **block_kwargs)
        finally:
            if planned is not None:
                finish_prefetch(ws.device, planned)
        return img

    def _prefetch(self, block_ws, block_kwargs):
        return prefetch_styles([getattr(self, f'b{res}') for res in self.block_resolutions], block_ws, block_kwargs)

    def extra_repr(self):
        return (f'w_dim={self.w_dim:d}, num_ws={self.num_ws:d}, img_resolution={self.img_resolution:d}, '
                f'img_channels={self.img_channels:d}, num_fp16_res={self.num_fp16_res:d}')


@persistence.persistent_class
class Generator(torch.nn.Module):
    """mapping + synthesis (:531-554)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = synthesis.num_ws               # one w per modulated layer, which only the synthesis network can count
        self.synthesis = synthesis
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        return self.synthesis(self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas),
                              update_emas=update_emas, **synthesis_kwargs)


@persistence.persistent_class
class DiscriminatorBlock(torch.nn.Module):
    """fromrgb (+skip) -> conv0 -> down-2 conv1, 'resnet' adds a down-2 1x1 branch (:559-643)."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, architecture='resnet',
                 activation='lrelu', resample_filter=[1, 3, 3, 1], conv_clamp=None, use_fp16=False, fp16_channels_last=False, freeze_layers=0):
        assert in_channels in [0, tmp_channels]
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.resolution, self.img_channels, self.first_layer_idx = in_channels, resolution, img_channels, first_layer_idx
        self.architecture, self.use_fp16 = architecture, use_fp16
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.num_layers = 0                          # layers created so far; layer k of the whole network is trainable iff k >= freeze_layers

        class _Flags:
            def __next__(flags):
                flag = (self.first_layer_idx + self.num_layers) >= freeze_layers
                self.num_layers += 1
                return flag
        trainable = _Flags()
        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation, trainable=next(trainable),
                                       conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation, trainable=next(trainable),
                                 conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2, trainable=next(trainable),
                                 resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=self.channels_last)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, trainable=next(trainable),
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x, img, force_fp32=False):
        probe = img if x is None else x
        dtype, fmt = self._working_format(probe, force_fp32)
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=fmt)
        if self.in_channels == 0 or self.architecture == 'skip':            # this block reads the image
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            img = img.to(dtype=dtype, memory_format=fmt)
            feats = self.fromrgb(img)
            x = feats if x is None else x + feats
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            shortcut = self.skip(x, gain=np.sqrt(0.5))
            x = shortcut.add_(self.conv1(self.conv0(x), gain=np.sqrt(0.5)))
        else:
            x = self.conv1(self.conv0(x))
        assert x.dtype == dtype
        return x, img

    def _working_format(self, probe, force_fp32):
        """dtype / memory format of this block (:625-628) — fp32 NCHW off the device; channels-last wherever the native convolution
        kernels run (inference above the GEMM-route size, and every training pass)."""
        force_fp32 = force_fp32 or probe.device.type != 'cuda'
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if native_channels_last and modconv.enabled and probe.is_cuda and not torch.is_grad_enabled() and self.resolution ** 2 > modconv.gemm_max_pixels:
            fmt = torch.channels_last
        elif _native_training(probe):
            fmt = torch.channels_last
        return dtype, fmt

    def extra_repr(self):
        return 'resolution={:d}, architecture={:s}'.format(self.resolution, self.architecture)


@persistence.persistent_class
class MinibatchStdLayer(torch.nn.Module):
    """Appends the per-group feature std as extra channel(s) (:648-672)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        n, c, h, w = x.shape
        with misc.suppress_tracer_warnings():
            g = torch.min(torch.as_tensor(self.group_size), torch.as_tensor(n)) if self.group_size is not None else n
        f = self.num_channels
        y = x.reshape(g, -1, f, c // f, h, w)
        y = y - y.mean(dim=0)
        y = (y.square().mean(dim=0) + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1).repeat(g, 1, h, w)
        return torch.cat([x, y], dim=1)

    def extra_repr(self):
        return f'group_size={self.group_size}, num_channels={self.num_channels:d}'


@persistence.persistent_class
class DiscriminatorEpilogue(torch.nn.Module):
    """4x4 tail: mbstd -> conv -> fc -> out, projected on the label embedding when cmap_dim > 0 (:677-733)."""

    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4,
                 mbstd_num_channels=1, activation='lrelu', conv_clamp=None):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.cmap_dim, self.resolution, self.img_channels, self.architecture = in_channels, cmap_dim, resolution, img_channels, architecture
        out_features = cmap_dim if cmap_dim > 0 else 1          # projection discriminator: dotted with the label embedding afterwards
        self.mbstd = None
        if mbstd_num_channels > 0:
            self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels)
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, out_features)

    def forward(self, x, img, cmap, force_fp32=False):
        del force_fp32                                           # the 4x4 tail always runs in fp32
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        assert x.dtype == torch.float32
        return x

    def extra_repr(self):
        return 'resolution={:d}, architecture={:s}'.format(self.resolution, self.architecture)


@persistence.persistent_class
class Discriminator(torch.nn.Module):
    """StyleGAN2 discriminator b{res}..b8 + b4 epilogue (:738-797)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=4,
                 conv_clamp=256, cmap_dim=None, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}):
        super().__init__()
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        if cmap_dim is None:
            cmap_dim = channels[4]
        if c_dim == 0:
            cmap_dim = 0
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        cur = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(channels[res] if res < img_resolution else 0, channels[res], channels[res // 2], resolution=res,
                                       first_layer_idx=cur, use_fp16=(res >= fp16_resolution), **block_kwargs, **common)
            setattr(self, f'b{res}', block)
            cur += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(channels[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **common)

    def forward(self, img, c, update_emas=False, **block_kwargs):
        _ = update_emas
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        cmap = self.mapping(None, c) if self.c_dim > 0 else None
        return self.b4(x, img, cmap)

    def extra_repr(self):
        return f'c_dim={self.c_dim:d}, img_resolution={self.img_resolution:d}, img_channels={self.img_channels:d}'
