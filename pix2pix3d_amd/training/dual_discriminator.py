"""Discriminators of the tri-plane GAN.  Mirror of training/dual_discriminator.py (line refs are to that file): same
class names, constructor arguments and parameter names.  ``DualDiscriminator`` sees the super-resolved image
concatenated with the bilinearly (anti-aliased) up-sized raw neural rendering (:157-172)."""
import os

import numpy as np
import torch

from ..torch_utils import persistence
from ..torch_utils.ops import upfirdn2d
from .networks_stylegan2 import DiscriminatorBlock, MappingNetwork, DiscriminatorEpilogue


def _build_pyramid(self, c_dim, img_resolution, img_channels, architecture, channel_base, channel_max, num_fp16_res, conv_clamp, cmap_dim,
                   block_kwargs, mapping_kwargs, epilogue_kwargs):
    """The StyleGAN2 discriminator body shared by every variant (:40-64, :125-155): b{res}..b8 blocks, label mapping, b4 epilogue."""
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


def _run_pyramid(self, img, cmap_input, block_kwargs):
    x = None
    for res in self.block_resolutions:
        x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
    cmap = self.mapping(None, cmap_input) if self.c_dim > 0 else None
    return self.b4(x, img, cmap)


@persistence.persistent_class
class SingleDiscriminator(torch.nn.Module):
    """Plain StyleGAN2 discriminator on ``img['image']`` (:20-82)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=4,
                 conv_clamp=256, cmap_dim=None, sr_upsample_factor=1, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}):
        super().__init__()
        _build_pyramid(self, c_dim, img_resolution, img_channels, architecture, channel_base, channel_max, num_fp16_res, conv_clamp, cmap_dim,
                       block_kwargs, mapping_kwargs, epilogue_kwargs)

    def forward(self, img, c, update_emas=False, **block_kwargs):
        _ = update_emas
        return _run_pyramid(self, img['image'], c, block_kwargs)

    def extra_repr(self):
        return f'c_dim={self.c_dim:d}, img_resolution={self.img_resolution:d}, img_channels={self.img_channels:d}'


native_upsize = os.environ.get('P3D_NATIVE_UPSIZE', '1') != '0'
_tri_filters = {}


def bilinear_upsize(x, s):
    """Bilinear up-sizing by an even integer factor (align_corners=False) on the native upfirdn2d kernels — forward, gradient and double backward (R1
    differentiates the discriminator's input resize twice).  Anti-aliasing does nothing when up-sizing (the filter support is max(1, in/out) = 1), so this
    IS ``interpolate(mode='bilinear', antialias=True)`` of the reference's filtered_resizing for the raw image (dual_discriminator.py:86-90, 166) to fp32
    rounding: zero-insertion by s, then the separable triangle of 2 s taps ((k + 0.5) / s), the clamped border rows realised by one replicated pixel.
    ATen's anti-aliased backward kernel took 1.0 ms per call for this 128^2 -> 512^2 resize (profiles/round4_a_train_kernel_stats.csv)."""
    assert s in (2, 4, 8)
    key = (s, x.device)
    f = _tri_filters.get(key)
    if f is None:
        taps = [(k + 0.5) / s for k in range(s)]
        f = _tri_filters[key] = upfirdn2d.setup_filter(taps + taps[::-1], device=x.device)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode='replicate')
    p0, p1 = s // 2 - 1, -(s // 2)
    return upfirdn2d.upfirdn2d(xp, f, up=s, padding=[p0, p1, p0, p1], gain=s * s)


def filtered_resizing(image_orig_tensor, size, f, filter_mode='antialiased'):
    """Resize the raw rendering to the discriminator resolution (:86-102)."""
    interp = lambda t, s, aa=False: torch.nn.functional.interpolate(t, size=(s, s), mode='bilinear', align_corners=False, antialias=aa)
    if filter_mode == 'antialiased':
        h, w = image_orig_tensor.shape[-2:]
        if native_upsize and image_orig_tensor.is_cuda and h == w and size in (2 * h, 4 * h, 8 * h) and image_orig_tensor.dtype == torch.float32:
            return bilinear_upsize(image_orig_tensor, size // h)
        return interp(image_orig_tensor, size, True)
    if filter_mode == 'classic':
        t = upfirdn2d.upsample2d(image_orig_tensor, f, up=2)
        t = interp(t, size * 2 + 2)
        return upfirdn2d.downsample2d(t, f, down=2, flip_filter=True, padding=-1)
    if filter_mode == 'none':
        return interp(image_orig_tensor, size)
    if type(filter_mode) == float:
        assert 0 < filter_mode < 1
        return (1 - filter_mode) * interp(image_orig_tensor, size) + filter_mode * interp(image_orig_tensor, size, True)
    raise ValueError(f'unknown filter_mode {filter_mode!r}')


@persistence.persistent_class
class DualDiscriminator(torch.nn.Module):
    """Discriminator on cat(image, resized image_raw): twice the image channels (:106-175)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=4,
                 conv_clamp=256, cmap_dim=None, disc_c_noise=0, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}, **unused_kwargs):
        super().__init__()
        _build_pyramid(self, c_dim, img_resolution, img_channels * 2, architecture, channel_base, channel_max, num_fp16_res, conv_clamp, cmap_dim,
                       block_kwargs, mapping_kwargs, epilogue_kwargs)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.disc_c_noise = disc_c_noise

    def forward(self, img, c, update_emas=False, **block_kwargs):
        _ = update_emas
        raw = filtered_resizing(img['image_raw'], size=img['image'].shape[-1], f=self.resample_filter)
        x = torch.cat([img['image'], raw], 1)
        if self.c_dim > 0 and self.disc_c_noise > 0:
            c += torch.randn_like(c) * c.std(0) * self.disc_c_noise
        return _run_pyramid(self, x, c, block_kwargs)

    def extra_repr(self):
        return f'c_dim={self.c_dim:d}, img_resolution={self.img_resolution:d}, img_channels={self.img_channels:d}'


@persistence.persistent_class
class DummyDualDiscriminator(torch.nn.Module):
    """Same input plumbing as DualDiscriminator but the raw branch is zeroed (:179-248)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=4,
                 conv_clamp=256, cmap_dim=None, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}):
        super().__init__()
        _build_pyramid(self, c_dim, img_resolution, img_channels * 2, architecture, channel_base, channel_max, num_fp16_res, conv_clamp, cmap_dim,
                       block_kwargs, mapping_kwargs, epilogue_kwargs)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.raw_fade = 1

    def forward(self, img, c, update_emas=False, **block_kwargs):
        _ = update_emas
        self.raw_fade = max(0, self.raw_fade - 1 / (500000 / 32))
        size = img['image'].shape[-1]
        raw = torch.nn.functional.interpolate(img['image_raw'], size=(size, size), mode='bilinear', align_corners=False, antialias=True) * self.raw_fade
        return _run_pyramid(self, torch.cat([img['image'], raw], 1), c, block_kwargs)

    def extra_repr(self):
        return f'c_dim={self.c_dim:d}, img_resolution={self.img_resolution:d}, img_channels={self.img_channels:d}'
