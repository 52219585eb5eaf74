"""Super-resolution heads: two StyleGAN2 synthesis blocks that lift the 128^2 (or 64^2) neural rendering to the
output resolution.  Mirror of training/superresolution.py (line refs are to that file); same class and
parameter names."""
import os

import torch

from ..torch_utils import persistence
from .networks_stylegan2 import SynthesisBlock, prefetch_styles, finish_prefetch


@persistence.persistent_class
class SynthesisBlockNoUp(SynthesisBlock):
    """SynthesisBlock whose conv0 keeps the resolution and whose skip image is not upsampled (:191-290)."""
    _in_div = 1

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, **kwargs):
        super().__init__(in_channels, out_channels, w_dim, resolution, img_channels, is_last, **kwargs)
        if in_channels != 0:
            self.conv0.up = 1                      # same weights/shapes as the x2 layer, no resampling


skip_dead_x = os.environ.get('P3D_SR_SKIP_DEAD_X', '1') != '0'      # the last block's activations are not stored when only its fused ToRGB reads them (0: A/B runs)


class _SuperresolutionBase(torch.nn.Module):
    """ws[:, -1] feeds all three layers of both blocks; inputs are resized to ``input_resolution`` first (:312-323)."""
    input_resolution = 128
    _resize_only_up = False          # SuperresolutionHybrid4X resizes only inputs SMALLER than its input resolution (:80); the others any mismatch

    def _prep(self, rgb, x):
        if (x.shape[-1] < self.input_resolution) if self._resize_only_up else (x.shape[-1] != self.input_resolution):
            size = (self.input_resolution, self.input_resolution)
            x = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
            rgb = torch.nn.functional.interpolate(rgb, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
        return rgb, x

    def prefetch_ahead(self, ws, **block_kwargs):
        """Device inference: this head's style affines and weight modulations, issued NOW on the prefetch stream for a forward that comes later in the step
        (they depend on ``ws`` alone; triplane._render has the backbone's forward call this right after its own plan).  ``forward(…, ws)`` with the same
        tensor picks the plan up; any other call drops it."""
        from ..torch_utils.ops import modconv
        if not (ws.is_cuda and not torch.is_grad_enabled()):
            return
        row = ws[:, -1:, :].expand(-1, 3, -1)
        keys = prefetch_styles([self.block0, self.block1], [row, row], block_kwargs, ahead=True)
        if keys is not None:
            modconv._ahead[id(self)] = (ws, keys)

    def forward(self, rgb, x, ws, **block_kwargs):
        from ..torch_utils.ops import modconv
        ahead = modconv._ahead.pop(id(self), None)
        if ahead is not None and ahead[0] is not ws:             # planned for another latent tensor: not ours
            for k in ahead[1]:
                modconv._plan.pop(k, None)
            ahead = None
        if ws.is_cuda and not torch.is_grad_enabled():
            ws = ws[:, -1:, :].expand(-1, 3, -1)                  # device inference: the three layers read the SAME row in place (a stride-0 view; repeat() is three launches per head)
        else:
            ws = ws[:, -1:, :].repeat(1, 3, 1)
        rgb, x = self._prep(rgb, x)
        planned = ahead[1] if ahead is not None else prefetch_styles([self.block0, self.block1], [ws, ws], block_kwargs)
        try:
            x, rgb = self.block0(x, rgb, ws, **block_kwargs)
            # block1's x is returned to nobody (:297-354 of the reference return rgb only): unless somebody hooked the block to look at it, its last
            # layer need not store it (networks_stylegan2.SynthesisBlock.forward: _x_dead)
            # (a hook on the block OR on any layer inside it — a feature extractor on block1.conv1 must still be handed the activations)
            gm = torch.nn.modules.module
            hooked = bool(gm._global_forward_hooks or getattr(gm, '_global_forward_hooks_always_called', None)) or any(m._forward_hooks for m in self.block1.modules())
            x, rgb = self.block1(x, rgb, ws, _x_dead=skip_dead_x and not hooked, **block_kwargs)
        finally:
            if planned is not None:
                finish_prefetch(ws.device, planned)
        return rgb


sr_channels_last = True      # fp16 SR blocks run channels_last (the reference's `fp16_channels_last` knob, off there by default):
                             # the layout the MFMA conv kernels of csrc/conv2d.hip consume

def _two_blocks(self, cls0, cls1, channels, mid, out, res0, res1, img_channels, use_fp16, block_kwargs):
    clamp = 256 if use_fp16 else None
    block_kwargs = dict(block_kwargs)
    block_kwargs.setdefault('fp16_channels_last', sr_channels_last)
    self.block0 = cls0(channels, mid, w_dim=512, resolution=res0, img_channels=img_channels, is_last=False, use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)
    self.block1 = cls1(mid, out, w_dim=512, resolution=res1, img_channels=img_channels, is_last=True, use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)


@persistence.persistent_class
class SuperresolutionHybrid8XDC(_SuperresolutionBase):
    """128^2 -> 512^2, channels 32 -> 256 -> 128 (:297-323)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        _two_blocks(self, SynthesisBlock, SynthesisBlock, channels, 256, 128, 256, 512, 3, sr_num_fp16_res > 0, block_kwargs)


@persistence.persistent_class
class SuperresolutionHybrid8XDC_semantic(_SuperresolutionBase):
    """Label-map twin of 8XDC: ToRGB emits ``semantic_channels`` (:328-354)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, semantic_channels, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        _two_blocks(self, SynthesisBlock, SynthesisBlock, channels, 256, 128, 256, 512, semantic_channels, sr_num_fp16_res > 0, block_kwargs)


@persistence.persistent_class
class SuperresolutionHybrid8X(_SuperresolutionBase):
    """128^2 -> 512^2, channels 32 -> 128 -> 64 (:28-56)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        _two_blocks(self, SynthesisBlock, SynthesisBlock, channels, 128, 64, 256, 512, 3, sr_num_fp16_res > 0, block_kwargs)
        from ..torch_utils.ops import upfirdn2d
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybrid4X(_SuperresolutionBase):
    """128^2 -> 256^2: a no-upsampling block then a x2 block (:62-88)."""
    _resize_only_up = True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 256
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        _two_blocks(self, SynthesisBlockNoUp, SynthesisBlock, channels, 128, 64, 128, 256, 3, sr_num_fp16_res > 0, block_kwargs)
        from ..torch_utils.ops import upfirdn2d
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybrid2X(_SuperresolutionBase):
    """64^2 -> 128^2 (ShapeNet cars): a no-upsampling block then a x2 block (:94-121)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 128
        self.input_resolution, self.sr_antialias = 64, sr_antialias
        _two_blocks(self, SynthesisBlockNoUp, SynthesisBlock, channels, 128, 64, 64, 128, 3, sr_num_fp16_res > 0, block_kwargs)
        from ..torch_utils.ops import upfirdn2d
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybrid2X_semantic(_SuperresolutionBase):
    """Label-map twin of 2X (:127-154)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, semantic_channels, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 128
        self.input_resolution, self.sr_antialias = 64, sr_antialias
        _two_blocks(self, SynthesisBlockNoUp, SynthesisBlock, channels, 128, 64, 64, 128, semantic_channels, sr_num_fp16_res > 0, block_kwargs)
        from ..torch_utils.ops import upfirdn2d
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybridDeepfp32(_SuperresolutionBase):
    """128^2 -> 256^2 head of the old 256x256 EG3D models (:160-188, "here for backwards compatibility"): a no-upsampling block then a x2
    block like 4X, but without the ``sr_antialias`` argument — inputs smaller than 128^2 are enlarged with plain bilinear interpolation.
    No generator of this repository's configurations selects it; it is reachable by name from old checkpoints."""
    _resize_only_up = True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, num_fp16_res=4, conv_clamp=None, channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 256
        self.input_resolution, self.sr_antialias = 128, False
        _two_blocks(self, SynthesisBlockNoUp, SynthesisBlock, channels, 128, 64, 128, 256, 3, sr_num_fp16_res > 0, block_kwargs)
        from ..torch_utils.ops import upfirdn2d
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
