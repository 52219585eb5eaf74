"""Conditional tri-plane generators of pix2pix3D: label-map / edge-map encoders, the conditional mapping
networks and the generators whose ``mapping`` / ``synthesis`` / ``sample`` / ``sample_mixed`` / ``forward``
the training loop and the applications call.

Mirror of training/triplane_cond.py (line refs are to that file).  Same class names, constructor
arguments, attribute and parameter names; ``synthesis()`` = ray sampler -> StyleGAN2 backbone ->
fused tri-plane ray-marcher -> super-resolution head(s).
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import dnnlib
from ..torch_utils import misc
from ..torch_utils import persistence
from ..torch_utils.ops import conv2d_gradfix
from .networks_stylegan2 import SynthesisNetwork, FullyConnectedLayer, normalize_2nd_moment, DiscriminatorBlock, track_w_avg, truncate_ws
from .networks_stylegan2 import Generator as StyleGAN2Backbone
from .triplane import OSGDecoder, _osg_mlp, _TriPlaneCore, frozen_pass
from .volumetric_rendering.renderer import ImportanceSemanticRenderer
from .volumetric_rendering.ray_sampler import RaySampler


@persistence.persistent_class
class EqualConv2d(torch.nn.Module):
    """conv2d with the 1/sqrt(fan_in) scale applied at run time (:29-62)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = torch.nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        co, ci, kh, kw = self.weight.shape
        if self.padding == 0 and tuple(input.shape[2:]) == (kh, kw):
            # the Encoder's 4x4 projector sees a 4x4 image: a plain matrix product (and no trip through the vendor conv library)
            flat_x, flat_w = input.reshape(input.shape[0], ci * kh * kw), (self.weight * self.scale).reshape(co, ci * kh * kw)
            if input.is_cuda and conv2d_gradfix.enabled and conv2d_gradfix.native and torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
                y = conv2d_gradfix.conv2d(flat_x[:, :, None, None], flat_w[:, :, None, None])[:, :, 0, 0]      # training: native kernels, every gradient order
            else:
                y = flat_x @ flat_w.t()
            if self.bias is not None:
                y = y + self.bias
            return y.reshape(input.shape[0], co, 1, 1)
        return F.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]}, {self.weight.shape[2]}, '
                f'stride={self.stride}, padding={self.padding})')


@persistence.persistent_class
class Encoder(torch.nn.Module):
    """Discriminator-style pyramid down to 4x4 then a 4x4 projection to ``n_latents`` w vectors (:65-196).

    Every branch of the original that CAN execute is here.  Its progressive-growing code calls two functions that exist nowhere in the reference
    (``downsample`` at :160, :163 and ``camera_9d_to_16d`` at :183 — neither defined nor imported: a NameError there), so: ``progressive`` works with the
    full pyramid (alpha = -1) and with a low-resolution head at alpha = 0 fed an image already at that resolution; an input that would have to be
    down-sized, a blend (0 < alpha < 1) and ``predict_camera`` raise NotImplementedError here, where the reference raises NameError."""

    def __init__(self, img_resolution, img_channels, bottleneck_factor=2, architecture='resnet', channel_base=1, channel_max=512,
                 num_fp16_res=0, conv_clamp=None, lowres_head=None, block_kwargs={}, model_kwargs={}, upsample_type='default',
                 progressive=False, **unused):
        super().__init__()
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, bottleneck_factor, -1)]
        self.architecture, self.lowres_head, self.upsample_type, self.progressive = architecture, lowres_head, upsample_type, progressive
        self.model_kwargs = model_kwargs
        self.output_mode = model_kwargs.get('output_mode', 'styles')
        if self.progressive:
            assert self.architecture == 'skip', 'not supporting other types for now.'
        self.predict_camera = model_kwargs.get('predict_camera', False)
        channel_base = int(channel_base * 32768)
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        cur = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(channels[res] if res < img_resolution else 0, channels[res], channels[res // 2], resolution=res,
                                       first_layer_idx=cur, use_fp16=(res >= fp16_resolution), **block_kwargs, **common)
            setattr(self, f'b{res}', block)
            cur += block.num_layers
        if self.output_mode not in ['W', 'W+', 'None']:
            raise NotImplementedError
        self.num_ws = model_kwargs.get('num_ws', 0)
        self.n_latents = self.num_ws if self.output_mode == 'W+' else (0 if self.output_mode == 'None' else 1)
        self.w_dim = model_kwargs.get('w_dim', 512)
        self.add_dim = model_kwargs.get('add_dim', 0) if not self.predict_camera else 9
        self.out_dim = self.w_dim * self.n_latents + self.add_dim
        assert self.out_dim > 0, 'output dimenstion has to be larger than 0'
        assert self.block_resolutions[-1] // 2 == 4, 'make sure the last resolution is 4x4'
        self.projector = EqualConv2d(channels[4], self.out_dim, 4, padding=0, bias=False)
        self.register_buffer('alpha', torch.scalar_tensor(-1))

    def set_resolution(self, res):
        self.curr_status = res                       # (n_levels, _, before_res, target_res) of the progressive schedule (:131)

    def set_alpha(self, alpha):
        if alpha is None:
            return
        self.alpha.fill_(alpha)

    def get_block_resolutions(self, input_img):
        """Which blocks run, the blend weight and the head resolution (:133-153)."""
        block_resolutions, lowres_head, alpha = self.block_resolutions, self.lowres_head, self.alpha
        if self.progressive and (self.lowres_head is not None) and (self.alpha > -1):
            if 0 < self.alpha < 1:
                try:
                    n_levels, _, before_res, target_res = self.curr_status
                    alpha, index = math.modf(self.alpha * n_levels)
                except Exception:                    # no schedule set: the original falls back to the input's own size
                    before_res = target_res = input_img.size(-1)
                if before_res == target_res:
                    alpha = 0                        # the generator did not upsample either: no blend
                block_resolutions = [res for res in self.block_resolutions if res <= target_res]
                lowres_head = before_res
            elif self.alpha == 0:
                block_resolutions = [res for res in self.block_resolutions if res <= lowres_head]
        return block_resolutions, alpha, lowres_head

    def forward(self, inputs, **block_kwargs):
        img = inputs['img'] if isinstance(inputs, dict) else inputs
        block_resolutions, alpha, lowres_head = self.get_block_resolutions(img)
        blending = self.progressive and (self.lowres_head is not None) and (-1 < self.alpha < 1) and (alpha > 0)
        if img.size(-1) > block_resolutions[0] or blending:
            raise NotImplementedError('Encoder: this call needs the `downsample` function of the original (triplane_cond.py:160-163), which the reference never '
                                      'defines (it raises NameError there); feed the image at the resolution of the first active block')
        if self.predict_camera:
            raise NotImplementedError('Encoder: predict_camera needs `camera_9d_to_16d` (triplane_cond.py:183), which the reference never defines')
        assert img.size(-1) == block_resolutions[0], 'Encoder: input must already be at the first active block\'s resolution'
        # progressive growing entered below the top of the pyramid: the first active block sees fromrgb(img) as its feature input (:165-166)
        x = None if (not self.progressive) or (block_resolutions[0] == self.img_resolution) else getattr(self, f'b{block_resolutions[0]}').fromrgb(img)
        for res in block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        out = self.projector(x)[:, :, 0, 0]
        if self.output_mode == 'W+':
            out = out.reshape(out.shape[0], self.num_ws, self.w_dim)
        elif self.output_mode == 'W':
            out = out.unsqueeze(1).expand(-1, self.num_ws, -1)
        else:
            out = None
        return {'ws': out}


class _EntangledMapping(torch.nn.Module):
    """Shared body of the two earlier conditional mapping networks (:201-296, :403-494): the conditioning image becomes ONE embedding
    (Encoder in 'W' mode) that is concatenated with z (and the embedded camera label) in front of the MLP; every w is the same."""
    _encoder_name = None

    def _setup(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features,
               activation, lr_multiplier, w_avg_beta):
        self.z_dim, self.c_dim, self.in_resolution, self.in_channels = z_dim, c_dim, in_resolution, in_channels
        self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = w_dim, num_ws, num_layers, w_avg_beta
        embed_features = w_dim if embed_features is None else embed_features
        layer_features = w_dim if layer_features is None else layer_features
        sizes = [z_dim + embed_features * (2 if c_dim > 0 else 1)] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        setattr(self, self._encoder_name, Encoder(img_resolution=in_resolution, img_channels=in_channels,
                                                  model_kwargs={'num_ws': 1, 'w_dim': embed_features, 'output_mode': 'W'}))
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(sizes[idx], sizes[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def _condition_image(self, batch):
        raise NotImplementedError

    def forward(self, z=None, c=None, batch=None, truncation_psi=1, truncation_cutoff=None, update_emas=False, **unused_kwargs):
        """MLP over cat(normalised z, normalised image embedding, normalised camera embedding); the same w for every layer."""
        feats = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            feats.append(normalize_2nd_moment(z.to(torch.float32)).contiguous())
        cond = self._condition_image(batch)
        misc.assert_shape(cond, [None, self.in_channels, self.in_resolution, self.in_resolution])
        emb = getattr(self, self._encoder_name)(cond.to(torch.float32))['ws'].squeeze(1)
        misc.assert_shape(emb, [None, self.w_dim])
        feats.append(normalize_2nd_moment(emb).contiguous())
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            feats.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        x = torch.cat(feats, dim=1) if len(feats) > 1 else feats[0]
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if update_emas:
            track_w_avg(self, x)
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        return truncate_ws(self, x, truncation_psi, truncation_cutoff)


@persistence.persistent_class
class MaskMappingNetwork(_EntangledMapping):
    """Label map -> one-hot -> a single embedding mixed into every w (:201-296)."""
    _encoder_name = 'embed_mask'

    def __init__(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, one_hot=True, **unused):
        super().__init__()
        self.one_hot = one_hot
        self._setup(z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features, activation, lr_multiplier, w_avg_beta)

    def _condition_image(self, batch):
        if self.one_hot:
            return torch.nn.functional.one_hot(batch['mask'].squeeze(1).long(), self.in_channels).permute(0, 3, 1, 2)
        return batch['mask']


@persistence.persistent_class
class EdgeMappingNetwork(_EntangledMapping):
    """Edge image -> a single embedding mixed into every w (:403-494)."""
    _encoder_name = 'embed_edge'

    def __init__(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, **unused):
        super().__init__()
        self._setup(z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features, activation, lr_multiplier, w_avg_beta)

    def _condition_image(self, batch):
        return batch['mask'].to(torch.float32)


class _DisentangledMapping(torch.nn.Module):
    """Shared body of the two conditional mapping networks (:301-399, :499-592): the first ``geometry_layer``
    (= 7) ws come from the conditioning image through the Encoder, the remaining ones from z (and the
    embedded camera label) through the usual 8-layer MLP."""

    def _setup(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features,
               activation, lr_multiplier, w_avg_beta):
        self.z_dim, self.c_dim, self.in_resolution, self.in_channels = z_dim, c_dim, in_resolution, in_channels
        self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = w_dim, num_ws, num_layers, w_avg_beta
        self.geometry_layer = 7
        embed_features = w_dim if embed_features is None else embed_features
        layer_features = w_dim if layer_features is None else layer_features
        sizes = [z_dim + (embed_features if c_dim > 0 else 0)] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        self.embed_mask = Encoder(img_resolution=in_resolution, img_channels=in_channels,
                                  model_kwargs={'num_ws': self.geometry_layer, 'w_dim': w_dim, 'output_mode': 'W+'})
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(sizes[idx], sizes[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([num_ws, w_dim]))

    def _condition_image(self, batch, n):
        raise NotImplementedError

    def forward(self, z=None, c=None, batch=None, truncation_psi=1, truncation_cutoff=None, update_emas=False, **unused_kwargs):
        """ws = [geometry ws from the conditioning image (Encoder, W+ mode)] ++ [the z / camera MLP's w, repeated]."""
        feats = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            feats.append(normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            feats.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        x = torch.cat(feats, dim=1) if len(feats) > 1 else feats[0]
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)

        batch_size = z.shape[0]
        cond = self._condition_image(batch, batch_size)
        misc.assert_shape(cond, [batch_size, self.in_channels, self.in_resolution, self.in_resolution])
        geometry = self.embed_mask(cond.to(torch.float32))['ws']
        misc.assert_shape(geometry, [None, self.geometry_layer, self.w_dim])
        if self.num_ws is not None:
            appearance = x.unsqueeze(1).repeat([1, self.num_ws - self.geometry_layer, 1])
            x = torch.cat([geometry, appearance], dim=1)
        if update_emas:
            track_w_avg(self, x)
        return truncate_ws(self, x, truncation_psi, truncation_cutoff)


@persistence.persistent_class
class MaskMappingNetwork_disentangle(_DisentangledMapping):
    """Label-map conditioning: integer mask -> one-hot -> Encoder (:301-399)."""

    def __init__(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, one_hot=True, **unused):
        super().__init__()
        self.one_hot = one_hot
        self._setup(z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features, activation, lr_multiplier, w_avg_beta)

    def _condition_image(self, batch, n):
        misc.assert_shape(batch['mask'], [n, 1, None, None])
        if self.one_hot:
            return torch.nn.functional.one_hot(batch['mask'].squeeze(1).long(), self.in_channels).permute(0, 3, 1, 2)
        return batch['mask']


@persistence.persistent_class
class EdgeMappingNetwork_disentangle(_DisentangledMapping):
    """Edge-map conditioning: the float edge image goes to the Encoder as is (:499-592)."""

    def __init__(self, z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, **unused):
        super().__init__()
        self._setup(z_dim, c_dim, in_resolution, in_channels, w_dim, num_ws, num_layers, embed_features, layer_features, activation, lr_multiplier, w_avg_beta)

    def _condition_image(self, batch, n):
        return batch['mask'].to(torch.float32)


@persistence.persistent_class
class Generator_cond(torch.nn.Module):
    """StyleGAN2 backbone whose mapping network is chosen by name through ``mapping_kwargs['class_name']`` (:596-621)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = synthesis.num_ws
        self.synthesis = synthesis
        self.mapping = dnnlib.util.construct_class_by_name(**mapping_kwargs, z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)


class OSGDecoder_semantic(torch.nn.Module):
    """Label decoder of the two-backbone generator (:859-887): one 32-64-33 MLP on the plane-averaged semantic features; channel 0 is
    the density, the rest the labels — raw logits, or clamped-sigmoid when ``options['sigmoid']`` (single-channel edge maps)."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])
        self.final_sigmoid = options['sigmoid']

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)
        n, m, c = x.shape
        y = self.net(x.reshape(n * m, c)).reshape(n, m, -1)
        labels = y[..., 1:]
        if self.final_sigmoid:
            labels = torch.sigmoid(labels) * (1 + 2 * 0.001) - 0.001
        return {'rgb': labels, 'sigma': y[..., 0:1]}


class OSGDecoder_semantic_entangle(torch.nn.Module):
    """ONE 32-64-(1 + decoder_output_dim) MLP whose outputs are split by position (:891-924): density, 3 colour channels (clamped sigmoid),
    ``semantic_channels`` raw label logits, and the remaining feature channels (clamped sigmoid) — or clamped sigmoid on everything when
    ``options['sigmoid']``.  No shipped configuration selects it (train.py builds the lateSeparate decoder); kept for checkpoints that name it.
    The fused ray-marcher does not implement this split, so renders with it take the tensor-op formulation."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])
        self.feature_sigmoid = options['sigmoid']
        self.semantic_channels = options['semantic_channels']

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)
        n, m, c = x.shape
        y = self.net(x.reshape(n * m, c)).reshape(n, m, -1)
        squash = lambda t: torch.sigmoid(t) * (1 + 2 * 0.001) - 0.001
        if self.feature_sigmoid:
            feature = squash(y[..., 1:])
        else:
            k = 4 + self.semantic_channels
            feature = torch.cat((squash(y[..., 1:4]), y[..., 4:k], squash(y[..., k:])), dim=-1)
        return {'rgb': feature, 'sigma': y[..., 0:1]}


class OSGDecoder_semantic_lateSeparate(torch.nn.Module):
    """Two independent 32-64-33 MLPs on the plane-averaged feature: a colour net and a label net; the density is
    channel 0 of the LABEL net (:926-970).  Output 'rgb' = cat(32 colour channels, 32 label channels)."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])
        self.net_semantic = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])
        self.semantic_sigmoid = options['sigmoid']

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)
        n, m, c = x.shape
        flat = x.reshape(n * m, c)
        colour = self.net(flat).reshape(n, m, -1)
        label = self.net_semantic(flat).reshape(n, m, -1)
        squash = lambda t: torch.sigmoid(t) * (1 + 2 * 0.001) - 0.001
        rgb = squash(colour[..., 1:])
        sem = squash(label[..., 1:]) if self.semantic_sigmoid else label[..., 1:]
        return {'rgb': torch.cat((rgb, sem), dim=-1), 'sigma': label[..., 0:1]}


class _TriPlaneBase(_TriPlaneCore):
    """The conditional generators' entry points: ``mapping`` / ``sample`` / ``forward`` take the data batch (label map + pose)."""
    _backbone_class = None      # Generator_cond, set below the class

    @frozen_pass
    def mapping(self, z, c, batch, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), batch, truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    def sample(self, coordinates, directions, z, c, batch, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """Colour features + density at arbitrary 3-D points (shape extraction)."""
        ws = self.mapping(z, batch['pose'], batch, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, update_emas=update_emas, **synthesis_kwargs)

    def forward(self, z, c, batch, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, batch['pose'], batch, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)


_TriPlaneBase._backbone_class = Generator_cond


@persistence.persistent_class
class TriPlaneGenerator(_TriPlaneBase):
    """Image-only conditional generator: one OSG decoder, one SR head (:626-715)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={}, rendering_kwargs={},
                 sr_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self._init_common(z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, rendering_kwargs, synthesis_kwargs)
        self.superresolution = dnnlib.util.construct_class_by_name(class_name=rendering_kwargs['superresolution_module'], channels=32,
                                                                   img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                                                   sr_antialias=rendering_kwargs['sr_antialias'], **sr_kwargs)
        self.decoder = OSGDecoder(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32})
        self._finish_init(rendering_kwargs)

    @frozen_pass
    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        feature_image, depth_image = self._render(ws, c, neural_rendering_resolution, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs, heads=(self.superresolution,))
        rgb_image = feature_image[:, :3]
        sr_image = self.superresolution(rgb_image, feature_image, ws, **self._sr_kwargs(synthesis_kwargs))
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image}


@persistence.persistent_class
class TriPlaneSemanticEntangleGenerator(_TriPlaneBase):
    """The generator train.py selects (train.py:374-380): shared planes, two-net decoder, colour and label SR heads (:975-1079)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, semantic_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, data_type=None, **synthesis_kwargs):
        super().__init__()
        self._init_common(z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, rendering_kwargs, synthesis_kwargs)
        self.semantic_channels, self.data_type = semantic_channels, data_type
        sr_common = dict(channels=32, img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res, sr_antialias=rendering_kwargs['sr_antialias'])
        self.superresolution = dnnlib.util.construct_class_by_name(class_name=rendering_kwargs['superresolution_module'], **sr_common, **sr_kwargs)
        self.superresolution_semantic = dnnlib.util.construct_class_by_name(class_name=rendering_kwargs['superresolution_module_semantic'],
                                                                            semantic_channels=semantic_channels, **sr_common, **sr_kwargs)
        self.decoder = OSGDecoder_semantic_lateSeparate(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32,
                                                             'sigmoid': semantic_channels == 1, 'semantic_channels': semantic_channels})
        self._finish_init(rendering_kwargs)

    @frozen_pass
    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        feature_image, depth_image = self._render(ws, c, neural_rendering_resolution, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs,
                                                  heads=(self.superresolution, self.superresolution_semantic))
        half = feature_image.shape[1] // 2
        rgb_feat, sem_feat = feature_image[:, :half], feature_image[:, half:]
        sr_kw = self._sr_kwargs(synthesis_kwargs)
        rgb_image = rgb_feat[:, :3]
        semantic_image = sem_feat[:, :self.semantic_channels]
        sr_image = self.superresolution(rgb_image, rgb_feat, ws, **sr_kw)
        sr_semantic = self.superresolution_semantic(semantic_image, sem_feat, ws, **sr_kw)
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image, 'semantic': sr_semantic, 'semantic_raw': semantic_image}


def _split_planes(planes, n_planes, channels):
    return planes.view(len(planes), n_planes, channels, planes.shape[-2], planes.shape[-1])


@persistence.persistent_class
class TriPlaneSemanticGenerator(_TriPlaneBase):
    """The two-backbone generator (:722-853): texture planes from an unconditional StyleGAN2 generator on z, semantic planes from a
    conditional one on the label map alone (z_dim = 0); ``ws`` carries both latents side by side ([N, num_ws, 2*w_dim]).  Rendering is
    ``ImportanceSemanticRenderer``'s tensor-op formulation (train.py no longer selects this class); backbones and SR heads run on the
    native layers."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, semantic_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, data_type=None, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.semantic_channels, self.data_type = semantic_channels, data_type
        self.renderer = ImportanceSemanticRenderer()
        self.ray_sampler = RaySampler()
        self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=256, img_channels=32 * 3, mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        self.backbone_semantic = Generator_cond(0, c_dim, w_dim, img_resolution=256, img_channels=32 * 3, mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        sr_common = dict(channels=32, img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res, sr_antialias=rendering_kwargs['sr_antialias'])
        self.superresolution = dnnlib.util.construct_class_by_name(class_name=rendering_kwargs['superresolution_module'], **sr_common, **sr_kwargs)
        self.superresolution_semantic = dnnlib.util.construct_class_by_name(class_name=rendering_kwargs['superresolution_module_semantic'],
                                                                            semantic_channels=semantic_channels, **sr_common, **sr_kwargs)
        lr_mul = rendering_kwargs.get('decoder_lr_mul', 1)
        self.decoder = OSGDecoder(64, {'decoder_lr_mul': lr_mul, 'decoder_output_dim': 32, 'sigmoid': True})
        self.decoder_semantic = OSGDecoder_semantic(32, {'decoder_lr_mul': lr_mul, 'decoder_output_dim': 32, 'sigmoid': semantic_channels == 1})
        self._finish_init(rendering_kwargs)

    @frozen_pass
    def mapping(self, z, c, batch, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        c = c * self.rendering_kwargs.get('c_scale', 0)
        kw = dict(truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return torch.cat([self.backbone.mapping(z, c, **kw), self.backbone_semantic.mapping(None, c, batch, **kw)], dim=-1)

    def _both_planes(self, ws, update_emas, synthesis_kwargs):
        assert ws.shape[-1] == self.w_dim * 2
        ws_texture, ws_semantic = ws[..., :self.w_dim], ws[..., self.w_dim:]
        tex = self.backbone.synthesis(ws_texture, update_emas=update_emas, **synthesis_kwargs)
        sem = self.backbone_semantic.synthesis(ws_semantic, update_emas=update_emas, **synthesis_kwargs)
        return _split_planes(tex, 3, 32), _split_planes(sem, 3, 32), ws_texture, ws_semantic

    @frozen_pass
    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        cam2world, intrinsics = c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        ray_o, ray_d = self.ray_sampler(cam2world, intrinsics, neural_rendering_resolution)
        n = ray_o.shape[0]
        tex, sem, ws_texture, ws_semantic = self._both_planes(ws, update_emas, synthesis_kwargs)
        feat, depth, _ = self.renderer(tex, sem, self.decoder, self.decoder_semantic, ray_o, ray_d, self.rendering_kwargs)
        r = self.neural_rendering_resolution
        feature_image = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], r, r).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, r, r)
        half = feature_image.shape[1] // 2
        rgb_feat, sem_feat = feature_image[:, :half], feature_image[:, half:]
        sr_kw = self._sr_kwargs(synthesis_kwargs)
        rgb_image = rgb_feat[:, :3]
        sr_image = self.superresolution(rgb_image, rgb_feat, ws_texture, **sr_kw)
        semantic_image = sem_feat[:, :self.semantic_channels]
        sr_semantic = self.superresolution_semantic(semantic_image, sem_feat, ws_semantic, **sr_kw)
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image, 'semantic': sr_semantic, 'semantic_raw': semantic_image}

    @frozen_pass
    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        tex, sem, _, _ = self._both_planes(ws, update_emas, synthesis_kwargs)
        return self.renderer.run_model(tex, sem, self.decoder, self.decoder_semantic, coordinates, directions, self.rendering_kwargs)


@persistence.persistent_class
class TriPlaneSemanticEntangleGenerator_withBG(TriPlaneSemanticEntangleGenerator):
    """``TriPlaneSemanticEntangleGenerator`` plus a background: a second, unconditional StyleGAN2 synthesis network driven by the last w
    paints a 64-channel panorama that is looked up by ray direction and composited behind the rendered foreground with the rays'
    leftover transmittance (:1084-1246).  The foreground is the fused ray-marcher; the panorama lookup is per ray (M per image) and
    stays a tensor op."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, semantic_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, data_type=None, **synthesis_kwargs):
        super().__init__(z_dim, c_dim, w_dim, img_resolution, img_channels, semantic_channels, sr_num_fp16_res=sr_num_fp16_res,
                         mapping_kwargs=mapping_kwargs, rendering_kwargs=rendering_kwargs, sr_kwargs=sr_kwargs, data_type=data_type, **synthesis_kwargs)
        mapping_bg_kwargs = dict(mapping_kwargs)
        mapping_bg_kwargs['class_name'] = None
        self.backbone_bg = StyleGAN2Backbone(z_dim, 0, w_dim, img_resolution=256, img_channels=32 * 2, mapping_kwargs=mapping_bg_kwargs, **synthesis_kwargs)

    @frozen_pass
    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        cam2world, intrinsics = c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        ray_o, ray_d = self.ray_sampler(cam2world, intrinsics, neural_rendering_resolution)
        n = ray_o.shape[0]
        planes = self._planes(ws, update_emas, synthesis_kwargs, cache_backbone, use_cached_backbone)
        feat, depth, wsum = self.renderer(planes, self.decoder, ray_o, ray_d, self.rendering_kwargs)
        ws_bg = ws[:, -1, :].unsqueeze(1).repeat([1, ws.shape[1], 1])
        planes_bg = self.backbone_bg.synthesis(ws_bg, update_emas=update_emas, **synthesis_kwargs)
        planes_bg = planes_bg.view(len(planes_bg), 64, planes_bg.shape[-2], planes_bg.shape[-1])
        feat, depth = self.combine_fg_bg(feat, depth, wsum, planes_bg, ray_o, ray_d, self.rendering_kwargs)
        r = self.neural_rendering_resolution
        feature_image = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], r, r).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, r, r)
        weight_image = wsum.permute(0, 2, 1).reshape(n, 1, r, r)
        half = feature_image.shape[1] // 2
        rgb_feat, sem_feat = feature_image[:, :half], feature_image[:, half:]
        sr_kw = self._sr_kwargs(synthesis_kwargs)
        rgb_image = rgb_feat[:, :3]
        sr_image = self.superresolution(rgb_image, rgb_feat, ws, **sr_kw)
        semantic_image = sem_feat[:, :self.semantic_channels]
        sr_semantic = self.superresolution_semantic(semantic_image, sem_feat, ws, **sr_kw)
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image, 'semantic': sr_semantic, 'semantic_raw': semantic_image,
                'weight': weight_image}

    def combine_fg_bg(self, feature_samples, depth_samples, weights_samples, planes_bg, ray_origins, ray_directions, rendering_kwargs):
        """[N,M,64] foreground + (1 - accumulated weight) x panorama(direction) (:1202-1246): the direction's azimuth / polar angle index
        the panorama; its colour half is squashed to [-1, 1], its label half to [-10, 10], and for label maps the background is pinned to
        class 0 (logit 20, the others 0); depth gets ``ray_end`` behind the foreground."""
        d = ray_directions / torch.norm(ray_directions, dim=-1, keepdim=True)
        theta = torch.atan2(d[:, :, 1], d[:, :, 0])
        phi = torch.acos(d[:, :, 2])
        grid = torch.stack([theta * 2 / np.pi, phi * 2 / np.pi - 1], dim=-1).unsqueeze(1)
        bg = F.grid_sample(planes_bg.float(), grid, mode='bilinear', padding_mode='border', align_corners=False).squeeze(2).permute(0, 2, 1)
        assert bg.shape == feature_samples.shape
        bg = (torch.sigmoid(bg) * (1 + 2 * 0.001) - 0.001) * 2 - 1
        scale = torch.ones(bg.shape[-1], device=bg.device)
        scale[32:] = 10
        bg = bg * scale
        if self.semantic_channels > 1:
            bg = bg.clone()
            bg[:, :, 32 + 1:32 + self.semantic_channels] = 0
            bg[:, :, 32] = 20
        feature_samples = feature_samples + bg * (1 - weights_samples)
        depth_samples = depth_samples + rendering_kwargs['ray_end'] * (1 - weights_samples)
        return feature_samples, depth_samples
