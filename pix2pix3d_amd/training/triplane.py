"""EG3D tri-plane generator and decoder (mirror of training/triplane.py; line refs are to that file), plus the core that the
conditional generators of ``triplane_cond`` share with it: camera split, backbone with the one-slot plane cache, fused
ray-marcher, point queries.  pix2pix3D's train.py selects the conditional generators (train.py:374-380); the unconditional
``TriPlaneGenerator`` is what EG3D checkpoints (``afhqcats512-128.pkl``) hold, so ``legacy.load_network_pkl`` resolves to it."""
import functools
import os

import torch

from .. import dnnlib
from ..torch_utils import persistence
from ..torch_utils.ops import modconv
from .networks_stylegan2 import FullyConnectedLayer, Generator as StyleGAN2Backbone
from .volumetric_rendering.renderer import ImportanceRenderer
from .volumetric_rendering.ray_sampler import RaySampler


frozen_passes_without_graph = os.environ.get('P3D_FROZEN_NO_GRAD', '1') != '0'
frozen_passes_exact_fp32 = os.environ.get('P3D_FROZEN_EXACT_FP32', '1') != '0'      # graph-less passes of a generator in TRAINING mode keep the arithmetic of its differentiated passes
train_products_bf16x3 = os.environ.get('P3D_TRAIN_G_BF16X3', '0') == '1'     # opt-in: the GENERATOR's fp32 training convolutions (forward + data gradient; the label-map
                                                                             # Encoder included) as bf16x3 — the arithmetic its inference passes use — while the
                                                                             # discriminators keep exact fp32 products (conv2d_gradfix.products)


def _tensors_of(values):
    for v in values:
        if isinstance(v, torch.Tensor):
            yield v
        elif isinstance(v, dict):
            yield from _tensors_of(v.values())


def _with_exact_products(method, self, args, kwargs):
    from ..torch_utils.ops import modconv
    from .volumetric_rendering import renderer as rmod
    prev = (modconv.split_bf16, rmod.mlp_bf16x3)
    modconv.split_bf16, rmod.mlp_bf16x3 = False, False
    try:
        return method(self, *args, **kwargs)
    finally:
        modconv.split_bf16, rmod.mlp_bf16x3 = prev


def frozen_pass(method):
    """Device passes through a generator that CANNOT record a graph — grad mode is on, but neither an argument nor a parameter requires a gradient: the
    generator passes of the discriminator phases (training_loop.py:516 leaves ``requires_grad`` on for the phase's own network only; loss.py:834-836,
    903-905 call run_G without no_grad) — run under ``torch.no_grad()``.  Same values and the same (graph-less) outputs; what changes is that the layers
    below see what they key their inference kernels on.  Those kernels default to bf16x3 products (three bf16 MFMAs per fp32 product) while the
    differentiated passes of training form exact fp32 products (training_loop.py:278-280): a generator in TRAINING mode therefore runs its frozen passes
    with the bf16x3 switches off — the discriminators are trained on fakes from the same function the generator is optimised through — unless the
    generator's training products are bf16x3 themselves (``train_products_bf16x3``).  (The parameter scan below is per call on purpose: a cached answer
    would go stale under ``p.requires_grad_()`` on a single parameter and silently drop that parameter's gradient.)"""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        exact = self.training and frozen_passes_exact_fp32 and not train_products_bf16x3
        if frozen_passes_without_graph and torch.is_grad_enabled():
            tensors = list(_tensors_of(list(args) + list(kwargs.values())))
            if tensors and all(t.is_cuda for t in tensors) and not any(t.requires_grad for t in tensors) and not any(p.requires_grad for p in self.parameters()):
                with torch.no_grad():
                    return _with_exact_products(method, self, args, kwargs) if exact else method(self, *args, **kwargs)
        elif exact and not torch.is_grad_enabled():                  # the loss's own no_grad passes of a training-mode generator (the cross-view block, loss.py:690-741)
            return _with_exact_products(method, self, args, kwargs)
        if train_products_bf16x3 and torch.is_grad_enabled():
            from ..torch_utils.ops import conv2d_gradfix
            with conv2d_gradfix.products(True):
                return method(self, *args, **kwargs)
        return method(self, *args, **kwargs)
    return wrapper


def _osg_mlp(n_features, hidden, out_dim, lr_mul):
    return torch.nn.Sequential(FullyConnectedLayer(n_features, hidden, lr_multiplier=lr_mul), torch.nn.Softplus(),
                               FullyConnectedLayer(hidden, out_dim, lr_multiplier=lr_mul))


class OSGDecoder(torch.nn.Module):
    """mean over planes -> FC(32,64) -> softplus -> FC(64, 1+C): density = channel 0, colour = clamped sigmoid of the rest (:112-135)."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)                      # [N, M, C]
        n, m, c = x.shape
        y = self.net(x.reshape(n * m, c)).reshape(n, m, -1)
        rgb = torch.sigmoid(y[..., 1:]) * (1 + 2 * 0.001) - 0.001      # MipNeRF-style sigmoid clamping
        return {'rgb': rgb, 'sigma': y[..., 0:1]}


class _TriPlaneCore(torch.nn.Module):
    """What every tri-plane generator shares.  Subclasses set ``_backbone_class`` (the StyleGAN2 generator that makes the planes),
    build ``self.decoder`` / the SR head(s) and define ``mapping`` / ``synthesis`` / ``sample`` / ``forward``."""
    _backbone_class = None

    def _init_common(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, rendering_kwargs, synthesis_kwargs):
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.renderer = ImportanceRenderer()
        self.ray_sampler = RaySampler()
        self.backbone = self._backbone_class(z_dim, c_dim, w_dim, img_resolution=256, img_channels=32 * 3, mapping_kwargs=mapping_kwargs, **synthesis_kwargs)

    def _finish_init(self, rendering_kwargs):
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self._last_planes = None

    def _planes(self, ws, update_emas, synthesis_kwargs, cache_backbone=False, use_cached_backbone=False):
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        return planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])

    def _render(self, ws, c, neural_rendering_resolution, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs, heads=()):
        """Planes -> fused ray-marcher.  ``heads``: the super-resolution modules the caller runs on the result with THIS ``ws`` — on the device their style affines
        and weight modulations are issued by the backbone's forward right behind its own (``prefetch_ahead``), under the backbone's launch-bound first layers."""
        cam2world = c[:, :16].view(-1, 4, 4)
        intrinsics = c[:, 16:25].view(-1, 3, 3)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        ray_o, ray_d = self.ray_sampler(cam2world, intrinsics, neural_rendering_resolution)
        n = ray_o.shape[0]
        if heads and modconv.sr_prefetch_ahead and ws.is_cuda and not torch.is_grad_enabled():
            sr_kw = self._sr_kwargs(synthesis_kwargs)
            modconv.after_prefetch.append(lambda: [h.prefetch_ahead(ws, **sr_kw) for h in heads if hasattr(h, 'prefetch_ahead')])
        try:
            planes = self._planes(ws, update_emas, synthesis_kwargs, cache_backbone, use_cached_backbone)
        finally:
            modconv.after_prefetch.clear()                         # (a backbone that made no plan never ran it)
        feat, depth, _ = self.renderer(planes, self.decoder, ray_o, ray_d, self.rendering_kwargs)
        r = self.neural_rendering_resolution
        if feat.is_cuda and not torch.is_grad_enabled():
            # the fused renderer's [N, rays, C] IS the channels-last image: the SR heads (channels-last, fp16) convert their half of it in the
            # one pass they need anyway, instead of after an NCHW copy of all of it.  Same values, same shape; only the strides differ.
            feature_image = feat.reshape(n, r, r, feat.shape[-1]).permute(0, 3, 1, 2)
        else:
            feature_image = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], r, r).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, r, r)
        return feature_image, depth_image

    def _sr_kwargs(self, synthesis_kwargs):
        kw = {k: v for k, v in synthesis_kwargs.items() if k != 'noise_mode'}
        return dict(noise_mode=self.rendering_kwargs['superresolution_noise_mode'], **kw)

    @frozen_pass
    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """Colour features + density at arbitrary 3-D points for given latents (shape extraction, density regularisation)."""
        planes = self._planes(ws, update_emas, synthesis_kwargs)
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)


@persistence.persistent_class
class TriPlaneGenerator(_TriPlaneCore):
    """EG3D's unconditional generator (:18-107): StyleGAN2 mapping on (z, camera label), one OSG decoder, one SR head."""
    _backbone_class = StyleGAN2Backbone

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
    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    @frozen_pass
    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        feature_image, depth_image = self._render(ws, c, neural_rendering_resolution, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs)
        rgb_image = feature_image[:, :3]
        sr_image = self.superresolution(rgb_image, feature_image, ws, **self._sr_kwargs(synthesis_kwargs))
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image}

    def sample(self, coordinates, directions, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, update_emas=update_emas, **synthesis_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)
