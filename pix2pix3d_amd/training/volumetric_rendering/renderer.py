"""Tri-plane importance renderer.

Host-side mirror of training/volumetric_rendering/renderer.py:88-253 (``ImportanceRenderer``): same
constructor, ``forward(planes, decoder, ray_origins, ray_directions, rendering_options)`` and
``run_model(...)`` signatures, same ``rendering_options`` keys, same RNG draws in the same order.

Device tensors run ONE fused HIP kernel (csrc/render.hip through ``p3d_render_forward`` /
``p3d_sample_points``); graphs that need gradients get the same forward and a fused backward
(csrc/render_bwd.hip through ``p3d_render_backward``, see ``_FusedRenderFn``).  CPU tensors run the
tensor-op restatement below (``_forward_tensor_ops``), which is also what the reference does on every
device.  ``fused_policy = 'require'`` turns any silent use of the tensor-op path on a device tensor
into an error.
"""
import ctypes
import os

import torch

from ... import _lib
from . import math_utils
from .ray_marcher import MipRayMarcher2

fused_policy = 'auto'          # 'auto' | 'require' | 'never'
_warned_routes = set()


def _warn_tensor_op_route(who, reason):
    """A device tensor is about to take the tensor-op formulation instead of the fused kernel: say so, once per reason — a silent
    fallback would look like the native path in every output but the profile.  (Expected cases: a gradient w.r.t. rays, depth or point coordinates, density_noise > 0, more than 64 coarse / fine samples per ray, a decoder that is not the OSG 32-64-33 MLP.)"""
    key = (who, reason)
    if key not in _warned_routes:
        _warned_routes.add(key)
        import warnings
        warnings.warn(f'{who}: device tensors on the tensor-op renderer, not the fused HIP kernel: {reason}', RuntimeWarning, stacklevel=3)
fused_training = True          # graphs that need gradients: fused forward + recompute-in-backward (see _FusedRenderFn)
fused_backward = True          # ... with the backward on the device kernels of csrc/render_bwd.hip (False: replay the tensor-op renderer)
mlp_bf16x3 = os.environ.get('P3D_MLP_BF16X3', '1') != '0'      # inference: the decoder MLPs as three bf16 MFMAs per fp32 product (csrc/render_device.h)
mlp_l1x6 = os.environ.get('P3D_MLP_L1X6', '1') != '0'          # exact forward passes (training, bf16x3 off), with modconv.f32_x6: layer 1 of the decoder MLPs as bf16x6 (fp32-accurate)


class _RenderDesc(ctypes.Structure):          # p3d_render_desc (include/p3d_hip.h)
    _fields_ = [('n_img', ctypes.c_int32), ('rays_per_img', ctypes.c_int32), ('plane_h', ctypes.c_int32), ('plane_w', ctypes.c_int32),
                ('n_nets', ctypes.c_int32), ('semantic_sigmoid', ctypes.c_int32), ('depth_resolution', ctypes.c_int32),
                ('depth_resolution_importance', ctypes.c_int32), ('disparity_space_sampling', ctypes.c_int32), ('white_back', ctypes.c_int32),
                ('ray_start', ctypes.c_float), ('ray_end', ctypes.c_float), ('box_warp', ctypes.c_float),
                ('image_stride', ctypes.c_int64), ('plane_stride', ctypes.c_int64), ('pixel_stride', ctypes.c_int64), ('raster_order', ctypes.c_int32),
                ('mlp_bf16x3', ctypes.c_int32)]


_vp, _i32, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
_lib.register('p3d_render_decoder_floats', ctypes.c_int, [])
_lib.register('p3d_planes_to_channels_last', ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp])
_lib.register('p3d_pack_decoder', ctypes.c_int, [_vp] * 8 + [_i32, _f32, _vp, _vp])
_lib.register('p3d_pack_decoder_bf16x3', ctypes.c_int, [_vp] * 8 + [_i32, _f32, _vp, _vp])
_lib.register('p3d_pack_decoder_l1x6', ctypes.c_int, [_vp] * 8 + [_i32, _f32, _vp, _vp])
_lib.register('p3d_render_bwd_decoder_floats', ctypes.c_int, [])
_lib.register('p3d_render_grad_decoder_floats', ctypes.c_int, [])
_lib.register('p3d_pack_decoder_bwd', ctypes.c_int, [_vp] * 4 + [ctypes.c_int32, ctypes.c_float, _vp, _vp])
_lib.register('p3d_render_backward', ctypes.c_int, [_vp] * 9 + [ctypes.POINTER(_RenderDesc)] + [_vp] * 6 + [_vp])
_lib.register('p3d_render_forward', ctypes.c_int, [_vp] * 8 + [ctypes.POINTER(_RenderDesc)] + [_vp] * 6 + [_vp])
_lib.register('p3d_render_forward_debug', ctypes.c_int, [_vp] * 8 + [ctypes.POINTER(_RenderDesc)] + [_vp] * 7 + [_vp])
_lib.register('p3d_sample_points', ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(_RenderDesc), _i32, _vp, _vp, _vp])
_lib.register('p3d_sample_points_backward', ctypes.c_int, [_vp] * 4 + [ctypes.POINTER(_RenderDesc), _i32] + [_vp] * 4 + [_vp])
_lib.register('p3d_importance_sample', ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp])
_lib.register('p3d_importance_sample_index', ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp])
_lib.register('p3d_render_decoder_floats_dual', ctypes.c_int, [])
_lib.register('p3d_pack_decoder_dual', ctypes.c_int, [_vp] * 8 + [_f32, _vp, _vp])
_lib.register('p3d_render_forward_dual', ctypes.c_int, [_vp] * 9 + [ctypes.POINTER(_RenderDesc)] + [_vp] * 4 + [_vp])
_lib.register('p3d_sample_points_dual', ctypes.c_int, [_vp] * 4 + [ctypes.POINTER(_RenderDesc), _i32, _vp, _vp, _vp])


def generate_planes():
    """Axes of the three feature planes (renderer.py:23-37): rows of each 3x3 are the plane's basis vectors."""
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


def project_onto_planes(planes, coordinates):
    """coordinates [N,M,3] -> in-plane (u, v) per plane, [N*n_planes, M, 2] (renderer.py:39-53)."""
    n, m, _ = coordinates.shape
    k = planes.shape[0]
    inv = torch.linalg.inv(planes)                                     # [k,3,3]
    uvw = torch.einsum('nmc,kcd->nkmd', coordinates, inv)              # row vector times inverse basis
    return uvw.reshape(n * k, m, 3)[..., :2]


def sample_from_planes(plane_axes, plane_features, coordinates, mode='bilinear', padding_mode='zeros', box_warp=None):
    """Bilinear taps of every plane at the projected points: [N, n_planes, M, C] (renderer.py:55-65)."""
    assert padding_mode == 'zeros'
    n, k, c, h, w = plane_features.shape
    m = coordinates.shape[1]
    uv = project_onto_planes(plane_axes, (2 / box_warp) * coordinates).unsqueeze(1)
    out = torch.nn.functional.grid_sample(plane_features.reshape(n * k, c, h, w), uv.float(), mode=mode,
                                          padding_mode=padding_mode, align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(n, k, m, c)


def sample_from_3dgrid(grid, coordinates):
    """Trilinear lookup in a dense feature volume (renderer.py:67-80; not called by any generator, kept for the module's surface):
    grid [1 or N, C, H, W, D], coordinates [N, P, 3] in [-1, 1] -> [N, P, C]."""
    n, _, dims = coordinates.shape
    out = torch.nn.functional.grid_sample(grid.expand(n, -1, -1, -1, -1), coordinates.reshape(n, 1, 1, -1, dims),
                                          mode='bilinear', padding_mode='zeros', align_corners=False)
    n, c, h, w, d = out.shape
    return out.permute(0, 4, 3, 2, 1).reshape(n, h * w * d, c)


def _decoder_nets(decoder):
    """Recognise the OSG decoders the fused kernel implements; returns (nets, lr_mul-free raw params, sigmoid flag) or None.

    OSGDecoder (training/triplane.py:112-135): one 32->64->33 MLP.  OSGDecoder_semantic_lateSeparate
    (training/triplane_cond.py:926-970): colour net + label net, density from the label net."""
    nets = []
    for name in ('net', 'net_semantic'):
        seq = getattr(decoder, name, None)
        if seq is None:
            continue
        if not (isinstance(seq, torch.nn.Sequential) and len(seq) == 3 and isinstance(seq[1], torch.nn.Softplus)):
            return None
        fc1, fc2 = seq[0], seq[2]
        ok = all(hasattr(fc, 'weight_gain') and hasattr(fc, 'bias_gain') and getattr(fc, 'activation', None) == 'linear' and fc.bias is not None for fc in (fc1, fc2))
        if not ok or tuple(fc1.weight.shape) != (64, 32) or tuple(fc2.weight.shape) != (33, 64):
            return None
        if seq[1].beta != 1 or seq[1].threshold != 20:
            return None
        nets.append((fc1, fc2))
    if len(nets) not in (1, 2) or (len(nets) == 2 and not hasattr(decoder, 'semantic_sigmoid')):
        return None
    if len(nets) == 1 and type(decoder).__name__ != 'OSGDecoder':
        return None                                  # e.g. OSGDecoder_semantic slices its outputs differently
    lr = float(nets[0][0].bias_gain)
    for fc1, fc2 in nets:
        if abs(fc1.weight_gain - lr / 32 ** 0.5) > 1e-12 or abs(fc2.weight_gain - lr / 64 ** 0.5) > 1e-12 or fc1.bias_gain != lr or fc2.bias_gain != lr:
            return None
    return nets, lr, bool(getattr(decoder, 'semantic_sigmoid', False))


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def pack_decoder(decoder_info, device, bf16x3):
    """The decoder's FullyConnectedLayer parameters in the kernels' LDS image (p3d_pack_decoder / _bf16x3 / _l1x6 for ``bf16x3`` = 0 / 1 / 2), on the current stream."""
    nets, lr_mul, _ = decoder_info
    lib = _lib.lib()
    packed = torch.empty([lib.p3d_render_decoder_floats()], dtype=torch.float32, device=device)
    ws = []
    for fc1, fc2 in nets:
        ws += [_f32c(fc1.weight), _f32c(fc1.bias), _f32c(fc2.weight), _f32c(fc2.bias)]
    ptrs = [_lib.ptr(t) for t in ws] + [None] * (8 - len(ws))
    pack = (lib.p3d_pack_decoder, lib.p3d_pack_decoder_bf16x3, lib.p3d_pack_decoder_l1x6)[int(bf16x3)]
    _lib.check(pack(*ptrs, len(nets), lr_mul, _lib.ptr(packed), _lib.stream_of(packed)), 'pack_decoder')
    return packed, ws


class _FusedContext:
    """Device-side operands shared by the fused entry points: channels-last planes + packed decoder."""

    def __init__(self, planes, decoder_info, bf16x3=False, packed=None):
        nets, lr_mul, sem_sigmoid = decoder_info
        self.bf16x3 = int(bf16x3)                               # p3d_render_desc.mlp_bf16x3: 0 exact, 1 bf16x3, 2 layer 1 as bf16x6
        lib = _lib.lib()
        n, k, c, h, w = planes.shape
        assert k == 3 and c == 32
        self.strides = (0, 0, 0)
        st = planes.stride()
        if planes.dtype == torch.float32 and st[2] == 1 and st[1] == 32 and st[4] >= 96 and st[3] == w * st[4] and st[0] == h * st[3] and st[4] % 4 == 0 \
                and planes.data_ptr() % 16 == 0:
            # already texel-major (a channels-last [N,96,H,W] backbone output viewed as [N,3,32,H,W]): read it in place
            self.planes_cl = planes.detach()
            self.strides = (st[0], st[1], st[4])
            src = self.planes_cl
        else:
            src = _f32c(planes)
            self.planes_cl = torch.empty([n, 3, h, w, 32], dtype=torch.float32, device=planes.device)
            _lib.check(lib.p3d_planes_to_channels_last(_lib.ptr(src), _lib.ptr(self.planes_cl), n, h, w, _lib.stream_of(src)), 'planes_to_channels_last')
        if packed is not None:                                 # (packed by pack_decoder() ahead of time, same arithmetic)
            self.packed, self._keep = packed
        else:
            self.packed, self._keep = pack_decoder(decoder_info, planes.device, self.bf16x3)
        self.n_nets, self.sem_sigmoid, self.n, self.h, self.w = len(nets), sem_sigmoid, n, h, w

    def desc(self, options, rays_per_img=1, start=0.0, end=0.0, raster=False):
        return _RenderDesc(self.n, rays_per_img, self.h, self.w, self.n_nets, int(self.sem_sigmoid),
                           int(options.get('depth_resolution', 0)), int(options.get('depth_resolution_importance', 0)),
                           int(bool(options.get('disparity_space_sampling', False))), int(bool(options.get('white_back', False))),
                           float(start), float(end), float(options['box_warp']), *self.strides, int(raster), int(self.bf16x3))


class ImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()

    # ------------------------------------------------------------------------------------------------
    def _fused_reason(self, planes, decoder, options, needs_grad, trainable=True):
        """None when the fused kernel applies, else why not."""
        if fused_policy == 'never':
            return 'fused_policy == never'
        if planes.device.type != 'cuda':
            return 'CPU tensors'
        if needs_grad and not (fused_training and trainable):
            return 'autograd graph requested (no fused backward for this entry point / fused_training is off)'
        if planes.ndim != 5 or planes.shape[1] != 3 or planes.shape[2] != 32:
            return f'planes shape {tuple(planes.shape)} is not [N,3,32,H,W]'
        if options.get('density_noise', 0) > 0:
            return 'density_noise > 0'
        if options.get('clamp_mode', 'softplus') != 'softplus':
            return "clamp_mode != 'softplus' (the tensor-op route raises the reference's assertion, ray_marcher.py:35)"
        if _decoder_nets(decoder) is None:
            return f'decoder {type(decoder).__name__} is not an OSG 32-64-33 decoder'
        return None

    def _tensor_op_guard(self, planes, reason):
        if planes.device.type != 'cuda' or fused_policy == 'never':
            return
        if fused_policy == 'require':
            raise RuntimeError(f'ImportanceRenderer: fused HIP path required but unavailable: {reason}')
        _warn_tensor_op_route(type(self).__name__, reason)

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        self.plane_axes = self.plane_axes.to(ray_origins.device)
        needs_grad = torch.is_grad_enabled() and (planes.requires_grad or ray_origins.requires_grad or ray_directions.requires_grad
                                                  or any(p.requires_grad for p in decoder.parameters()))
        reason = self._fused_reason(planes, decoder, rendering_options, needs_grad)
        if reason is None:
            out = self._forward_fused(planes, decoder, ray_origins, ray_directions, rendering_options, needs_grad)
            if out is not None:
                return out
            reason = 'sample counts outside the fused kernel envelope (<= 64 coarse, 1..64 fine)'
        self._tensor_op_guard(planes, reason)
        return self._forward_tensor_ops(planes, decoder, ray_origins, ray_directions, rendering_options)

    # ------------------------------------------------------------------------------------------------
    def _ray_limits(self, ray_origins, ray_directions, opt):
        """('auto' branch, renderer.py:91-97) per-ray near/far from the box, invalid rays patched."""
        t0, t1 = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=opt['box_warp'])
        ok = t1 > t0
        if torch.any(ok).item():
            t0[~ok] = t0[ok].min()
            t1[~ok] = t0[ok].max()
        return t0, t1

    def _forward_fused(self, planes, decoder, ray_origins, ray_directions, opt, needs_grad=False):
        n, m, _ = ray_origins.shape
        sc, sf = int(opt['depth_resolution']), int(opt['depth_resolution_importance'])
        if not (4 <= sc <= 64 and 1 <= sf <= 64):
            return None
        dev = planes.device
        t0 = t1 = None
        if opt['ray_start'] == opt['ray_end'] == 'auto':
            t0, t1 = self._ray_limits(ray_origins, ray_directions, opt)
        # the reference's draws, in its order: rand_like(depths_coarse) [N,M,Sc,1] (renderer.py:190) then rand(N*M, Sf) (:237)
        if t0 is None:
            u_c = torch.rand([n, m, sc, 1], device=dev, dtype=torch.float32)
        else:   # tensor-limits branch: rand_like of the permuted [S,N,M,1] linspace fills in ITS memory order (:184-186)
            u_c = torch.rand([sc, n, m, 1], device=dev, dtype=torch.float32).permute(1, 2, 0, 3)
        u_f = torch.rand([n * m, sf], device=dev, dtype=torch.float32)
        if needs_grad:
            params = [p for p in decoder.parameters()]
            return _FusedRenderFn.apply(self, decoder, opt, u_c, u_f, t0, t1, planes, ray_origins, ray_directions, *params)
        return fused_render(planes, decoder, ray_origins, ray_directions, opt, u_c, u_f, t0, t1)

    # ------------------------------------------------------------------------------------------------
    def _forward_tensor_ops(self, planes, decoder, ray_origins, ray_directions, opt, point_fn=None):
        """The differentiable tensor-op restatement of forward().  ``point_fn(points, directions) -> {'rgb', 'sigma'}`` replaces the
        plane lookup + decoder (ImportanceSemanticRenderer)."""
        if opt['ray_start'] == opt['ray_end'] == 'auto':
            t0, t1 = self._ray_limits(ray_origins, ray_directions, opt)
            z_c = self.sample_stratified(ray_origins, t0, t1, opt['depth_resolution'], opt['disparity_space_sampling'])
        else:
            z_c = self.sample_stratified(ray_origins, opt['ray_start'], opt['ray_end'], opt['depth_resolution'], opt['disparity_space_sampling'])
        n, m, sc, _ = z_c.shape

        def decode(z):
            s = z.shape[2]
            pts = (ray_origins.unsqueeze(-2) + z * ray_directions.unsqueeze(-2)).reshape(n, -1, 3)
            dirs = ray_directions.unsqueeze(-2).expand(-1, -1, s, -1).reshape(n, -1, 3)
            if point_fn is not None:
                out = point_fn(pts, dirs)
            elif torch.is_grad_enabled():                    # this method IS the tensor-op formulation: under autograd its point queries are tensor ops too
                out = self._points_tensor_ops(planes, decoder, pts, dirs, opt)      # (the replay backward and the tests' reference runs rely on that)
            else:
                out = self.run_model(planes, decoder, pts, dirs, opt)
            return out['rgb'].reshape(n, m, s, -1), out['sigma'].reshape(n, m, s, 1)

        c_c, s_c = decode(z_c)
        sf = opt['depth_resolution_importance']
        if sf > 0:
            _, _, w = self.ray_marcher(c_c, s_c, z_c, opt)
            z_f = self.sample_importance(z_c, w, sf)
            c_f, s_f = decode(z_f)
            z_all, c_all, s_all = self.unify_samples(z_c, c_c, s_c, z_f, c_f, s_f)
            rgb, depth, w = self.ray_marcher(c_all, s_all, z_all, opt)
        else:
            rgb, depth, w = self.ray_marcher(c_c, s_c, z_c, opt)
        return rgb, depth, w.sum(2)

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        """Sample the planes at 3-D points and decode: {'rgb': [N,P,C], 'sigma': [N,P,1]} (renderer.py:142-148)."""
        self.plane_axes = self.plane_axes.to(sample_coordinates.device)
        needs_grad = torch.is_grad_enabled() and (planes.requires_grad or sample_coordinates.requires_grad
                                                  or any(p.requires_grad for p in decoder.parameters()))
        # a graph that needs gradients (the density regularisation, loss.py:681-706) takes the fused forward + p3d_sample_points_backward;
        # only a gradient w.r.t. the coordinates themselves has no fused form
        reason = self._fused_reason(planes, decoder, options, needs_grad, trainable=not (torch.is_grad_enabled() and sample_coordinates.requires_grad))
        if reason is None:
            if needs_grad:
                rgb, sigma = _FusedPointsFn.apply(decoder, options, sample_coordinates, planes, *decoder.parameters())
            else:
                rgb, sigma = fused_sample_points(planes, decoder, sample_coordinates, options)
            return {'rgb': rgb, 'sigma': sigma}
        self._tensor_op_guard(planes, reason)
        return self._points_tensor_ops(planes, decoder, sample_coordinates, sample_directions, options)

    def _points_tensor_ops(self, planes, decoder, sample_coordinates, sample_directions, options):
        """run_model as the reference writes it (renderer.py:142-148): grid_sample + decoder, differentiable in everything."""
        self.plane_axes = self.plane_axes.to(sample_coordinates.device)
        feats = sample_from_planes(self.plane_axes, planes, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
        out = decoder(feats, sample_directions)
        if options.get('density_noise', 0) > 0:
            out['sigma'] += torch.randn_like(out['sigma']) * options['density_noise']
        return out

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _sorted_gather(depths, colors, densities):
        _, order = torch.sort(depths, dim=-2)
        take = lambda t: torch.gather(t, -2, order.expand(-1, -1, -1, t.shape[-1]))
        return take(depths), take(colors), take(densities)

    def sort_samples(self, all_depths, all_colors, all_densities):
        return self._sorted_gather(all_depths, all_colors, all_densities)

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2):
        """Concatenate both sample sets along the ray and order them by depth (renderer.py:157-167)."""
        return self._sorted_gather(torch.cat([depths1, depths2], dim=-2), torch.cat([colors1, colors2], dim=-2),
                                   torch.cat([densities1, densities2], dim=-2))

    def sample_stratified(self, ray_origins, ray_start, ray_end, depth_resolution, disparity_space_sampling=False):
        """Jittered, evenly spaced depths [N,M,S,1]; one uniform per sample (renderer.py:169-192)."""
        n, m, _ = ray_origins.shape
        dev, s = ray_origins.device, depth_resolution
        if disparity_space_sampling:
            t = torch.linspace(0, 1, s, device=dev).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
            t += torch.rand_like(t) * (1 / (s - 1))
            return 1. / (1. / ray_start * (1. - t) + 1. / ray_end * t)
        if isinstance(ray_start, torch.Tensor):
            z = math_utils.linspace(ray_start, ray_end, s).permute(1, 2, 0, 3)
            z += torch.rand_like(z) * ((ray_end - ray_start) / (s - 1))[..., None]
            return z
        z = torch.linspace(ray_start, ray_end, s, device=dev).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
        z += torch.rand_like(z) * ((ray_end - ray_start) / (s - 1))
        return z

    def sample_importance(self, z_vals, weights, N_importance):
        """Importance depths from the coarse weights [N,M,S_f,1], detached (renderer.py:194-212)."""
        with torch.no_grad():
            n, m, s, _ = z_vals.shape
            z = z_vals.reshape(n * m, s)
            w = weights.reshape(n * m, -1)
            w = torch.nn.functional.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
            w = torch.nn.functional.avg_pool1d(w, 2, 1).squeeze(1) + 0.01
            mids = 0.5 * (z[:, :-1] + z[:, 1:])
            return self.sample_pdf(mids, w[:, 1:-1], N_importance).detach().reshape(n, m, N_importance, 1)

    def sample_pdf(self, bins, weights, N_importance, det=False, eps=1e-5):
        """Inverse-CDF sampling of ``N_importance`` depths per ray (renderer.py:214-253)."""
        rays, nb = weights.shape
        w = weights + eps
        cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
        cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
        if det:
            u = torch.linspace(0, 1, N_importance, device=bins.device).expand(rays, N_importance)
        else:
            u = torch.rand(rays, N_importance, device=bins.device)
        u = u.contiguous()
        idx = torch.searchsorted(cdf, u, right=True)
        lo, hi = (idx - 1).clamp_min(0), idx.clamp_max(nb)
        c0, c1 = cdf.gather(1, lo), cdf.gather(1, hi)
        b0, b1 = bins.gather(1, lo), bins.gather(1, hi)
        span = c1 - c0
        span = torch.where(span < eps, torch.ones_like(span), span)
        return b0 + (u - c0) / span * (b1 - b0)


def _osg_pair(decoder, n_in, squash_required=None):
    """(fc1, fc2, lr_mul) of a single-MLP OSG decoder FC(n_in, 64) - Softplus - FC(64, 33), else None."""
    seq = getattr(decoder, 'net', None)
    if not (isinstance(seq, torch.nn.Sequential) and len(seq) == 3 and isinstance(seq[1], torch.nn.Softplus)) or hasattr(decoder, 'net_semantic'):
        return None
    fc1, fc2 = seq[0], seq[2]
    if not all(hasattr(fc, 'weight_gain') and getattr(fc, 'activation', None) == 'linear' and fc.bias is not None for fc in (fc1, fc2)):
        return None
    if tuple(fc1.weight.shape) != (64, n_in) or tuple(fc2.weight.shape) != (33, 64) or seq[1].beta != 1 or seq[1].threshold != 20:
        return None
    lr = float(fc1.bias_gain)
    if abs(fc1.weight_gain - lr / n_in ** 0.5) > 1e-12 or abs(fc2.weight_gain - lr / 8.0) > 1e-12 or fc2.bias_gain != lr:
        return None
    return fc1, fc2, lr


def _plane_set_cl(planes):
    """[N,3,32,H,W] planes as the kernels read them: (tensor, (image, plane, pixel) strides in floats).  A channels-last [N,96,H,W]
    backbone output viewed as [N,3,32,H,W] is read in place; anything else goes through one re-layout pass to [N][3][H][W][32]."""
    n, k, c, h, w = planes.shape
    st = planes.stride()
    if planes.dtype == torch.float32 and st[2] == 1 and st[1] == 32 and st[4] >= 96 and st[3] == w * st[4] and st[0] == h * st[3] and st[4] % 4 == 0 \
            and planes.data_ptr() % 16 == 0:
        return planes.detach(), (st[0], st[1], st[4])
    src = _f32c(planes)
    out = torch.empty([n, 3, h, w, 32], dtype=torch.float32, device=planes.device)
    _lib.check(_lib.lib().p3d_planes_to_channels_last(_lib.ptr(src), _lib.ptr(out), n, h, w, _lib.stream_of(src)), 'planes_to_channels_last')
    return out, (0, 0, 0)


class ImportanceSemanticRenderer(ImportanceRenderer):
    """Renderer of the two-backbone generator (reference: renderer.py:256-438): a texture plane set and a semantic plane set; the label
    decoder reads the semantic features and provides density + labels, the colour decoder reads cat(texture, semantic).  Sampling
    and compositing are ImportanceRenderer's, over the feature vector cat(colour, label).

    Device tensors without an autograd graph run the DUAL variant of the fused kernel (``p3d_render_forward_dual`` /
    ``p3d_sample_points_dual``, csrc/render_device.h): both plane sets are gathered per sample, the colour net's first layer takes its
    64 inputs as two 32-wide MFMA blocks.  Graphs that need gradients (train.py no longer selects this generator, :375) and CPU tensors
    take the tensor-op formulation."""

    def _dual_operands(self, planes_texture, planes_semantic, decoder_texture, decoder_semantic, options, needs_grad):
        """Packed operands of the DUAL kernels, or the reason they do not apply."""
        if fused_policy == 'never':
            return 'fused_policy == never'
        if planes_texture.device.type != 'cuda':
            return 'CPU tensors'
        if needs_grad:
            return 'autograd graph requested (the two-plane-set kernel is inference only)'
        if planes_texture.shape != planes_semantic.shape or planes_texture.ndim != 5 or tuple(planes_texture.shape[1:3]) != (3, 32):
            return f'plane sets {tuple(planes_texture.shape)} / {tuple(planes_semantic.shape)} are not two [N,3,32,H,W] sets'
        if options.get('density_noise', 0) > 0 or options.get('clamp_mode', 'softplus') != 'softplus':
            return 'density_noise / clamp_mode outside the kernel'
        tex, sem = _osg_pair(decoder_texture, 64), _osg_pair(decoder_semantic, 32)
        if tex is None or sem is None or tex[2] != sem[2] or type(decoder_texture).__name__ != 'OSGDecoder' or not hasattr(decoder_semantic, 'final_sigmoid'):
            return 'decoders are not OSGDecoder(64) + OSGDecoder_semantic(32)'
        pt, st_t = _plane_set_cl(planes_texture)
        ps, st_s = _plane_set_cl(planes_semantic)
        if st_t != st_s:                                     # one descriptor describes both sets: bring the odd one to the default layout
            if st_t != (0, 0, 0):
                pt, st_t = _plane_set_cl(planes_texture.contiguous())
            if st_s != (0, 0, 0):
                ps, st_s = _plane_set_cl(planes_semantic.contiguous())
        lib = _lib.lib()
        packed = torch.empty([lib.p3d_render_decoder_floats_dual()], dtype=torch.float32, device=pt.device)
        ws = [_f32c(t) for t in (tex[0].weight, tex[0].bias, tex[1].weight, tex[1].bias, sem[0].weight, sem[0].bias, sem[1].weight, sem[1].bias)]
        _lib.check(lib.p3d_pack_decoder_dual(*[_lib.ptr(t) for t in ws], tex[2], _lib.ptr(packed), _lib.stream_of(packed)), 'pack_decoder_dual')
        n, _, _, h, w = planes_texture.shape

        def desc(rays_per_img=1, start=0.0, end=0.0):
            return _RenderDesc(n, rays_per_img, h, w, 2, int(bool(decoder_semantic.final_sigmoid)), int(options.get('depth_resolution', 0)),
                               int(options.get('depth_resolution_importance', 0)), int(bool(options.get('disparity_space_sampling', False))),
                               int(bool(options.get('white_back', False))), float(start), float(end), float(options['box_warp']), *st_t, 1, 0)
        return pt, ps, packed, desc, ws

    def forward(self, planes_texture, planes_semantic, decoder_texture, decoder_semantic, ray_origins, ray_directions, rendering_options):
        self.plane_axes = self.plane_axes.to(ray_origins.device)
        opt = rendering_options
        needs_grad = torch.is_grad_enabled() and (planes_texture.requires_grad or planes_semantic.requires_grad or ray_origins.requires_grad
                                                  or any(p.requires_grad for d in (decoder_texture, decoder_semantic) for p in d.parameters()))
        ops = self._dual_operands(planes_texture, planes_semantic, decoder_texture, decoder_semantic, opt, needs_grad)
        sc, sf = int(opt['depth_resolution']), int(opt['depth_resolution_importance'])
        if not isinstance(ops, str) and not (4 <= sc <= 64 and 1 <= sf <= 64):
            ops = 'sample counts outside the fused kernel envelope (4..64 coarse, 1..64 fine)'
        if not isinstance(ops, str):
            pt, ps, packed, desc, _keep = ops
            n, m, _ = ray_origins.shape
            dev = pt.device
            auto = opt['ray_start'] == opt['ray_end'] == 'auto'
            t0 = t1 = None
            if auto:                                                        # tensor limits: the reference's rand_like fills a permuted [S,N,M,1] tensor
                t0, t1 = self._ray_limits(ray_origins, ray_directions, opt)
                u_c = torch.rand([sc, n, m, 1], device=dev, dtype=torch.float32).permute(1, 2, 0, 3)
                t0, t1 = _f32c(t0).reshape(-1), _f32c(t1).reshape(-1)
            else:
                u_c = torch.rand([n, m, sc, 1], device=dev, dtype=torch.float32)    # rand_like(depths_coarse) (renderer.py:190)
            u_f = torch.rand([n * m, sf], device=dev, dtype=torch.float32)           # sample_pdf's draw (:237)
            feat = torch.empty([n, m, 64], device=dev, dtype=torch.float32)
            depth = torch.empty([n, m, 1], device=dev, dtype=torch.float32)
            wsum = torch.empty([n, m, 1], device=dev, dtype=torch.float32)
            mm = torch.empty([2], device=dev, dtype=torch.int32)
            d = desc(m, 0.0 if auto else opt['ray_start'], 0.0 if auto else opt['ray_end'])
            code = _lib.lib().p3d_render_forward_dual(_lib.ptr(pt), _lib.ptr(ps), _lib.ptr(packed), _lib.ptr(_f32c(ray_origins)), _lib.ptr(_f32c(ray_directions)),
                                                      _lib.ptr(_f32c(u_c)), _lib.ptr(u_f), _lib.ptr(t0), _lib.ptr(t1), ctypes.byref(d), _lib.ptr(feat), _lib.ptr(depth),
                                                      _lib.ptr(wsum), _lib.ptr(mm), _lib.stream_of(feat))
            _lib.check(code, 'render_forward_dual')
            return feat, depth, wsum
        self._tensor_op_guard(planes_texture, ops)

        def point_fn(pts, dirs):
            out = self._run_model_tensor_ops(planes_texture, planes_semantic, decoder_texture, decoder_semantic, pts, dirs, rendering_options)
            return {'rgb': torch.cat([out['rgb'], out['semantic']], dim=-1), 'sigma': out['sigma']}
        return self._forward_tensor_ops(None, None, ray_origins, ray_directions, rendering_options, point_fn=point_fn)

    def run_model(self, planes_texture, planes_semantic, decoder_texture, decoder_semantic, sample_coordinates, sample_directions, options):
        """-> {'rgb': [N,P,32], 'sigma': [N,P,1], 'semantic': [N,P,32]}  (renderer.py:324-333)."""
        self.plane_axes = self.plane_axes.to(sample_coordinates.device)
        needs_grad = torch.is_grad_enabled() and (planes_texture.requires_grad or planes_semantic.requires_grad or sample_coordinates.requires_grad
                                                  or any(p.requires_grad for d in (decoder_texture, decoder_semantic) for p in d.parameters()))
        ops = self._dual_operands(planes_texture, planes_semantic, decoder_texture, decoder_semantic, options, needs_grad)
        if isinstance(ops, str):
            self._tensor_op_guard(planes_texture, ops)
            return self._run_model_tensor_ops(planes_texture, planes_semantic, decoder_texture, decoder_semantic, sample_coordinates, sample_directions, options)
        pt, ps, packed, desc, _keep = ops
        n, p, _ = sample_coordinates.shape
        both = torch.empty([n, p, 64], device=pt.device, dtype=torch.float32)
        sigma = torch.empty([n, p, 1], device=pt.device, dtype=torch.float32)
        d = desc()
        code = _lib.lib().p3d_sample_points_dual(_lib.ptr(pt), _lib.ptr(ps), _lib.ptr(packed), _lib.ptr(_f32c(sample_coordinates)), ctypes.byref(d), p,
                                                 _lib.ptr(both), _lib.ptr(sigma), _lib.stream_of(both))
        _lib.check(code, 'sample_points_dual')
        return {'sigma': sigma, 'rgb': both[..., :32], 'semantic': both[..., 32:]}

    def _run_model_tensor_ops(self, planes_texture, planes_semantic, decoder_texture, decoder_semantic, sample_coordinates, sample_directions, options):
        tex = sample_from_planes(self.plane_axes, planes_texture, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
        sem = sample_from_planes(self.plane_axes, planes_semantic, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
        label = decoder_semantic(sem, sample_directions)
        colour = decoder_texture(torch.cat([tex, sem], dim=-1), sample_directions)
        out = {'sigma': label['sigma'], 'rgb': colour['rgb'], 'semantic': label['rgb']}
        if options.get('density_noise', 0) > 0:
            out['sigma'] = out['sigma'] + torch.randn_like(out['sigma']) * options['density_noise']
        return out


def importance_sample_native(z_coarse, w_coarse, u_fine, sort=False):
    """p3d_importance_sample on device tensors: z [R,Sc], w [R,Sc-1], u [R,Sf] -> z_fine [R,Sf]."""
    z, w, u = _f32c(z_coarse), _f32c(w_coarse), _f32c(u_fine)
    out = torch.empty_like(u)
    code = _lib.lib().p3d_importance_sample(_lib.ptr(z), _lib.ptr(w), _lib.ptr(u), _lib.ptr(out), z.shape[0], z.shape[1], u.shape[1], int(sort), _lib.stream_of(z))
    _lib.check(code, 'importance_sample')
    return out


def importance_sample_index_native(z_coarse, w_coarse, u_fine):
    """p3d_importance_sample_index: (sorted z_fine [R,Sf], bin index per draw [R,Sf] int32, merge pattern [R,Sc+Sf] bool: True where the k-th sample of
    the merged ray is an importance sample) — the integer side of sample_pdf / unify_samples, for the parity tests."""
    z, w, u = _f32c(z_coarse), _f32c(w_coarse), _f32c(u_fine)
    out = torch.empty_like(u)
    bins = torch.empty(u.shape, dtype=torch.int32, device=u.device)
    words = torch.empty([u.shape[0], 4], dtype=torch.int32, device=u.device)
    code = _lib.lib().p3d_importance_sample_index(_lib.ptr(z), _lib.ptr(w), _lib.ptr(u), _lib.ptr(out), _lib.ptr(bins), _lib.ptr(words), z.shape[0], z.shape[1], u.shape[1], 1,
                                                  _lib.stream_of(z))
    _lib.check(code, 'importance_sample_index')
    k = torch.arange(z.shape[1] + u.shape[1], device=u.device)
    merged = ((words[:, (k >> 5)] >> (k & 31)) & 1).bool()
    return out, bins, merged


class _replay_draws:
    """Make torch.rand_like / torch.rand hand back given tensors, in order (the renderer's two uniform draws)."""

    def __init__(self, *draws):
        self.draws = list(draws)

    def __enter__(self):
        self._rl, self._r = torch.rand_like, torch.rand
        it = iter(self.draws)
        torch.rand_like = lambda t, *a, **k: next(it).to(t.device).reshape(t.shape)
        torch.rand = lambda *a, **k: next(it)
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand = self._rl, self._r


backward_calls = {'fused': 0, 'replay': 0, 'points': 0}      # which backward _FusedRenderFn took (tests: the training step must take 'fused'); 'points': _FusedPointsFn


class _FusedRenderFn(torch.autograd.Function):
    """Training-mode rendering: the FORWARD is the fused kernel and keeps nothing per-sample (the reference holds ~1.2 GB of
    sampled features per image for autograd); the BACKWARD recomputes on the device — ``p3d_render_backward``: the forward
    sweep again with a tape, then a point-wise MFMA backward (csrc/render_bwd.hip) — and returns gradients for the planes and
    the decoder parameters.  If a gradient w.r.t. the rays or the depth output is requested (no training loss does), or with
    ``fused_backward = False``, it replays the differentiable tensor-op renderer on the same draws under autograd instead."""

    @staticmethod
    def forward(ctx, renderer, decoder, opt, u_c, u_f, t0, t1, planes, ray_o, ray_d, *params):
        out = fused_render(planes, decoder, ray_o, ray_d, opt, u_c, u_f, t0, t1, exact_fp32=True)
        if out is None:
            raise RuntimeError('fused training render: sample counts outside the kernel envelope')
        ctx.renderer, ctx.decoder, ctx.opt = renderer, decoder, opt
        ctx.limits = (t0, t1)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(u_c, u_f, planes, ray_o, ray_d)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable          # the device kernels are not differentiable: a double backward must raise, not return zeros
    def backward(ctx, g_feat, g_depth, g_wsum):
        u_c, u_f, planes, ray_o, ray_d = ctx.saved_tensors
        params = [p for p in ctx.decoder.parameters()]
        rays_need_grad = ctx.needs_input_grad[8] or ctx.needs_input_grad[9]
        if fused_backward and g_depth is None and not rays_need_grad and g_feat is not None:
            backward_calls['fused'] += 1
            g_planes, g_params = fused_render_backward(planes, ctx.decoder, ray_o, ray_d, ctx.opt, u_c, u_f, ctx.limits[0], ctx.limits[1], g_feat, g_wsum)
            return (None,) * 7 + (g_planes if ctx.needs_input_grad[7] else None, None, None) + tuple(g_params)
        # depth gradients / ray gradients: the differentiable tensor-op renderer, replayed on the same draws
        backward_calls['replay'] += 1
        n, m = ray_o.shape[0], ray_o.shape[1]
        dev = planes.device
        nch = 32 * len(_decoder_nets(ctx.decoder)[0])
        g_feat = torch.zeros([n, m, nch], device=dev) if g_feat is None else g_feat
        g_depth = torch.zeros([n, m, 1], device=dev) if g_depth is None else g_depth
        g_wsum = torch.zeros([n, m, 1], device=dev) if g_wsum is None else g_wsum
        with torch.enable_grad():
            pl = planes.detach().requires_grad_(ctx.needs_input_grad[7])
            ro = ray_o.detach().requires_grad_(ctx.needs_input_grad[8])
            rd = ray_d.detach().requires_grad_(ctx.needs_input_grad[9])
            with _replay_draws(u_c, u_f):
                feat, depth, wsum = ctx.renderer._forward_tensor_ops(pl, ctx.decoder, ro, rd, ctx.opt)
            wanted = [t for t in [pl, ro, rd] + params if t.requires_grad]
            grads = torch.autograd.grad([feat, depth, wsum], wanted, [g_feat, g_depth, g_wsum], allow_unused=True)
        it = iter(grads)
        res = [next(it) if t.requires_grad else None for t in [pl, ro, rd] + params]
        return (None,) * 7 + tuple(res)


def fused_sample_points(planes, decoder, coordinates, opt):
    """One launch of p3d_sample_points: (rgb [N,P,32*n_nets], sigma [N,P,1]) at coordinates [N,P,3] (exact fp32 MFMA decoder)."""
    n, p, _ = coordinates.shape
    ctx = _FusedContext(planes, _decoder_nets(decoder))
    xyz = _f32c(coordinates)
    rgb = torch.empty([n, p, 32 * ctx.n_nets], device=planes.device, dtype=torch.float32)
    sigma = torch.empty([n, p, 1], device=planes.device, dtype=torch.float32)
    d = ctx.desc(opt)
    code = _lib.lib().p3d_sample_points(_lib.ptr(ctx.planes_cl), _lib.ptr(ctx.packed), _lib.ptr(xyz), ctypes.byref(d), p,
                                        _lib.ptr(rgb), _lib.ptr(sigma), _lib.stream_of(rgb))
    _lib.check(code, 'sample_points')
    return rgb, sigma


def _decoder_param_grads(decoder, nets, d_dec):
    """Effective-weight gradient record of the device kernels -> gradients of the raw FullyConnectedLayer parameters, in ``decoder.parameters()`` order."""
    stride = _lib.lib().p3d_render_grad_decoder_floats() // 2
    by_param = {}
    for i, (fc1, fc2) in enumerate(nets):
        g = d_dec[i * stride:(i + 1) * stride]
        by_param[id(fc1.weight)] = g[0:2048].reshape(64, 32) * fc1.weight_gain
        by_param[id(fc1.bias)] = g[2048:2112] * fc1.bias_gain
        by_param[id(fc2.weight)] = g[2112:4224].reshape(33, 64) * fc2.weight_gain
        by_param[id(fc2.bias)] = g[4224:4257] * fc2.bias_gain
    return [by_param.get(id(p)) if p.requires_grad else None for p in decoder.parameters()]


def _pack_decoder_bwd(nets, lr_mul, dev):
    lib = _lib.lib()
    packed_bwd = torch.empty([lib.p3d_render_bwd_decoder_floats()], dtype=torch.float32, device=dev)
    w1s = [_f32c(fc1.weight) for fc1, _ in nets] + [None]
    w2s = [_f32c(fc2.weight) for _, fc2 in nets] + [None]
    _lib.check(lib.p3d_pack_decoder_bwd(_lib.ptr(w1s[0]), _lib.ptr(w2s[0]), _lib.ptr(w1s[1]), _lib.ptr(w2s[1]), len(nets), lr_mul, _lib.ptr(packed_bwd),
                                        _lib.stream_of(packed_bwd)), 'pack_decoder_bwd')
    return packed_bwd


def fused_sample_points_backward(planes, decoder, coordinates, opt, g_rgb, g_sigma):
    """dL/dplanes and dL/d(decoder parameters) of ``fused_sample_points`` from dL/drgb [N,P,32*n_nets] and dL/dsigma [N,P,1] (either may be
    None), by one launch of p3d_sample_points_backward (csrc/render_bwd.hip)."""
    info = _decoder_nets(decoder)
    nets, lr_mul, _ = info
    lib = _lib.lib()
    n, p, _ = coordinates.shape
    dev = planes.device
    ctx = _FusedContext(planes, info)
    packed_bwd = _pack_decoder_bwd(nets, lr_mul, dev)
    xyz = _f32c(coordinates)
    gr = None if g_rgb is None else _f32c(g_rgb)
    gs = None if g_sigma is None else _f32c(g_sigma).reshape(-1)
    d_planes = torch.empty([n, 3, ctx.h, ctx.w, 32], dtype=torch.float32, device=dev)
    d_dec = torch.empty([lib.p3d_render_grad_decoder_floats()], dtype=torch.float32, device=dev)
    d = ctx.desc(opt)
    code = lib.p3d_sample_points_backward(_lib.ptr(ctx.planes_cl), _lib.ptr(ctx.packed), _lib.ptr(packed_bwd), _lib.ptr(xyz), ctypes.byref(d), p,
                                          _lib.ptr(gr), _lib.ptr(gs), _lib.ptr(d_planes), _lib.ptr(d_dec), _lib.stream_of(d_planes))
    _lib.check(code, 'sample_points_backward')
    g_planes = d_planes.permute(0, 1, 4, 2, 3)                    # [N, 3, 32, H, W] view of the channels-last gradient
    if planes.dim() == 5 and planes.is_contiguous():
        g_planes = g_planes.contiguous()
    return g_planes, _decoder_param_grads(decoder, nets, d_dec)


class _FusedPointsFn(torch.autograd.Function):
    """Point queries under autograd (G.sample_mixed in the density regularisation, loss.py:681-706): fused forward, nothing per-point
    kept; the backward recomputes gather + decoder on the device (``p3d_sample_points_backward``) and returns gradients for the
    planes and the decoder parameters."""

    @staticmethod
    def forward(ctx, decoder, opt, coordinates, planes, *params):
        rgb, sigma = fused_sample_points(planes, decoder, coordinates, opt)
        ctx.decoder, ctx.opt = decoder, opt
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(planes, coordinates)
        return rgb, sigma

    @staticmethod
    @torch.autograd.function.once_differentiable          # the device kernels are not differentiable: a double backward must raise, not return zeros
    def backward(ctx, g_rgb, g_sigma):
        planes, coordinates = ctx.saved_tensors
        backward_calls['points'] += 1
        if g_rgb is None and g_sigma is None:
            return (None,) * (4 + len(list(ctx.decoder.parameters())))
        g_planes, g_params = fused_sample_points_backward(planes, ctx.decoder, coordinates, ctx.opt, g_rgb, g_sigma)
        if g_rgb is None:                                    # only the density carries a gradient (the density regularisation): with two nets the
            nets = _decoder_nets(ctx.decoder)[0]             # colour net is not part of the graph — None for its parameters, as autograd reports it
            if len(nets) == 2:
                skip = {id(p) for fc in nets[0] for p in fc.parameters()}
                g_params = [None if id(p) in skip else g for p, g in zip(ctx.decoder.parameters(), g_params)]
        return (None, None, None, g_planes if ctx.needs_input_grad[3] else None) + tuple(g_params)


def fused_render_backward(planes, decoder, ray_origins, ray_directions, opt, u_coarse, u_fine, t_start, t_end, g_feat, g_wsum=None, debug=False):
    """dL/dplanes (same shape and layout class as ``planes``) and dL/d(decoder parameters) (in ``decoder.parameters()`` order) of the
    fused render, by the two recomputing launches of csrc/render_bwd.hip.  Decoder parameters must require grad to get an entry."""
    info = _decoder_nets(decoder)
    nets, lr_mul, _ = info
    lib = _lib.lib()
    n, m, _ = ray_origins.shape
    sc, sf = int(opt['depth_resolution']), int(opt['depth_resolution_importance'])
    s_all = sc + sf
    dev = planes.device
    ctx = _FusedContext(planes, info)
    packed_bwd = _pack_decoder_bwd(nets, lr_mul, dev)
    auto = t_start is not None
    t0 = _f32c(t_start).reshape(-1) if auto else None
    t1 = _f32c(t_end).reshape(-1) if auto else None
    ro, rd, uc, uf = _f32c(ray_origins), _f32c(ray_directions), _f32c(u_coarse), _f32c(u_fine)
    gf = _f32c(g_feat)
    gw = None if g_wsum is None else _f32c(g_wsum).reshape(-1)
    tape_i = torch.empty([n * m, s_all - 1, 4], dtype=torch.float32, device=dev)
    tape_s = torch.empty([n * m, s_all, 4], dtype=torch.float32, device=dev)
    d_planes = torch.empty([n, 3, ctx.h, ctx.w, 32], dtype=torch.float32, device=dev)
    d_dec = torch.empty([lib.p3d_render_grad_decoder_floats()], dtype=torch.float32, device=dev)
    d = ctx.desc(opt, rays_per_img=m, start=0.0 if auto else opt['ray_start'], end=0.0 if auto else opt['ray_end'], raster=True)
    code = lib.p3d_render_backward(_lib.ptr(ctx.planes_cl), _lib.ptr(ctx.packed), _lib.ptr(packed_bwd), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(uc), _lib.ptr(uf),
                                   _lib.ptr(t0), _lib.ptr(t1), ctypes.byref(d), _lib.ptr(gf), _lib.ptr(gw), _lib.ptr(tape_i), _lib.ptr(tape_s),
                                   _lib.ptr(d_planes), _lib.ptr(d_dec), _lib.stream_of(d_planes))
    _lib.check(code, 'render_backward')
    g_planes = d_planes.permute(0, 1, 4, 2, 3)                    # [N, 3, 32, H, W] view of the channels-last gradient
    if planes.dim() == 5 and planes.is_contiguous():
        g_planes = g_planes.contiguous()
    g_params = _decoder_param_grads(decoder, nets, d_dec)
    if debug:                                                     # the per-sample tape: z, colour weight, dL/dsigma (tests)
        return g_planes, g_params, tape_s
    return g_planes, g_params


def fused_render(planes, decoder, ray_origins, ray_directions, opt, u_coarse, u_fine, t_start=None, t_end=None, debug=False, exact_fp32=False, packed=None):
    """One launch of the fused ray-marcher with explicit uniforms (u_coarse [N,M,Sc(,1)], u_fine [N*M,Sf]).
    Returns (feat [N,M,C], depth [N,M,1], wsum [N,M,1]) and, with debug, the sorted fine depths [N*M,Sf] and the
    coarse weights [N*M,Sc-1] the kernel used; with debug='bins' also the searchsorted index of every draw [N*M,Sf] (int32, draw order)."""
    info = _decoder_nets(decoder)
    if info is None:
        raise RuntimeError(f'fused_render: unsupported decoder {type(decoder).__name__}')
    n, m, _ = ray_origins.shape
    sc, sf = int(opt['depth_resolution']), int(opt['depth_resolution_importance'])
    dev = planes.device
    from pix2pix3d_amd.torch_utils.ops import modconv
    mode = 1 if (mlp_bf16x3 and not exact_fp32) else (2 if (mlp_l1x6 and modconv.f32_x6) else 0)      # (the training forward stays fp32-accurate: its backward recomputes in fp32)
    ctx = _FusedContext(planes, info, bf16x3=mode, packed=packed)
    auto = t_start is not None
    t0 = _f32c(t_start).reshape(-1) if auto else None
    t1 = _f32c(t_end).reshape(-1) if auto else None
    ro, rd, uc, uf = _f32c(ray_origins), _f32c(ray_directions), _f32c(u_coarse), _f32c(u_fine)
    assert uc.numel() == n * m * sc and uf.numel() == n * m * sf
    feat = torch.empty([n, m, 32 * ctx.n_nets], device=dev, dtype=torch.float32)
    depth = torch.empty([n, m, 1], device=dev, dtype=torch.float32)
    wsum = torch.empty([n, m, 1], device=dev, dtype=torch.float32)
    mm = torch.empty([2], device=dev, dtype=torch.int32)
    dbg_f = torch.empty([n * m, sf], device=dev, dtype=torch.float32) if debug else None
    dbg_w = torch.empty([n * m, sc - 1], device=dev, dtype=torch.float32) if debug else None
    dbg_b = torch.empty([n * m, sf], device=dev, dtype=torch.int32) if debug == 'bins' else None
    # raster=True is a pure scheduling hint (which wave takes which ray); results never depend on it
    d = ctx.desc(opt, rays_per_img=m, start=0.0 if auto else opt['ray_start'], end=0.0 if auto else opt['ray_end'], raster=True)
    with _lib.kernel_timer('render_forward', feat):
        if dbg_b is not None:
            code = _lib.lib().p3d_render_forward_debug(_lib.ptr(ctx.planes_cl), _lib.ptr(ctx.packed), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(uc), _lib.ptr(uf),
                                                       _lib.ptr(t0), _lib.ptr(t1), ctypes.byref(d), _lib.ptr(feat), _lib.ptr(depth), _lib.ptr(wsum),
                                                       _lib.ptr(mm), _lib.ptr(dbg_f), _lib.ptr(dbg_w), _lib.ptr(dbg_b), _lib.stream_of(feat))
        else:
            code = _lib.lib().p3d_render_forward(_lib.ptr(ctx.planes_cl), _lib.ptr(ctx.packed), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(uc), _lib.ptr(uf),
                                                 _lib.ptr(t0), _lib.ptr(t1), ctypes.byref(d), _lib.ptr(feat), _lib.ptr(depth), _lib.ptr(wsum),
                                                 _lib.ptr(mm), _lib.ptr(dbg_f), _lib.ptr(dbg_w), _lib.stream_of(feat))
    if code == _lib.P3D_ERR_UNSUPPORTED:
        return None
    _lib.check(code, 'render_forward')
    if dbg_b is not None:
        return feat, depth, wsum, dbg_f, dbg_w, dbg_b
    return (feat, depth, wsum, dbg_f, dbg_w) if debug else (feat, depth, wsum)
