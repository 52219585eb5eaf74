"""Pixel-centre rays from OpenCV-convention cameras (reference: training/volumetric_rendering/ray_sampler.py:24-62)."""
import ctypes

import torch

from ... import _lib

_lib.register('p3d_ray_sample', ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int32] * 2 + [ctypes.c_void_p])
_lib.register('p3d_ray_sample_labels', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 2 + [ctypes.c_void_p])
native = True        # device inference: one launch instead of ~20 tensor ops (csrc/small_ops.hip)


class RaySampler(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_origins_h, self.ray_directions, self.depths, self.image_coords, self.rendering_options = None, None, None, None, None

    def forward(self, cam2world_matrix, intrinsics, resolution):
        """cam2world [N,4,4], intrinsics [N,3,3] (normalised: fx, fy, cx, cy, skew) -> origins, directions [N, R*R, 3].

        Pixel (row i, col j) has image coordinates ((j+.5)/R, (i+.5)/R); it is lifted to z = 1 in the camera
        frame, moved to world space and the direction normalised.  Rays are ordered row-major."""
        n, dev = cam2world_matrix.shape[0], cam2world_matrix.device
        r = int(resolution)
        if native and cam2world_matrix.is_cuda and cam2world_matrix.dtype == torch.float32 and intrinsics.dtype == torch.float32 \
                and not (torch.is_grad_enabled() and (cam2world_matrix.requires_grad or intrinsics.requires_grad)):
            origins = torch.empty([n, r * r, 3], dtype=torch.float32, device=dev)
            dirs = torch.empty_like(origins)
            c2w0, k0 = cam2world_matrix.detach(), intrinsics.detach()
            if (c2w0.stride() == (c2w0.stride(0), 4, 1) and k0.stride() == (c2w0.stride(0), 3, 1) and c2w0.stride(0) >= 25 and c2w0.untyped_storage().data_ptr() == k0.untyped_storage().data_ptr()
                    and k0.storage_offset() == c2w0.storage_offset() + 16):
                # the two views of ONE [N, >= 25] camera-label tensor (triplane.py: c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3)): read in place
                _lib.check(_lib.lib().p3d_ray_sample_labels(_lib.ptr(c2w0), c2w0.stride(0), _lib.ptr(origins), _lib.ptr(dirs), n, r, _lib.stream_of(origins)), 'ray_sample_labels')
                return origins, dirs
            c2w, k = c2w0.contiguous(), k0.contiguous()
            _lib.check(_lib.lib().p3d_ray_sample(_lib.ptr(c2w), _lib.ptr(k), _lib.ptr(origins), _lib.ptr(dirs), n, r, _lib.stream_of(origins)), 'ray_sample')
            return origins, dirs
        fx, fy = intrinsics[:, 0, 0, None], intrinsics[:, 1, 1, None]
        cx, cy, sk = intrinsics[:, 0, 2, None], intrinsics[:, 1, 2, None], intrinsics[:, 0, 1, None]
        centres = torch.arange(r, dtype=torch.float32, device=dev) * (1. / r) + (0.5 / r)
        x_img = centres.repeat(r)[None].expand(n, -1)                    # column coordinate varies fastest
        y_img = centres.repeat_interleave(r)[None].expand(n, -1)
        x_cam = (x_img - cx + cy * sk / fy - sk * y_img / fy) / fx
        y_cam = (y_img - cy) / fy
        ones = torch.ones_like(x_cam)
        pts = torch.stack([x_cam, y_cam, ones, ones], dim=-1)            # [N, M, 4] homogeneous, z = 1
        world = torch.bmm(cam2world_matrix, pts.transpose(1, 2)).transpose(1, 2)[:, :, :3]
        origin = cam2world_matrix[:, :3, 3]
        dirs = torch.nn.functional.normalize(world - origin[:, None, :], dim=2)
        return origin[:, None, :].expand(-1, r * r, -1).contiguous(), dirs
