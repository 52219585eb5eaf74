"""Midpoint-rule volume compositing (reference: training/volumetric_rendering/ray_marcher.py:25-57).
This tensor-op form serves CPU tensors and autograd; on the GPU inference path the same arithmetic runs
inside the fused kernel (csrc/render.hip, phase C)."""
import torch
import torch.nn.functional as F


class MipRayMarcher2(torch.nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rendering_options):
        """colors [N,M,S,C], densities [N,M,S,1], depths [N,M,S,1] -> rgb [N,M,C], depth [N,M,1], weights [N,M,S-1,1]."""
        assert rendering_options['clamp_mode'] == 'softplus', 'MipRayMarcher only supports `clamp_mode`=`softplus`!'
        lo, hi = slice(None, -1), slice(1, None)
        seg = depths[:, :, hi] - depths[:, :, lo]
        c_mid = (colors[:, :, lo] + colors[:, :, hi]) / 2
        z_mid = (depths[:, :, lo] + depths[:, :, hi]) / 2
        sigma = F.softplus((densities[:, :, lo] + densities[:, :, hi]) / 2 - 1)
        alpha = 1 - torch.exp(-(sigma * seg))
        trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], dim=-2), dim=-2)[:, :, :-1]
        weights = alpha * trans
        rgb = (weights * c_mid).sum(-2)
        total = weights.sum(2)
        depth = (weights * z_mid).sum(-2) / total
        depth = torch.nan_to_num(depth, float('inf'))
        depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
        if rendering_options.get('white_back', False):
            rgb = rgb + 1 - total
        return rgb * 2 - 1, depth, weights

    def forward(self, colors, densities, depths, rendering_options):
        return self.run_forward(colors, densities, depths, rendering_options)
