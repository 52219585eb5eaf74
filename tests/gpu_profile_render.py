"""Runs only the fused ray-marcher at the bench workload (4 img x 128^2 rays x 64+64 samples), for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
from pix2pix3d_amd.training.triplane_cond import OSGDecoder_semantic_lateSeparate
from pix2pix3d_amd import configs
import numpy as np
torch.manual_seed(0)
N, R, S = 4, 128, int(os.environ.get('S', 64))
dec = OSGDecoder_semantic_lateSeparate(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32, 'sigmoid': False, 'semantic_channels': 6}).cuda().requires_grad_(False)
planes = torch.randn(N, 256, 256, 96, device='cuda').permute(0, 3, 1, 2).reshape(N, 3, 32, 256, 256) if os.environ.get('NHWC', '1') == '1' else torch.randn(N, 3, 32, 256, 256, device='cuda')
c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, pivot=(0, 0, -0.06)) for k in range(N)]), device='cuda')
o, d = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), R)
opt = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode='softplus')
uc = torch.rand(N, R * R, S, device='cuda'); uf = torch.rand(N * R * R, S, device='cuda')
for _ in range(int(os.environ.get('REPS', 3))):
    out = rmod.fused_render(planes, dec, o, d, opt, uc, uf)
torch.cuda.synchronize()
s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
for _ in range(5): rmod.fused_render(planes, dec, o, d, opt, uc, uf)
e.record(); torch.cuda.synchronize()
print(f'render {s.elapsed_time(e) / 5:.3f} ms per launch (incl. pack + clamp)')
