"""Times p3d_render_backward alone (tape sweep + point-wise backward) at BASELINE config 3's size: 4 images x 128^2 rays x 48+48 samples."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib
from pix2pix3d_amd.training.volumetric_rendering import renderer as R

n, res = 4, 128
kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
torch.manual_seed(0)
G = dnnlib.util.construct_class_by_name(**kw).cuda().train().requires_grad_(True)
opt = G.rendering_kwargs
cam = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=opt['avg_camera_radius'], pivot=opt['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32, device='cuda')
ro, rd = G.ray_sampler(cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), res)
planes = torch.randn(n, 256, 256, 96, device='cuda').permute(0, 3, 1, 2).reshape(n, 3, 32, 256, 256) * 0.5
uc = torch.rand(n, res * res, 48, device='cuda'); uf = torch.rand(n * res * res, 48, device='cuda')
gf = torch.randn(n, res * res, 64, device='cuda')
for _ in range(2):
    R.fused_render_backward(planes, G.decoder, ro, rd, opt, uc, uf, None, None, gf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    R.fused_render_backward(planes, G.decoder, ro, rd, opt, uc, uf, None, None, gf)
e1.record(); torch.cuda.synchronize()
print(f'render backward (tape sweep + point-wise backward + packs + memsets): {e0.elapsed_time(e1) / 5:.3f} ms  [{os.path.basename(os.environ.get("P3D_LIB_PATH", "libp3d_hip.so"))}]')
