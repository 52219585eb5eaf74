"""Times the renderer in TRAINING mode (gradients to planes + decoder): seg2cat decoder, N images x R^2 rays x 48+48 samples."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib
from pix2pix3d_amd.training.volumetric_rendering import renderer as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
torch.manual_seed(0)
G = dnnlib.util.construct_class_by_name(**kw).cuda().train().requires_grad_(True)
opt = G.rendering_kwargs
import numpy as np
cam = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=opt['avg_camera_radius'], pivot=opt['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32, device='cuda')
ro, rd = G.ray_sampler(cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), res)
planes = (torch.randn(n, 3, 32, 256, 256, device='cuda') * 0.5).requires_grad_(True)


def step():
    feat, depth, w = G.renderer(planes, G.decoder, ro, rd, opt)
    loss = feat.square().mean() + w.mean()
    loss.backward()
    planes.grad = None
    for p in G.decoder.parameters():
        p.grad = None


for policy in (('fused',) if os.environ.get('P3D_ONLY_FUSED') else ('fused', True, False)):
    R.fused_backward = policy == 'fused'
    R.fused_training = bool(policy)
    if not policy:
        R.fused_policy = 'never'
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'renderer fwd+bwd, N={n}, {res}^2 rays, 48+48 samples, {"fused forward + fused backward kernels" if policy == "fused" else ("fused forward + tensor-op recompute backward" if policy else "tensor ops only (reference formulation)")}: '
          f'{dt * 1e3:.1f} ms, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    torch.cuda.reset_peak_memory_stats()
