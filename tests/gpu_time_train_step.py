"""Times one generator training pass (G.synthesis forward + backward, training mode, gradients to every parameter) on the device:
seg2cat, batch from argv (default 4), neural rendering resolution from argv (default 64), 48+48 samples."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nrr = int(sys.argv[2]) if len(sys.argv) > 2 else 64
conv2d_gradfix.enabled = True
kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
rk = kw['rendering_kwargs']
torch.manual_seed(0)
G = dnnlib.util.construct_class_by_name(**kw).cuda().train().requires_grad_(True)
ws = torch.randn(n, G.backbone.num_ws, 512, device='cuda')
c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32, device='cuda')


def step():
    out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='random')
    loss = out['image'].float().square().mean() + out['semantic'].float().square().mean() + out['image_raw'].square().mean()
    loss.backward()
    for p in G.parameters():
        p.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'G.synthesis forward + backward (training mode), batch {n}, {nrr}^2 rays x 48+48 samples: {dt * 1e3:.1f} ms, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
