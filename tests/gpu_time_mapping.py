"""Times G.mapping (label-map Encoder + MLP) on the device: seg2cat, batch from argv (default 4)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = configs.generator_kwargs('seg2cat')
rk = kw['rendering_kwargs']
torch.manual_seed(0)
G = dnnlib.util.construct_class_by_name(**kw).cuda().eval().requires_grad_(False)
z = torch.randn(n, G.z_dim, device='cuda')
cam = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32, device='cuda')
mask = torch.randint(0, 6, [n, 1, 512, 512], device='cuda').float()
batch = {'mask': mask, 'pose': cam}
with torch.no_grad():
    for _ in range(2):
        ws = G.mapping(z, cam, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ws = G.mapping(z, cam, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
print(f'G.mapping batch {n}: {dt * 1e3:.2f} ms ({dt * 1e3 / n:.2f} ms per label map), ws {tuple(ws.shape)}')
