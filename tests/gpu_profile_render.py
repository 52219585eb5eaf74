"""Runs only the fused ray-marcher at the bench workload (bench.py's generator, latents and cameras: 4 img x 128^2 rays x 64+64 samples on
the backbone's own 256^2 x 96 channels-last planes), for the rocprofv3 --pmc passes of tests/gpu_pmc_render.py.  PLANES=random: N(0,1)
planes and a fresh decoder instead (the round-1/2 profiling workload)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pix2pix3d_amd import configs                                                              # noqa: E402
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod                       # noqa: E402

DATASET = os.environ.get('P3D_PMC_DATASET', 'seg2cat')        # edge2car: BASELINE configs[3] per GPU (batch 8, 64^2 rays x 64+64, white background, sigmoid labels)
N = int(os.environ.get('P3D_PMC_BATCH', 8 if DATASET == 'edge2car' else 4))
R, S = configs.dataset_info(DATASET)['nrr'], int(os.environ.get('S', 64))
torch.manual_seed(0)
if os.environ.get('PLANES', 'model') == 'random':
    from pix2pix3d_amd.training.triplane_cond import OSGDecoder_semantic_lateSeparate
    from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    dec = OSGDecoder_semantic_lateSeparate(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32, 'sigmoid': False, 'semantic_channels': 6}).cuda().requires_grad_(False)
    planes = torch.randn(N, 256, 256, 96, device='cuda').permute(0, 3, 1, 2).reshape(N, 3, 32, 256, 256)
    c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, pivot=(0, 0, -0.06)) for k in range(N)]), device='cuda')
    o, d = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), R)
    opt = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode='softplus')
else:
    import bench
    args = argparse.Namespace(dataset=DATASET, depth=2 * S, batch=N)
    G, kw, info, ws, c = bench.build(args, 'cuda')
    G = G.cuda()
    ws, c = ws.cuda(), c.cuda()
    with torch.no_grad():
        planes = G.backbone.synthesis(ws, noise_mode='const')
        planes = planes.view(N, 3, 32, planes.shape[-2], planes.shape[-1])
        o, d = G.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), R)
    dec, opt = G.decoder, G.rendering_kwargs
uc = torch.rand(N, R * R, S, device='cuda')
uf = torch.rand(N * R * R, S, device='cuda')
for _ in range(int(os.environ.get('REPS', 3))):
    out = rmod.fused_render(planes, dec, o, d, opt, uc, uf)
torch.cuda.synchronize()
s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
ITERS = int(os.environ.get("ITERS", 5))
for _ in range(ITERS):
    rmod.fused_render(planes, dec, o, d, opt, uc, uf)
e.record(); torch.cuda.synchronize()
print(f"render {s.elapsed_time(e) / ITERS:.4f} ms per launch (incl. pack + clamp; {os.environ.get('P3D_LIB_PATH', 'default build')})")
