"""Census of the kernels of one G.synthesis step that are NOT this package's: which ATen / vendor kernels are still launched, how often,
how long, and from which Python line.  Run on a GPU box:  python tests/gpu_aten_census.py [dataset] [batch]"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dataset = sys.argv[1] if len(sys.argv) > 1 else 'seg2cat'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    args = SimpleNamespace(dataset=dataset, batch=batch, depth=128)
    G, kw, info, ws, c = bench.build(args, 'cuda')
    G, ws, c = G.cuda(), ws.cuda(), c.cuda()
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    rmod.fused_policy = 'require'
    syn = dict(noise_mode='const', neural_rendering_resolution=info['nrr'])
    with torch.no_grad():
        for _ in range(3):
            G.synthesis(ws, c, **syn)
        torch.cuda.synchronize()
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            G.synthesis(ws, c, **syn)
            torch.cuda.synchronize()
    rows = {}
    total = 0.0
    n_launch = 0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CUDA:
            continue
        dur = ev.device_time if hasattr(ev, 'device_time') else ev.cuda_time
        total += dur
        n_launch += 1
        name = ev.name
        if 'p3d::' in name:
            continue
        key = name[:90]
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += dur
    print(f'step: {n_launch} device events, {total / 1e3:.3f} ms of kernel time')
    print('kernels outside p3d::')
    for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:4d} x {t / max(n, 1):8.1f} us = {t / 1e3:7.3f} ms  {k}')
    # ATen ops (CPU side) with their innermost package frame
    ops = {}
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::') or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
            continue
        frame = next((f for f in (ev.stack or []) if 'pix2pix3d_amd' in f), '?')
        k = (ev.name, frame.split('pix2pix3d_amd/')[-1][:70])
        ops[k] = ops.get(k, 0) + 1
    print('top-level aten ops by call site')
    for (name, frame), n in sorted(ops.items(), key=lambda kv: -kv[1])[:70]:
        print(f'{n:4d}  {name:32s} {frame}')


if __name__ == '__main__':
    main()
