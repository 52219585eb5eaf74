"""One generator training pass (G.synthesis forward + backward, training mode) and one discriminator 'Dboth' pass (logits, R1
double-backward under no_weight_gradients, parameter gradients; loss.py:849-891) at the afhq training configuration of
BASELINE config 3 (seg2cat, batch 4 per GPU, 128^2... ``nrr`` rays x 48+48 samples, fp16 SR heads and fp16 discriminator top
blocks): wall time per pass and a census of the device kernels that are NOT this package's — in particular any vendor
convolution / GEMM kernel (miopen*, naive_conv*, Cijk_*).

    python tests/gpu_train_census.py [batch=4] [nrr=64] [--json path]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib                                     # noqa: E402
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix                      # noqa: E402

VENDOR = ('miopen', 'naive_conv', 'Cijk_', 'igemm', 'Im2d', 'Col2Im', 'col2im', 'im2col', 'ck::', 'gemm')


def build(n, nrr):
    kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
    rk = kw['rendering_kwargs']
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).cuda().train().requires_grad_(True)
    D = dnnlib.util.construct_class_by_name(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=512, img_channels=3,
                                            channel_base=32768, channel_max=512, num_fp16_res=4, conv_clamp=256, disc_c_noise=0,
                                            block_kwargs=dict(freeze_layers=0), mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=4)).cuda().train().requires_grad_(True)
    ws = torch.randn(n, G.backbone.num_ws, 512, device='cuda')
    c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]),
                     dtype=torch.float32, device='cuda')
    real = {'image': torch.randn(n, 3, 512, 512, device='cuda'), 'image_raw': torch.randn(n, 3, nrr, nrr, device='cuda')}
    return G, D, ws, c, real


def g_step(G, ws, c, nrr):
    out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='random')
    loss = out['image'].float().square().mean() + out['semantic'].float().square().mean() + out['image_raw'].square().mean()
    loss.backward()
    for p in G.parameters():
        p.grad = None


def d_step(D, real, c):
    img = {k: v.detach().requires_grad_(True) for k, v in real.items()}
    logits = D(img, c)
    with conv2d_gradfix.no_weight_gradients():
        grads = torch.autograd.grad(outputs=[logits.sum()], inputs=list(img.values()), create_graph=True, only_inputs=True)
    r1 = sum(g.square().sum([1, 2, 3]) for g in grads)
    (torch.nn.functional.softplus(-logits) + r1 * 5).mean().backward()
    for p in D.parameters():
        p.grad = None


def timed(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def census(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ours, other, total = 0.0, {}, 0.0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CUDA:
            continue
        dur = ev.device_time if hasattr(ev, 'device_time') else ev.cuda_time
        total += dur
        if 'p3d::' in ev.name:
            ours += dur
        else:
            r = other.setdefault(ev.name[:100], [0, 0.0])
            r[0] += 1; r[1] += dur
    vendor = {k: v for k, v in other.items() if any(t in k for t in VENDOR)}
    return dict(kernel_ms=total / 1e3, p3d_ms=ours / 1e3, other=other, vendor=vendor)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if len(args) > 0 else 4
    nrr = int(args[1]) if len(args) > 1 else 64
    conv2d_gradfix.enabled = True                                              # training_loop.py:281
    G, D, ws, c, real = build(n, nrr)
    report = {'batch': n, 'nrr': nrr}
    for name, fn in (('G', lambda: g_step(G, ws, c, nrr)), ('Dboth', lambda: d_step(D, real, c))):
        c0 = dict(conv2d_gradfix.native_calls)
        ms = timed(fn)
        torch.cuda.reset_peak_memory_stats()
        cen = census(fn) if '--no-census' not in sys.argv else dict(kernel_ms=0.0, p3d_ms=0.0, other={}, vendor={})     # (under rocprofv3 the two tracers collide)
        calls = {k: conv2d_gradfix.native_calls[k] - c0[k] for k in c0}
        print(f'{name} pass, batch {n}: {ms:.1f} ms wall ({n / ms * 1e3:.1f} img/s), kernels {cen["kernel_ms"]:.1f} ms of which p3d:: {cen["p3d_ms"]:.1f} ms; '
              f'conv calls over 6 passes {calls}; peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB')
        print('  vendor conv / GEMM kernels:', {k: (v[0], round(v[1] / 1e3, 3)) for k, v in cen['vendor'].items()} or 'none')
        for k, (cnt, t) in sorted(cen['other'].items(), key=lambda kv: -kv[1][1])[:14]:
            print(f'  {cnt:4d} x {t / max(cnt, 1):8.1f} us = {t / 1e3:7.3f} ms  {k}')
        report[name] = dict(ms=round(ms, 2), img_per_s=round(n / ms * 1e3, 2), kernel_ms=round(cen['kernel_ms'], 2), p3d_ms=round(cen['p3d_ms'], 2),
                            vendor={k: [v[0], round(v[1] / 1e3, 3)] for k, v in cen['vendor'].items()}, aten_conv_calls=calls['aten'])
    if '--json' in sys.argv:
        json.dump(report, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
