"""Timing of p3d_torgb_nhwc_f16 at the SR heads' shapes (256 channels @256^2, 128 @512^2; batch 4, 3 output channels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv
for ci, h in ((256, 256), (128, 512), (64, 512)):
    n, co = 4, 3
    x = torch.randn(n, ci, h, h, device='cuda').half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device='cuda'); s = torch.randn(n, ci, device='cuda') + 1; b = torch.randn(co, device='cuda')
    out = torch.zeros(n, co, h, h, device='cuda')
    ref = (torch.einsum('oc,nc,nchw->nohw', w.reshape(co, ci).double(), s.double(), x.double()) + b.double().reshape(1, co, 1, 1)).clamp(-256, 256)
    y = modconv.torgb(x, w, s, b, clamp=256.0)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    for _ in range(3): modconv.torgb(x, w, s, b, clamp=256.0, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): modconv.torgb(x, w, s, b, clamp=256.0, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'torgb ci {ci} @ {h}^2 x{n}: {us:.1f} us  {x.numel() * 2 / us / 1e6:.2f} TB/s of x  (max err / range {err:.1e})')
