"""Timing of p3d_up2_fir_f16 at the two SR-head shapes (P3D_UP2_DEBUG: 1 = K loop only, 2 = epilogue only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
for ci, co, h in ((32, 256, 128), (256, 128, 256)):
    n = 4
    x = torch.randn(n, ci, h, h, device='cuda').half().contiguous(memory_format=torch.channels_last)
    wmod = modconv.modulate_weights(torch.randn(co, ci, 3, 3, device='cuda'), torch.randn(n, ci, device='cuda') + 1)
    bias = torch.randn(co, device='cuda'); nz = torch.randn(2 * h, 2 * h, device='cuda'); ns = torch.tensor(0.1, device='cuda')
    taps = modconv._separable_fir(f)
    for fused in (True, False):
        def run():
            if fused:
                return modconv.up2_fir(x, wmod, taps, bias, nz, ns, 1, 1.41, 256.0)
            y = modconv.conv2d(x, wmod, transposed=True)
            return modconv.fir4_bias_act(y, f, bias, nz, ns, 'lrelu', 1.41, 256.0)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * n * ci * co * 9 * h * h
        print(f'ci {ci} co {co} h {h} fused {fused}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.0f} TF/s (debug {os.environ.get("P3D_UP2_DEBUG", "0")})')
