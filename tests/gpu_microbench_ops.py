"""Quick HBM-roofline probe for the streaming ops (run on the GPU box; prints one line per case)."""
import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import bias_act, upfirdn2d

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

for dt in (torch.float32, torch.float16):
    x = torch.randn(4, 128, 512, 512, device='cuda', dtype=dt); b = torch.randn(128, device='cuda', dtype=dt)
    t = timeit(lambda: bias_act.bias_act(x, b, act='lrelu', clamp=256))
    print(f'bias_act {dt} {x.numel()*x.element_size()*2/t/1e12:.2f} TB/s  {t*1e6:.0f} us')
    t = timeit(lambda: torch.nn.functional.leaky_relu(x, 0.2))
    print(f'  torch leaky_relu same bytes {x.numel()*x.element_size()*2/t/1e12:.2f} TB/s')
    f = upfirdn2d.setup_filter([1,3,3,1], device='cuda')
    xi = torch.randn(4, 128, 513, 513, device='cuda', dtype=dt)
    t = timeit(lambda: upfirdn2d.upfirdn2d(xi, f, padding=[1,1,1,1], gain=4))
    print(f'upfirdn2d up1 {dt} {(xi.numel()+4*128*512*512)*xi.element_size()/t/1e12:.2f} TB/s {t*1e6:.0f} us')
    xs = torch.randn(4, 96, 128, 128, device='cuda', dtype=dt)
    t = timeit(lambda: upfirdn2d.upsample2d(xs, f))
    print(f'upsample2d {dt} {(xs.numel()*5)*xs.element_size()/t/1e12:.2f} TB/s {t*1e6:.0f} us')
