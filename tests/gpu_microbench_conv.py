"""What the vendor conv path (ATen -> MIOpen) delivers for the SR / backbone layer shapes in different formulations.
Run on the GPU box.  Prints TFLOP/s per case."""
import torch, sys, os
torch.backends.cudnn.benchmark = True

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

N = 4
cases = [('sr.b0.conv1', 256, 256, 256, torch.float16), ('sr.b1.conv1', 128, 128, 512, torch.float16),
         ('bb.b64.conv1', 512, 512, 64, torch.float32), ('bb.b128.conv1', 256, 256, 128, torch.float32), ('bb.b256.conv1', 128, 128, 256, torch.float32)]
for name, ci, co, r, dt in cases:
    fl = 2 * N * ci * co * 9 * r * r
    x = torch.randn(N, ci, r, r, device='cuda', dtype=dt)
    w = torch.randn(co, ci, 3, 3, device='cuda', dtype=dt)
    wg = torch.randn(N * co, ci, 3, 3, device='cuda', dtype=dt)
    xg = x.reshape(1, N * ci, r, r)
    res = {}
    res['grouped NCHW'] = timeit(lambda: torch.nn.functional.conv2d(xg, wg, padding=1, groups=N))
    res['plain NCHW'] = timeit(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    xc, wc = x.to(memory_format=torch.channels_last), w.to(memory_format=torch.channels_last)
    res['plain NHWC'] = timeit(lambda: torch.nn.functional.conv2d(xc, wc, padding=1))
    xgc, wgc = xg.to(memory_format=torch.channels_last), wg.to(memory_format=torch.channels_last)
    res['grouped NHWC'] = timeit(lambda: torch.nn.functional.conv2d(xgc, wgc, padding=1, groups=N))
    if dt == torch.float32:
        xh, wh = xc.half(), wc.half()
        res['plain NHWC fp16'] = timeit(lambda: torch.nn.functional.conv2d(xh, wh, padding=1))
    print(name, dt, ' | '.join(f'{k}: {fl / t / 1e12:.1f} TF ({t * 1e3:.2f} ms)' for k, t in res.items()), flush=True)
# transposed stride-2
for name, ci, co, r, dt in [('sr.b1.conv0 T2', 256, 128, 256, torch.float16), ('bb.b256.conv0 T2', 256, 128, 128, torch.float32)]:
    fl = 2 * N * ci * co * 9 * r * r
    x = torch.randn(N, ci, r, r, device='cuda', dtype=dt)
    w = torch.randn(ci, co, 3, 3, device='cuda', dtype=dt)
    wg = torch.randn(N * ci, co, 3, 3, device='cuda', dtype=dt)
    res = {}
    res['grouped NCHW'] = timeit(lambda: torch.nn.functional.conv_transpose2d(x.reshape(1, N * ci, r, r), wg, stride=2, groups=N))
    res['plain NCHW'] = timeit(lambda: torch.nn.functional.conv_transpose2d(x, w, stride=2))
    xc, wc = x.to(memory_format=torch.channels_last), w.to(memory_format=torch.channels_last)
    res['plain NHWC'] = timeit(lambda: torch.nn.functional.conv_transpose2d(xc, wc, stride=2))
    print(name, dt, ' | '.join(f'{k}: {fl / t / 1e12:.1f} TF ({t * 1e3:.2f} ms)' for k, t in res.items()), flush=True)

# ---- this repo's MFMA implicit-GEMM kernel on the SR shapes ----
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv
for name, ci, co, r, tr in [('sr.b0.conv1', 256, 256, 256, False), ('sr.b1.conv1', 128, 128, 512, False), ('sr.b1.conv0 T2', 256, 128, 256, True), ('sr.b0.conv0 T2', 32, 256, 128, True)]:
    x = torch.randn(N, ci, r, r, device='cuda').half().to(memory_format=torch.channels_last)
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles)
    bias = torch.randn(co, device='cuda')
    fl = 2 * N * ci * co * 9 * r * r
    t = timeit(lambda: modconv.conv3x3(x, wmod, transposed=tr, bias=None if tr else bias, act=0 if tr else 1, gain=1.414, clamp=-1 if tr else 256))
    tm = timeit(lambda: modconv.modulate_weights(weight, styles))
    print(f'p3d {name}: {fl / t / 1e12:.1f} TF ({t * 1e3:.3f} ms)   modulate_weights {tm * 1e6:.0f} us', flush=True)
