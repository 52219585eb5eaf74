import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
N = 4
for name, ci, co, r, tr in [('sr.b0.conv1', 256, 256, 256, False), ('sr.b1.conv1', 128, 128, 512, False), ('sr.b1.conv0 T2', 256, 128, 256, True), ('sr.b0.conv0 T2', 32, 256, 128, True)]:
    x = torch.randn(N, ci, r, r, device='cuda').half().to(memory_format=torch.channels_last)
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles)
    bias = torch.randn(co, device='cuda')
    fl = 2 * N * ci * co * 9 * r * r
    t = timeit(lambda: modconv.conv3x3(x, wmod, transposed=tr, bias=None if tr else bias, act=0 if tr else 1, gain=1.414, clamp=-1 if tr else 256))
    print(f'p3d {name}: {fl / t / 1e12:.1f} TF ({t * 1e3:.3f} ms)', flush=True)

for name, ci, co, r, tr in [('bb.b64.conv1', 512, 512, 64, False), ('bb.b128.conv1', 256, 256, 128, False), ('bb.b256.conv1', 128, 128, 256, False),
                            ('bb.b128.conv0 T2', 512, 256, 64, True), ('bb.b256.conv0 T2', 256, 128, 128, True)]:
    x = torch.randn(N, ci, r, r, device='cuda').to(memory_format=torch.channels_last)
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles, dtype=torch.float32)
    bias = torch.randn(co, device='cuda')
    fl = 2 * N * ci * co * 9 * r * r
    t = timeit(lambda: modconv.conv2d(x, wmod, transposed=tr, bias=None if tr else bias, act=0 if tr else 1, gain=1.414))
    print(f'p3d fp32 {name}: {fl / t / 1e12:.1f} TF ({t * 1e3:.3f} ms)', flush=True)

from pix2pix3d_amd.torch_utils.ops import upfirdn2d as _up
f4 = _up.setup_filter([1, 3, 3, 1], device='cuda')
for name, c, r, dt in [('sr.b1 fir 512^2x128 f16', 128, 512, torch.float16), ('sr.b0 fir 256^2x256 f16', 256, 256, torch.float16), ('bb.b256 fir 256^2x128 f32', 128, 256, torch.float32)]:
    y = torch.randn(N, c, r + 1, r + 1, device='cuda').to(dt).to(memory_format=torch.channels_last)
    bias = torch.randn(c, device='cuda'); nz = torch.randn(r, r, device='cuda'); ns = torch.full([], 0.1, device='cuda')
    t = timeit(lambda: modconv.fir4_bias_act(y, f4, bias, nz, ns, 'lrelu', 1.414, 256.0))
    es = y.element_size()
    gb = N * c * ((r + 1) ** 2 + r * r) * es / 1e9
    print(f'p3d {name}: {t * 1e3:.3f} ms, {gb / t / 1e3:.2f} TB/s (algorithmic read+write)', flush=True)

for name, ci, r in [('bb.b256.torgb 1x1 128->96', 128, 256), ('bb.b128.torgb 1x1 256->96', 256, 128), ('bb.b64.torgb 1x1 512->96', 512, 64)]:
    x = torch.randn(N, ci, r, r, device='cuda').to(memory_format=torch.channels_last)
    weight = torch.randn(96, ci, 1, 1, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles, demodulate=False, dtype=torch.float32)
    bias = torch.randn(96, device='cuda')
    t = timeit(lambda: modconv.conv2d(x, wmod, bias=bias))
    gb = N * r * r * (ci + 96) * 4 / 1e9
    print(f'p3d fp32 {name}: {t * 1e3:.3f} ms, {2 * N * ci * 96 * r * r / t / 1e12:.1f} TF, {gb / t / 1e3:.2f} TB/s (x read + y write)', flush=True)
