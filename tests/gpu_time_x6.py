"""Times the fp32-accurate (bf16x6) 3x3 'same' convolution of the training-size layers and checks it against the f32-input MFMA kernel.
    P3D_X6_PRESPLIT=0|1 python tests/gpu_time_x6.py     (0: in-register splits, conv3x3_halo_kernel<float, X6>; 1: operands split once per work-group)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops import modconv


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


tag = 'presplit' if os.environ.get('P3D_X6_PRESPLIT', '1') != '0' else 'inreg'
N = 4
torch.manual_seed(0)
for name, ci, co, r, per_img in [('512->512 @64', 512, 512, 64, True), ('256->256 @128', 256, 256, 128, True), ('128->128 @256', 128, 128, 256, True),
                                 ('64->64 @512 (shared)', 64, 64, 512, False), ('128->128 @256 (shared)', 128, 128, 256, False), ('512->512 @32', 512, 512, 32, True),
                                 ('96->128 @256 (shared)', 96, 128, 256, False)]:
    x = torch.randn(N, ci, r, r, device='cuda').to(memory_format=torch.channels_last)
    weight = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
    if per_img:
        wmod = modconv.modulate_weights(weight, torch.randn(N, ci, device='cuda') + 1, dtype=torch.float32)
    else:
        wmod = weight.permute(0, 2, 3, 1).reshape(1, co, 9, ci).contiguous()
    bias = torch.randn(co, device='cuda')
    fl = 2 * N * ci * co * 9 * r * r
    run = lambda: modconv.conv2d(x, wmod, bias=bias, act=1, gain=1.414)
    modconv.f32_x6 = True
    y6 = run(); t = timeit(run)
    modconv.f32_x6 = False
    y1 = run()
    modconv.f32_x6 = True
    err = float((y6 - y1).abs().max() / y1.abs().max())
    print(f'x6 [{tag}] {name}: {fl / t / 1e12:.1f} TF ({t * 1e3:.3f} ms)  max |x6 - f32 mfma| / max |y| = {err:.2e}', flush=True)
