"""Weight-gradient kernel (p3d_conv2d_bwd_weight) at the layer shapes of a training iteration: time and TFLOP/s per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd.torch_utils.ops.conv2d_gradfix import _Cfg, _weight_grad_impl
SHAPES = [  # (dtype, ci, co, h, k, stride) : x [4, ci, h, h] -> y [4, co, h/stride, h/stride]
    ('f16', 64, 64, 512, 3, 1), ('f16', 64, 128, 513, 3, 2), ('f16', 128, 128, 256, 3, 1), ('f16', 128, 256, 257, 3, 2), ('f16', 256, 256, 128, 3, 1),
    ('f16', 256, 512, 129, 3, 2), ('f16', 512, 512, 64, 3, 1), ('f16', 256, 256, 256, 3, 1), ('f16', 128, 128, 512, 3, 1), ('f16', 64, 128, 256, 1, 1),
    ('f32', 512, 512, 64, 3, 1), ('f32', 256, 256, 128, 3, 1), ('f32', 128, 128, 256, 3, 1), ('f32', 512, 512, 32, 3, 1), ('f32', 512, 512, 16, 3, 1),
    ('f32', 64, 64, 512, 3, 1), ('f32', 128, 128, 256, 3, 2)]
for dt, ci, co, h, k, stride in SHAPES:
    dtype = torch.float16 if dt == 'f16' else torch.float32
    n = 4
    pad = k // 2 if stride == 1 else 0
    oh = h if stride == 1 else (h - 3) // 2 + 1
    x = torch.randn(n, ci, h, h, device='cuda', dtype=dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, co, oh, oh, device='cuda', dtype=dtype).contiguous(memory_format=torch.channels_last)
    cfg = _Cfg(False, (co, ci, k, k), stride, pad, 0, 1, 1)
    for _ in range(2): _weight_grad_impl(gy, x, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _weight_grad_impl(gy, x, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * n * ci * co * k * k * oh * oh
    print(f'{dt} ci {ci:4d} co {co:4d} h {h:4d} k {k} s {stride}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s')
