"""The backbone's large bf16x3 layers on split activations (modconv.SplitActs), timed one by one: 3x3 (ring / halo kernels) and the x2
transposed form (generic kernel).  P3D_CONV_NO_R2=1: the 3x3 layers on the halo-slab kernel instead of the ring kernel."""
import os, sys, torch
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
for name, ci, co, r, tr in [('b64.conv1', 512, 512, 64, False), ('b128.conv1', 256, 256, 128, False), ('b256.conv1', 128, 128, 256, False),
                            ('b128.conv0 T2', 512, 256, 64, True), ('b256.conv0 T2', 256, 128, 128, True)]:
    x = torch.randn(N, ci, r, r, device='cuda').to(memory_format=torch.channels_last)
    w3 = modconv.modulate_weights(torch.randn(co, ci, 3, 3, device='cuda'), torch.randn(N, ci, device='cuda') + 1, dtype=modconv.BF16X3)
    v = x.permute(0, 2, 3, 1).reshape(N, r, r, ci // 32, 32)
    hi = v.to(torch.bfloat16)
    rows = torch.stack([hi, (v - hi.float()).to(torch.bfloat16)], dim=-2).reshape(N, r, r, ci // 32, 64).view(torch.float32).reshape(N, r, r, ci)
    xs = modconv.SplitActs(rows.permute(0, 3, 1, 2))                               # x in the [32 hi | 32 lo] K-row layout
    fl = 2 * N * ci * co * 9 * r * r
    t_plain = timeit(lambda: modconv.conv2d(x, w3, transposed=tr, split=True))
    t_split = timeit(lambda: modconv.conv2d(xs, w3, transposed=tr, split=True, out_split=not tr))
    print(f'{name}: plain input {fl / t_plain / 1e12:6.1f} TF-eq ({t_plain * 1e3:.3f} ms) | split input {fl / t_split / 1e12:6.1f} TF-eq ({t_split * 1e3:.3f} ms)', flush=True)
