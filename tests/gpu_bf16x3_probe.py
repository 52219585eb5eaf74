"""bf16x3 (three bf16 MFMAs per fp32 product) vs the exact fp32 MFMA kernels on the backbone's five M >= 4096 layers: speed, and error
against an fp64 convolution of the same fp32 operands (max |d| / max |ref|, and rms(d) / rms(ref))."""
import os, sys
import torch
import torch.nn.functional as F
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
torch.manual_seed(0)
for name, ci, co, r, tr in [('bb.b64.conv1', 512, 512, 64, False), ('bb.b128.conv1', 256, 256, 128, False), ('bb.b256.conv1', 128, 128, 256, False),
                            ('bb.b128.conv0 T2', 512, 256, 64, True), ('bb.b256.conv0 T2', 256, 128, 128, True)]:
    x = torch.randn(N, ci, r, r, device='cuda').to(memory_format=torch.channels_last)
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(N, ci, device='cuda') + 1
    w32 = modconv.modulate_weights(weight, styles, dtype=torch.float32)
    w3 = modconv.modulate_weights(weight, styles, dtype=modconv.BF16X3)
    fl = 2 * N * ci * co * 9 * r * r
    t32 = timeit(lambda: modconv.conv2d(x, w32, transposed=tr))
    t3 = timeit(lambda: modconv.conv2d(x, w3, transposed=tr, split=True))
    y32 = modconv.conv2d(x, w32, transposed=tr)
    y3 = modconv.conv2d(x, w3, transposed=tr, split=True)
    # fp64 reference of image 0 on the CPU (same fp32 operand values)
    wq = w32[0].double().reshape(co, 3, 3, ci).permute(0, 3, 1, 2).cpu()
    x0 = x[:1].double().cpu()
    ref = F.conv_transpose2d(x0, wq.transpose(0, 1), stride=2) if tr else F.conv2d(x0, wq, padding=1)
    def err(y):
        d = y[:1].double().cpu() - ref
        return (d.abs().max() / ref.abs().max()).item(), (d.square().mean().sqrt() / ref.square().mean().sqrt()).item()
    e32, e3 = err(y32), err(y3)
    print(f'{name}: fp32 {fl / t32 / 1e12:6.1f} TF ({t32 * 1e3:.3f} ms) err max {e32[0]:.2e} rms {e32[1]:.2e} | bf16x3 {fl / t3 / 1e12:6.1f} TF-equivalent ({t3 * 1e3:.3f} ms) '
          f'err max {e3[0]:.2e} rms {e3[1]:.2e} | speed-up {t32 / t3:.2f}x', flush=True)
