"""Weight-gradient kernel alone on the geometries that carry the training iteration's weight-gradient time (profiles/round4_o_train_conv_geometries.txt),
event-timed over 20 calls each, one row per geometry; the environment (P3D_WGRAD_F32_KP16, P3D_WGRAD_WG_PER_CU, ...) selects the variant.
    python tests/gpu_time_wgrad.py [label]  -> appends to gpurun_out/wgrad_variants.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pix2pix3d_amd import _lib

# (dtype, N, c_small-side image H, cs, cb, k, stride, pad): stride-1 same-size 3x3 layers and the stride-2 down-sampling ones
GEO = [(torch.float32, 4, 128, 256, 256, 3, 1, 1), (torch.float32, 4, 256, 128, 128, 3, 1, 1), (torch.float32, 4, 64, 512, 512, 3, 1, 1),
       (torch.float32, 4, 512, 64, 64, 3, 1, 1), (torch.float32, 4, 32, 512, 512, 3, 1, 1), (torch.float32, 4, 16, 512, 512, 3, 1, 1),
       (torch.float32, 4, 256, 128, 64, 3, 2, 0), (torch.float32, 4, 128, 256, 128, 3, 2, 0), (torch.float32, 4, 64, 512, 256, 3, 2, 0),
       (torch.float16, 4, 256, 128, 64, 3, 2, 0), (torch.float16, 4, 256, 64, 128, 3, 1, 1),
       (torch.float16, 4, 512, 64, 64, 3, 1, 1), (torch.float16, 4, 256, 128, 128, 3, 1, 1), (torch.float16, 4, 128, 256, 256, 3, 1, 1)]
label = sys.argv[1] if len(sys.argv) > 1 else 'default'
L = _lib.lib()
dev = torch.device('cuda', 0)
rows = []
for dt, n, hs, cs, cb, k, stride, pad in GEO:
    hb = hs if stride == 1 else hs * 2 + 1
    g = torch.Generator(device=dev).manual_seed(3)
    small = torch.randn(n, hs, hs, cs, device=dev, generator=g).to(dt)
    big = torch.randn(n, hb, hb, cb, device=dev, generator=g).to(dt)
    gw = torch.empty(cs, cb, k, k, dtype=dt, device=dev)
    code = 4 if (dt == torch.float32 and os.environ.get('WGRAD_X6') == '1') else _lib.DTYPE_CODE[dt]      # 4 = P3D_F32_BF16X6
    nbytes = int(L.p3d_conv2d_bwd_weight_workspace(code, n, hs, hs, cs, cb, k))
    work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    call = lambda: _lib.check(L.p3d_conv2d_bwd_weight(_lib.ptr(small), _lib.ptr(big), _lib.ptr(gw), _lib.ptr(work), nbytes, code, n, hs, hs, cs, hb, hb, cb, k, stride, pad,
                                                      _lib.stream_of(gw)), 'wgrad')
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tf = 2.0 * n * hs * hs * cs * cb * k * k / us / 1e6
    rows.append(f'{label:12s} {str(dt)[6:]:8s} N{n} {hs:3d}^2 {cs:3d}x{cb:<3d} k{k} s{stride} | {us:8.1f} us {tf:7.1f} TFLOP/s  sum {float(gw.float().abs().sum()):.6e}')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'wgrad_variants.txt'), 'a') as f:
    f.write('\n'.join(rows) + '\n')
print('\n'.join(rows))
