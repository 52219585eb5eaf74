"""Where the convolution time of the six-phase training iteration goes, by geometry: every call of conv2d_gradfix's native forward / data-gradient /
weight-gradient entry is bracketed with events (one synchronisation at the end), keyed by (kind, dtype, N, Ci, Co, k, stride, transposed, H x W).
    python tests/gpu_time_train_convs.py  -> gpurun_out/train_conv_geometries.txt"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix as cg

rec = []
orig_conv, orig_wg = cg._native_conv, cg._native_weight_grad


def timed(kind, key, fn, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(*a); e1.record()
    rec.append((kind, key, e0, e1))
    return out


def conv(x, w, cfg, k, stride):
    n, ci, h, wd = x.shape
    co = w.shape[1] if cfg.transpose else w.shape[0]
    return timed('conv', (str(x.dtype)[6:], n, ci, co, k, stride, int(cfg.transpose), h, wd, int(bool(cfg.split))), orig_conv, x, w, cfg, k, stride)


def wgrad(gy, x, cfg, k, stride, scale=None):
    return timed('wgrad', (str(x.dtype)[6:], x.shape[0], x.shape[1], gy.shape[1], k, stride, int(cfg.transpose), x.shape[2], x.shape[3], 0), lambda *a: orig_wg(*a, scale=scale), gy, x, cfg, k, stride)


class A: pass
args = A(); args.dataset, args.batch, args.train_nrr = 'seg2cat', 4, 128
dev = torch.device('cuda', 0)
cg.enabled = True
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
rmod.fused_policy = 'require'
st = bench.train_setup(args, dev, 1)
for _ in range(2):
    bench.train_iteration(st, {})
torch.cuda.synchronize()
cg._native_conv, cg._native_weight_grad = conv, wgrad
bench.train_iteration(st, {})
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for kind, key, e0, e1 in rec:
    a = agg[(kind,) + key]; a[0] += 1; a[1] += e0.elapsed_time(e1)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'train_conv_geometries.txt'), 'w') as f:
    f.write(f'# convolution calls of one six-phase iteration: {len(rec)} calls, {tot:.1f} ms (event-bracketed, includes weight re-layout / reduce launches of each call)\n')
    f.write('# kind dtype N Ci Co k stride transposed HxW bf16x3 | calls  total ms  avg us  GFLOP/call  TFLOP/s\n')
    for key, (n, ms) in rows[:90]:
        kind, dt, nb, ci, co, k, stride, tr, h, w, sp = key
        pix = h * w if not (tr and stride == 2) else h * w          # MACs are counted on the input grid for the transposed op and on the output grid of a strided one
        if kind == 'conv' and stride == 2 and not tr:
            pix = ((h - 3) // 2 + 1) * ((w - 3) // 2 + 1)
        gf = 2.0 * nb * ci * co * k * k * pix / 1e9
        f.write(f'{kind:6s} {dt:8s} {nb:2d} {ci:4d} {co:4d} {k} s{stride} t{tr} {h:4d}x{w:<4d} b{sp} | {n:4d} {ms:9.2f} {ms / n * 1e3:9.1f} {gf:9.2f} {gf / (ms / n) :8.1f}\n')
print(open(os.path.join(ROOT, 'gpurun_out', 'train_conv_geometries.txt')).read())
