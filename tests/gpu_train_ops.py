"""Which ATen operators (with shapes) a G training pass still spends device time in: torch profiler, grouped by op + input shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_train_census import build, g_step, d_step
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
from torch.profiler import profile, ProfilerActivity
conv2d_gradfix.enabled = True
n, nrr = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 64
G, D, ws, c, real = build(n, nrr)
which = sys.argv[2] if len(sys.argv) > 2 else 'G'
fn = (lambda: g_step(G, ws, c, nrr)) if which == 'G' else (lambda: d_step(D, real, c))
for _ in range(2): fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fn(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None: t = e.self_cuda_time_total
    if t > 0 and e.key.startswith('aten::'):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
print('total aten self device ms', sum(r[0] for r in rows) / 1e3)
for t, cnt, k, sh in rows[:40]:
    print(f'{t / 1e3:8.3f} ms {cnt:4d} {k:28s} {sh}')
