"""Per phase of the six-phase training iteration: host time to ENQUEUE the phase (perf_counter between phase boundaries, no synchronisation inside the iteration)
against the device time between the same boundaries (events).  A phase whose host time reaches its device time is host-bound: the queue runs dry."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix as cg
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod


class A: pass
args = A(); args.dataset, args.batch, args.train_nrr = 'seg2cat', 4, 128
dev = torch.device('cuda', 0)
cg.enabled = True
rmod.fused_policy = 'require'
st = bench.train_setup(args, dev, 1)
for _ in range(2):
    bench.train_iteration(st, {})
torch.cuda.synchronize()
names = [ph['name'] for ph in st['phases']]
host = {n: [] for n in names + ['finish:' + n for n in names]}
orig_finish = bench._finish_phase


def timed_finish(st_, ph):
    t = time.perf_counter(); out = orig_finish(st_, ph); host['finish:' + ph['name']].append((time.perf_counter() - t) * 1e3); return out


bench._finish_phase = timed_finish
dev_ms = {n: [] for n in names}
for it in range(4):
    timers = {}
    marks = []
    orig_mark_time = time.perf_counter()
    # host timestamps at the same boundaries train_iteration marks with events: wrap accumulate_gradients
    loss = st['loss']; orig_acc = loss.accumulate_gradients
    stamps = []
    def acc(**kw):
        stamps.append((kw['phase'], time.perf_counter())); return orig_acc(**kw)
    loss.accumulate_gradients = acc
    t_begin = time.perf_counter()
    bench.train_iteration(st, timers)
    t_end = time.perf_counter()
    loss.accumulate_gradients = orig_acc
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    starts = [s for _, s in stamps] + [t_end]
    for k, n in enumerate(names):
        host[n].append((starts[k + 1] - starts[k]) * 1e3)
    keys = ['t0'] + names
    for a, b in zip(keys[:-1], keys[1:]):
        dev_ms[b].append(timers[a][0].elapsed_time(timers[b][0]))
    print(f'iteration {it}: host enqueue {1e3 * (t_end - t_begin):.1f} ms, then waited {1e3 * (t_sync - t_end):.1f} ms for the device', flush=True)
print(f'{"phase":16s} {"host ms":>9s} {"of which finish":>16s} {"device ms":>10s}')
for n in names:
    print(f'{n:16s} {np.median(host[n]):9.1f} {np.median(host["finish:" + n]):16.1f} {np.median(dev_ms[n]):10.1f}')
