"""Which ATen operators (with shapes) an inference G.synthesis step still launches: torch profiler, grouped by op + input shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2pix3d_amd import configs, dnnlib
from torch.profiler import profile, ProfilerActivity
kw = configs.generator_kwargs('seg2cat', depth=(64, 64))
rk = kw['rendering_kwargs']
torch.manual_seed(0)
G = dnnlib.util.construct_class_by_name(**kw).cuda().eval().requires_grad_(False)
n = 4
ws = torch.randn(n, G.backbone.num_ws, 512, device='cuda')
c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32, device='cuda')
fn = lambda: G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=128)
with torch.no_grad():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        fn(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
    t = getattr(e, 'self_device_time_total', None)
    if t is None: t = e.self_cuda_time_total
    if t > 0 and e.key.startswith('aten::'):
        st = [s for s in e.stack if 'pix2pix3d_amd' in s][:2]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90], ' <- '.join(x.split('pix2pix3d_amd/')[-1][:60] for x in st)))
rows.sort(reverse=True)
print('total aten self device us', sum(r[0] for r in rows))
for t, cnt, k, sh, st in rows[:30]:
    print(f'{t:8.1f} us {cnt:3d} {k:18s} {sh}  {st}')
