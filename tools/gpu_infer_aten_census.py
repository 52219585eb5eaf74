"""Which Python lines issue the ATen launches that are left in ONE inference step (bench.py's G.synthesis, seg2cat batch 4): torch.profiler with stacks, CPU-side op
records, grouped by (op, innermost frames inside this repository).   python tools/gpu_infer_aten_census.py -> gpurun_out/infer_aten_census.txt"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix as cg
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod

args = argparse.Namespace(dataset='seg2cat', depth=128, batch=4)
dev = torch.device('cuda', 0)
cg.enabled = True
rmod.fused_policy = 'require'
G, kw, info, ws, c = bench.build(args, dev)
G, ws, c = G.to(dev), ws.to(dev), c.to(dev)


def step():
    with torch.no_grad():
        return G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=info['nrr'])


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
DEPTH = int(os.environ.get('P3D_CENSUS_DEPTH', '3'))
SKIP = ('aten::empty', 'aten::empty_strided', 'aten::empty_like', 'aten::view', 'aten::reshape', 'aten::as_strided', 'aten::permute', 'aten::select', 'aten::slice', 'aten::narrow',
        'aten::unsqueeze', 'aten::squeeze', 'aten::expand', 'aten::t', 'aten::transpose', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::item', 'aten::_local_scalar_dense',
        'aten::is_nonzero', 'aten::resize_', 'aten::result_type', 'aten::lift_fresh', 'aten::to', 'aten::contiguous', 'aten::clone', 'aten::_to_copy', 'aten::zeros', 'aten::zeros_like',
        'aten::ones', 'aten::full', 'aten::rand', 'aten::rand_like', 'aten::repeat', 'aten::unfold', 'aten::flatten', 'aten::view_as', 'aten::unbind', 'aten::split', 'aten::chunk', 'aten::stride',
        'aten::is_contiguous', 'aten::numel', 'aten::size', 'aten::dim', 'aten::set_', 'aten::record_stream', 'aten::cudnn_is_acceptable', 'aten::new_empty', 'aten::new_zeros', 'aten::expand_as')
count, shapes = collections.Counter(), {}
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.name in SKIP:
        continue
    frames = [fr.replace(ROOT + '/', '').replace('pix2pix3d_amd/', '') for fr in (ev.stack or []) if ('pix2pix3d_amd' in fr or 'bench.py' in fr) and 'site-packages' not in fr]
    where = ' < '.join(frames[:DEPTH]) if frames else '<no stack>'
    count[(ev.name, where)] += 1
    shapes.setdefault((ev.name, where), str(getattr(ev, 'input_shapes', ''))[:120])
rows = sorted(count.items(), key=lambda kv: -kv[1])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'infer_aten_census.txt'), 'w') as f:
    f.write(f'# ATen ops that launch (or may launch) a kernel in one G.synthesis step, by issuing line: {sum(count.values())} ops\n')
    for (op, where), n in rows:
        f.write(f'{n:4d}  {op:24s} {where[:400]}   {shapes[(op, where)]}\n')
print(open(os.path.join(ROOT, 'gpurun_out', 'infer_aten_census.txt')).read()[:8000])
