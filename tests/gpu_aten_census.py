"""Which Python lines issue the ATen glue launches of one six-phase training iteration (bench.train_setup / train_iteration): torch.profiler with stacks,
CPU-side op records only, grouped by (op, innermost frame inside this repository).  Backward ops run on the autograd thread without a Python stack; they are
listed by op name alone ('<autograd>').
    python tests/gpu_aten_census.py -> gpurun_out/aten_census.txt"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    bench.train_iteration(st, {})
    torch.cuda.synchronize()
DEPTH = int(os.environ.get('P3D_CENSUS_DEPTH', '1'))
GLUE = ('aten::mul', 'aten::mul_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::sum', 'aten::div', 'aten::div_', 'aten::neg', 'aten::sub',
        'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::cat', 'aten::pow', 'aten::sqrt', 'aten::rsqrt', 'aten::where', 'aten::clamp', 'aten::square',
        'aten::addcmul_', 'aten::lerp_', 'aten::zeros', 'aten::zeros_like', 'aten::empty_like', 'aten::index', 'aten::mean', 'aten::exp', 'aten::sigmoid')
LEAF = {'aten::mul', 'aten::mul_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::sum', 'aten::div', 'aten::div_', 'aten::neg', 'aten::sub', 'aten::cat',
        'aten::pow', 'aten::sqrt', 'aten::rsqrt', 'aten::where', 'aten::clamp', 'aten::addcmul_', 'aten::lerp_', 'aten::index', 'aten::mean', 'aten::exp', 'aten::sigmoid',
        'aten::sub_', 'aten::addcdiv_', 'aten::clamp_', 'aten::clamp_min', 'aten::gt', 'aten::lt', 'aten::softplus', 'aten::softplus_backward', 'aten::abs', 'aten::sgn',
        'aten::maximum', 'aten::minimum', 'aten::argmax', 'aten::scatter_', 'aten::nan_to_num', 'aten::nan_to_num_', 'aten::normal_', 'aten::uniform_', 'aten::bernoulli_'}
count = collections.Counter()
samples = {}
for ev in prof.events():
    if ev.name not in LEAF:
        continue
    frames = [fr.replace(ROOT + '/', '').replace('pix2pix3d_amd/', '') for fr in (ev.stack or []) if ('pix2pix3d_amd' in fr or 'bench.py' in fr) and 'site-packages' not in fr]
    where = ' < '.join(frames[:DEPTH]) if frames else '<autograd>'
    count[(ev.name, where)] += 1
    samples.setdefault((ev.name, where), [fr.replace(ROOT + '/', '') for fr in (ev.stack or [])][:14])
rows = sorted(count.items(), key=lambda kv: -kv[1])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'aten_census.txt'), 'w') as f:
    f.write(f'# ATen element-wise / copy / reduce ops of one training iteration by issuing line: {sum(count.values())} ops\n')
    byop = collections.Counter()
    for (op, _), n in rows:
        byop[op] += n
    f.write('# by op: ' + ', '.join(f'{op} {n}' for op, n in byop.most_common()) + '\n')
    for (op, where), n in rows[:150]:
        f.write(f'{n:6d}  {op:22s} {where[:300]}\n')
    f.write('# one raw stack per entry (first 40 entries)\n')
    for (op, where), n in rows[:40]:
        f.write(f'## {n} {op} {where[:120]}\n   ' + '\n   '.join(samples[(op, where)]) + '\n')
print(open(os.path.join(ROOT, 'gpurun_out', 'aten_census.txt')).read()[:6000])
