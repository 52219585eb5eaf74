"""Why some optimizer steps of the training iteration run per-tensor kernels (lerp_ / addcmul_) instead of the multi-tensor ones: per phase, the op names of
opt.step() and every (param, grad, state) triple whose strides / dtypes / density would fail the foreach fast route."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix as cg
from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
from pix2pix3d_amd import dp
from torch.profiler import profile, ProfilerActivity


class A: pass
args = A(); args.dataset, args.batch, args.train_nrr = 'seg2cat', 4, 128
dev = torch.device('cuda', 0)
cg.enabled = True
rmod.fused_policy = 'require'
st = bench.train_setup(args, dev, 1)
bench.train_iteration(st, {})
torch.cuda.synchronize()
for ph in st['phases']:
    ph['opt'].zero_grad(set_to_none=True)
    ph['module'].requires_grad_(True)
    st['loss'].accumulate_gradients(phase=ph['name'], batch=st['batch'], gen_z=ph['gen_z'], gen_c=ph['gen_c'], gain=ph['interval'], cur_nimg=200000)
    ph['module'].requires_grad_(False)
    dp.allreduce_gradients(ph['module'], world_size=1, out=st['flat'])
    opt = ph['opt']
    bad = []
    n = 0
    for g in opt.param_groups:
        for p in g['params']:
            if p.grad is None:
                continue
            n += 1
            s = opt.state.get(p, {})
            for nm, t in (('grad', p.grad), ('exp_avg', s.get('exp_avg')), ('exp_avg_sq', s.get('exp_avg_sq'))):
                if t is None:
                    continue
                if t.stride() != p.stride() or t.dtype != p.dtype or not t.is_contiguous() or t.device != p.device or t.layout != p.layout:
                    bad.append((nm, tuple(p.shape), p.stride(), t.stride(), str(t.dtype), t.is_contiguous()))
            if type(p) not in (torch.Tensor, torch.nn.Parameter):
                bad.append(('type', type(p)))
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        opt.step()
        torch.cuda.synchronize()
    c = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith('aten::') and ev.name not in ('aten::empty', 'aten::empty_strided', 'aten::as_strided', 'aten::item', 'aten::_local_scalar_dense', 'aten::select', 'aten::view', 'aten::detach', 'aten::is_nonzero'))
    print(ph['name'], 'params with grad', n, 'defaults', {k: opt.defaults.get(k) for k in ('foreach', 'fused', 'capturable', 'differentiable')}, flush=True)
    print('   ops:', dict(c.most_common(14)), flush=True)
    print('   suspicious:', len(bad), bad[:6], flush=True)
