"""Why the discriminator test does not 'move the seed': for input seeds 31..45 of the `dual` record, the smallest |pre-activation| / max over every
leaky-ReLU layer of the reference discriminator (run on the CPU from /root/reference).  Every seed has units at 4e-8..5e-7 of the range in the
262 144-element layers — a perturbation of 5e-6 (the bf16x3 class) flips a few of them whatever the seed.  Authoring container only.

    python tests/golden/seed_search_discriminator.py
"""
import sys, os
sys.path.insert(0, '/root/reference')
import numpy as np, torch, importlib.util
torch.set_num_threads(8)
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('p3d_weights', os.path.join(HERE, 'weights.py')); weights = importlib.util.module_from_spec(spec); spec.loader.exec_module(weights)
import dnnlib
from torch_utils.ops import bias_act as BA
rec = []
orig = BA.bias_act
def hooked(x, b=None, dim=1, act='linear', **k):
    if act == 'lrelu':
        with torch.no_grad():
            v = x if b is None else x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)]).to(x.dtype)
            rec.append((v.numel(), float(v.abs().min()), float(v.abs().max())))
    return orig(x, b=b, dim=dim, act=act, **k)
BA.bias_act = hooked
import training.networks_stylegan2 as NS
NS.bias_act.bias_act = hooked
kw = dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=64, img_channels=3, channel_base=1024,
          channel_max=32, num_fp16_res=0, conv_clamp=None, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2))
torch.manual_seed(0)
D = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(False)
weights.seed_module(D, seed=9)
for seed in range(31, 46):
    gz = torch.Generator().manual_seed(seed)
    img = dict(image=torch.randn(4, 3, 64, 64, generator=gz), image_raw=torch.randn(4, 3, 16, 16, generator=gz))
    c = torch.randn(4, 25, generator=gz)
    rec.clear()
    with torch.no_grad():
        D(img, c)
    worst = min(r[1] / r[2] for r in rec)
    small = min(r[1] / r[2] for r in rec if r[0] <= 40000)
    print(seed, f'worst {worst:.2e} small-layer worst {small:.2e}', [(r[0], f'{r[1]/r[2]:.1e}') for r in rec])
