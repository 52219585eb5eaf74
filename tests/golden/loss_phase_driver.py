"""Drives ``Pix2Pix3DLoss.accumulate_gradients`` phase by phase the way training_loop.py:514-529 does — for ANY implementation of the
loss / networks: the golden recorder runs it on the reference's training/loss.py + the reference's modules, the tests run it on (a) the
reference's training/loss.py over this package's mirrors (dropin) and (b) this package's restatement of the loss.  torch + numpy only;
loaded by path (tests/golden/make_golden.py must not import this repository's package).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, fname):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, fname))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


det_rng = _load('p3d_det_rng', 'det_rng.py')


def lpips_standin(a, b):
    """Differentiable stand-in for lpips.LPIPS(net='vgg') (a VGG with downloaded weights: neither the package nor the weights exist here), shaped like
    the real thing ([N,1,1,1]): mean squared difference of the 4x-average-pooled images."""
    pa, pb = torch.nn.functional.avg_pool2d(a.float(), 4), torch.nn.functional.avg_pool2d(b.float(), 4)
    return (pa - pb).square().mean(dim=[1, 2, 3], keepdim=True)


def install_lpips_stub():
    """training/loss.py:20 does ``import lpips``: give it a module whose LPIPS is the stand-in."""
    class LPIPS(torch.nn.Module):
        def __init__(self, net='vgg', **kw):
            super().__init__()

        def forward(self, a, b):
            return lpips_standin(a, b)
    mod = types.ModuleType('lpips')
    mod.LPIPS, mod.lpips_standin = LPIPS, lpips_standin
    sys.modules['lpips'] = mod
    return mod


# (phase, the network it updates, gain = the phase's interval: training_loop.py:360-373 with G_reg_interval 4, D_reg_interval 16)
LOSS_PHASES = [('Gmain', 'G', 1), ('Greg', 'G', 4), ('Dmain', 'D', 1), ('Dreg', 'D', 16), ('D_semanticmain', 'D_semantic', 1), ('D_semanticreg', 'D_semantic', 16)]
# train_scripts/afhq_seg.sh's loss arguments, except: both reconstruction weights on and the full-resolution terms kept (only_raw_recons False), the
# cross-view weight at 1 (at the script's 1e-4 its gradient would vanish next to Gmain's) — every term of the loss carries weight in the record.
# (32^2 rays: at 64^2 the 2X head's first block adds its ToRGB output IN PLACE into a view of the rendered feature image — superresolution.py:279 on
#  triplane_cond.py:1056 — which autograd rejects on the CPU, where no fp16 copy of that image is made)
LOSS_KW = dict(r1_gamma=5, blur_init_sigma=0, blur_fade_kimg=200.0, gpc_reg_prob=0.5, gpc_reg_fade_kimg=0, dual_discrimination=True,
               neural_rendering_resolution_initial=32, neural_rendering_resolution_final=None, neural_rendering_resolution_fade_kimg=1000,
               filter_mode='antialiased', style_mixing_prob=0, lambda_l1=1.0, lambda_lpips=1.0, lambda_D_semantic=0.1,
               seg_weight=0, edge_weight=2, only_raw_recons=False, silhouette_loss=False, lambda_cross_view=1.0)
_MAIN = [p for p in LOSS_PHASES if p[0] in ('Gmain', 'Dmain', 'D_semanticmain')]
# (tag, loss arguments on top of LOSS_KW, phases, cur_nimg): the image-pose branch; the random-pose branch (loss.py:526-531); the discriminator-input blur
# of the first kimgs (sigma 1.5 at cur_nimg 850 of a 1-kimg fade: a 9-tap separable filter, loss.py:457-466, 545-551)
RUNS = [('img', dict(random_c_prob=0.0), LOSS_PHASES, 0),
        ('rnd', dict(random_c_prob=1.0), _MAIN, 0),
        ('blur', dict(random_c_prob=0.0, blur_init_sigma=10, blur_fade_kimg=1.0), [p for p in LOSS_PHASES if p[0] in ('Gmain', 'Dmain', 'D_semanticreg')], 850)]


def seed_networks(weights, G, D, D_semantic):
    weights.seed_module(G, seed=1)
    weights.seed_discriminator(D, seed=9)
    weights.seed_discriminator(D_semantic, seed=11)


def loss_phase_inputs(configs, n=2, device='cpu'):
    """Synthetic minibatch of the training loop (training_loop.py:483-507): images in [-1, 1], uint8 label maps, camera labels, z, gen_c."""
    g = torch.Generator().manual_seed(77)
    batch = {'image': torch.rand(n, 3, 128, 128, generator=g) * 2 - 1,
             'mask': torch.randint(0, 6, [n, 1, 128, 128], generator=g, dtype=torch.uint8),
             'pose': torch.tensor(np.stack([configs.orbit_camera(k, radius=1.7, focal=1.7074) for k in (4, 31)][:n]))}
    gen_z = torch.randn(n, 512, generator=g)
    gen_c = torch.tensor(np.stack([configs.orbit_camera(k, radius=1.7, focal=1.7074) for k in (58, 97)][:n]))
    return {k: v.to(device) for k, v in batch.items()}, gen_z.to(device), gen_c.to(device)


def run_loss_phases(loss, nets, batch, gen_z, gen_c, stats_sink, phases=LOSS_PHASES, cur_nimg=0):
    """One phase at a time: zero the phase network's gradients, requires_grad_(True), accumulate_gradients under DetRNG(100 + position of the phase in
    LOSS_PHASES), requires_grad_(False).  Returns {phase: (parameter names, gradient norms (-1: no gradient), {name: gradient}, statistics, draw log)}."""
    out = {}
    order = [p[0] for p in LOSS_PHASES]
    for phase, which, gain in phases:
        module = nets[which]
        for p in module.parameters():
            p.grad = None
        module.requires_grad_(True)
        stats_sink.clear()
        with det_rng.DetRNG(100 + order.index(phase)) as rng:
            loss.accumulate_gradients(phase=phase, batch=dict(batch), gen_z=gen_z, gen_c=gen_c, gain=gain, cur_nimg=cur_nimg)
        module.requires_grad_(False)
        names = [nm for nm, _ in module.named_parameters()]
        params = dict(module.named_parameters())
        norms = np.array([float(params[nm].grad.double().norm()) if params[nm].grad is not None else -1.0 for nm in names])
        out[phase] = (names, norms, {nm: params[nm].grad for nm in names if params[nm].grad is not None}, dict(stats_sink), list(rng.log))
    return out


def phase_record(key, names, norms, grads, stats, log):
    """What the golden file keeps of a phase: every parameter's gradient norm, the first 64 entries of four gradients, the mean of every reported
    statistic, the sequence of random draws (kind + shape)."""
    have = [nm for nm in names if nm in grads]
    heads = [have[0], have[len(have) // 3], have[(2 * len(have)) // 3], have[-1]]
    rec = {key + '.grad_names': np.array(names), key + '.grad_norms': norms, key + '.head_names': np.array(heads)}
    for j, nm in enumerate(heads):
        rec[f'{key}.h{j}'] = grads[nm].detach().float().cpu().reshape(-1)[:64].clone().numpy()
    rec[key + '.stat_names'] = np.array(sorted(stats))
    rec[key + '.stat_means'] = np.array([float(np.mean(stats[k])) for k in sorted(stats)])
    rec[key + '.draws'] = np.array([f'{k}{list(s)}' for k, s in log])
    return rec


def make_sink():
    """(dict, callable(name, value)): collects the mean of every reported statistic, as training_stats.report would be fed."""
    sink = {}

    def capture(name, value):
        v = torch.as_tensor(value).detach().double()
        sink.setdefault(name, []).append(float(v.mean()) if v.numel() else 0.0)
        return value
    return sink, capture
