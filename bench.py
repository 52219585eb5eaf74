#!/usr/bin/env python
"""Headline benchmark: rendered img/s of ``G.synthesis`` (seg2cat, 512^2 output, 128^2 rays x 128 depth samples).

    python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run, one rank per GPU)
    python bench.py --gpus N --train-step                   (BASELINE config 3 instead: one training iteration per step)

A "step" is one pass of the hot path — StyleGAN2 tri-plane backbone -> fused tri-plane ray-marcher -> two
super-resolution heads — over one batch of synthetic inputs already resident in HBM (random-init weights of the
real architecture, N(0,1) latents, orbit cameras).  Inference shards by image: every rank renders its own batch,
no data-path collective ("weak" scaling); the only collectives are the barriers bracketing the timed region and a
MAX-reduce of the elapsed time.  Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
``roofline`` for the dominant hand-written kernel (the fused ray-marcher; HIP-event timed inside the timed region, on the
stream it is launched on): ``bound`` / ``achieved`` / ``peak`` / ``frac`` name the resource that the committed counter passes of THIS kernel
(profiles/render_pmc.json, hash-checked against the kernel sources) show closest to its peak, busy units per launch / the LIVE launch
duration / the resource's peak at 2.4 GHz; ``tap_bytes`` keeps the survey's convention (1 543 algorithmic bytes per sample vs the 8 TB/s
HBM peak — the taps are served by L1 / L2, so that figure can exceed 1 and is not a bound); ``traffic`` = memory-side bytes per launch from
the PMC passes; ``exact_fp32`` = the same timed loop with every bf16x3 switch off (exact fp32 MFMA in the backbone and the decoder — the
reference's own arithmetic class, training_loop.py:278-280); ``cpu_baseline`` (the REFERENCE itself in a child
process when its checkout is reachable — ``kind: "reference"`` — else the CPU oracle, a port of the reference's force_fp32 CPU
path — ``kind: "port"``; a bounded sample: batch 1, same resolution and sample counts) and ``train_step`` (one training
iteration of BASELINE config 3 on the same GPUs: the SIX phases of training_loop.py:360-373 for train_scripts/afhq_seg.sh — Gmain (with the D_semantic
term and the cross-view block: four generator passes), Greg (density regularisation), Dmain, Dreg (R1), D_semanticmain, D_semanticreg — each driven
through Pix2Pix3DLoss.accumulate_gradients, followed by its flat gradient all-reduce over RCCL and its Adam step, then the G_ema update).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3          # v_mfma_f32_32x32x2_f32: the fp32 vector rate (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 1543.0         # algorithmic bytes per ray-sample of the ray-marcher (SURVEY §8d / DESIGN.md)
MLP_FLOP_PER_SAMPLE = 16640.0     # two-net OSG decoder, per evaluated point; the coarse pass evaluates layer 1 of the density
COARSE_FACTOR = 1.125             # net only on half the samples: + 1/8 of a full decode per final sample
FLOP_PER_IMG = 485e9              # modulated-conv FLOPs per 512^2 image


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=4, help='images per GPU per step (BASELINE configs[1]: 4)')
    p.add_argument('--depth', type=int, default=128, choices=[96, 128], help='depth samples per ray, coarse+fine (metric: 128)')
    p.add_argument('--dataset', default='seg2cat')
    p.add_argument('--force-fp32', action='store_true', help='run the super-resolution heads in fp32 too')
    p.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a captured hipGraph')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--miopen-find', action='store_true', help='let MIOpen benchmark its solvers for the vendor-library convs (slow warm-up)')
    p.add_argument('--cpu-reps', type=int, default=6, help='CPU baseline runs per thread count: 1 warm-up + (n - 1) timed (median reported; SURVEY 8(d): >= 5)')
    p.add_argument('--train-step', action='store_true', help='time training iterations of BASELINE config 3 (all six phases of training_loop.py, gradient all-reduce + Adam per phase) instead of inference')
    p.add_argument('--no-train-step', action='store_true', help='skip the short train_step extra of the default run')
    p.add_argument('--no-exact-fp32', action='store_true', help='skip the second timed loop with the bf16x3 switches off')
    p.add_argument('--no-configs', action='store_true', help="skip the short timed loops of BASELINE.json's other single-GPU configurations")
    p.add_argument('--train-nrr', type=int, default=128, help='neural rendering resolution of the training passes (train.py: 128 for the 512^2 configs)')
    return p.parse_args()


def build(args, device):
    from pix2pix3d_amd import configs, dnnlib
    import importlib.util
    spec = importlib.util.spec_from_file_location('p3d_weights', os.path.join(ROOT, 'tests', 'golden', 'weights.py'))
    _w = importlib.util.module_from_spec(spec)                       # name-seeded synthetic weights (torch + zlib only)
    spec.loader.exec_module(_w)
    kw = configs.generator_kwargs(args.dataset, depth=(args.depth // 2, args.depth // 2))
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
    _w.seed_module(G, seed=1)
    info = configs.dataset_info(args.dataset)
    rk = kw['rendering_kwargs']
    n = args.batch
    g = torch.Generator().manual_seed(1234 + int(os.environ.get('RANK', 0)))
    ws = torch.randn(n, G.backbone.num_ws, 512, generator=g)
    c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]))
    return G, kw, info, ws, c


def _reference_root():
    root = os.environ.get('P3D_REFERENCE', '/root/reference')
    return root if os.path.isfile(os.path.join(root, 'training', 'triplane_cond.py')) else None


def cpu_baseline(args, G_cpu, kw, info, ws, c):
    """CPU path of the same workload on the host cores, batch 1.  The reference itself (child process: its packages share names with
    this repository's mirrors) wherever its checkout exists; the oracle's port of it otherwise (the GPU box has no checkout)."""
    import subprocess
    nrr, rk = info['nrr'], kw['rendering_kwargs']
    what = f'batch 1, {nrr}^2 rays x {args.depth} samples -> {info["res"]}^2, fp32'
    ref = _reference_root()
    if ref is not None:
        cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'ref_baseline.py'), args.dataset, str(nrr), str(rk['depth_resolution']), str(rk['depth_resolution_importance']),
               str(args.cpu_reps)]
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, P3D_REFERENCE=ref), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        if r.returncode == 0:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            t = d['seconds_per_image']
            return {'value': round(1.0 / t, 4), 'unit': 'img/s', 'cores': d['threads'], 'kind': 'reference',
                    'sample': f'reference G.synthesis imported from {ref} in a child process (its force_fp32 CPU fallbacks), {what}, median of {d["reps"]} '
                              f'after 1 warm-up ({t:.2f} s/img; host has {d["logical_cores"]} logical cores)'}
    from oracle import model_oracle as M
    from pix2pix3d_amd import configs
    cfg = configs.oracle_cfg(args.dataset, depth=(args.depth // 2, args.depth // 2))
    sd = {k: v.float() for k, v in G_cpu.state_dict().items()}
    torch.manual_seed(1)
    u_c = torch.rand(1, nrr * nrr, rk['depth_resolution'], 1)
    u_f = torch.rand(nrr * nrr, rk['depth_resolution_importance'])

    def run():
        t0 = time.perf_counter()
        with torch.no_grad():
            M.synthesis(sd, cfg, ws[:1], c[:1], u_c, u_f, nrr=nrr, noise_mode='const')
        return time.perf_counter() - t0
    # the box has far more logical cores than this problem can use: time it at every core and at 32 threads, report the faster
    all_threads = torch.get_num_threads()
    best = None
    for threads in sorted({all_threads, min(all_threads, 32)}, reverse=True):
        torch.set_num_threads(threads)
        run()                                                        # warm-up at this thread count
        ts = [run() for _ in range(max(args.cpu_reps - 1, 1))]
        t = float(np.median(ts))
        if best is None or t < best[0]:
            best = (t, threads, len(ts))
    torch.set_num_threads(all_threads)
    t, threads, reps = best
    return {'value': round(1.0 / t, 4), 'unit': 'img/s', 'cores': threads, 'kind': 'port',
            'sample': f'oracle.model_oracle.synthesis (the reference checkout is not reachable here), {what}, median of {reps} after 1 warm-up at the '
                      f'faster of {all_threads} / {min(all_threads, 32)} threads ({t:.2f} s/img; host has {os.cpu_count()} logical cores)'}


def mfma_real_data_ceiling():
    """What an MFMA-only loop sustains on this part with the operands the SR heads multiply (profiles/round5_a_mfma_rate_probe.json, taken by
    tools/probe_mfma_rate.py on csrc/probes/mfma_rate_probe.hip): the pipe issues every 32 cycles, the chip holds ~1.69 GHz instead of 2.4 — the ceiling of
    ANY fp16 / bf16 kernel on non-zero data, printed beside the fractions of the quoted 2.5 PFLOP/s."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'round5_a_mfma_rate_probe.json')))
        rows = [r for r in d['rows'] if r['data'] == 'sr_layer' and r['chains'] == 8 and r['waves_per_simd'] == 2 and r['iters'] >= 500 and r['blocks'] % 256 == 0]
        tf = float(np.median([r['tflops'] for r in rows]))
        return {'tflops': round(tf, 1), 'frac_of_2p5pf': round(tf / 2500.0, 3), 'clock_ghz': round(float(np.median([r['clock_ghz'] for r in rows])), 3),
                'zero_operands_tflops': round(float(np.median([r['tflops'] for r in d['rows'] if r['data'] == 'zeros' and r['chains'] == 8])), 1),
                'source': 'profiles/round5_a_mfma_rate_probe.json: MFMA-only loop in conv3x3_h2_f16_kernel\'s register blocking, operands distributed like the SR heads\' (DESIGN.md section 2.4)'}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def _kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in ('pix2pix3d_amd/csrc/render.hip', 'pix2pix3d_amd/csrc/render_device.h'):
        h.update(open(os.path.join(ROOT, rel), 'rb').read())
    return h.hexdigest()[:16]


def pmc_record(args, nrr, exact=False):
    """The committed counter passes of the ray-marcher (profiles/render_pmc.json / render_pmc_exact_fp32.json, written by tests/gpu_pmc_render.py)
    — only for the workload they were taken on and only while the kernel sources are the ones they were taken from (else None: a stale
    PMC is not quoted)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'render_pmc_exact_fp32.json' if exact else 'render_pmc.json')))
        if args.batch == 4 and args.depth == 128 and nrr == 128 and args.dataset == 'seg2cat' and pmc.get('kernel_src_sha16') == _kernel_source_hash():
            return pmc
    except (OSError, KeyError, ValueError):
        pass
    return None


FUSED_ADAM = os.environ.get('P3D_BENCH_FUSED_ADAM', '1') != '0'
G_REG_INTERVAL, D_REG_INTERVAL = 4, 16        # train.py:239, 466 (--density_reg_every) and training_loop.py:249: the lazy-regularisation schedule
PHASE_ORDER = ('Gmain', 'Greg', 'Dmain', 'Dreg', 'D_semanticmain', 'D_semanticreg')


def train_setup(args, device, world):
    """BASELINE config 3 per GPU, as train_scripts/afhq_seg.sh + train.py assemble it: seg2cat generator in training mode (unfused modulation, fp16 SR
    heads), the dual discriminator D and — ``--dis_mask=True`` — the label-aware D_semantic on image + 6 label channels (fp16 top blocks, conv_clamp 256;
    train.py:289-318, 381-387, 509-512, 528-531; training_loop.py:299-308), batch 4, 128^2 rays x 48+48 samples; G_ema; one Adam per network with the
    lazy-regularisation correction of training_loop.py:360-373; the loss with the script's weights (configs.AFHQ_SEG_LOSS) minus LPIPS."""
    import copy
    from pix2pix3d_amd import configs, dnnlib
    from pix2pix3d_amd.training.loss import Pix2Pix3DLoss
    kw = configs.generator_kwargs(args.dataset, depth=(48, 48))
    rk = kw['rendering_kwargs']
    info = configs.dataset_info(args.dataset)
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).to(device).train().requires_grad_(False)
    d_kw = dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=info['res'], channel_base=32768, channel_max=512, num_fp16_res=4,
                conv_clamp=256, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=4))
    D = dnnlib.util.construct_class_by_name(img_channels=3, **d_kw).to(device).train().requires_grad_(False)
    D_sem = dnnlib.util.construct_class_by_name(img_channels=3 + info['sem'], **d_kw).to(device).train().requires_grad_(False)      # training_loop.py:308
    G_ema = copy.deepcopy(G).eval()
    nets = {'G': G, 'D': D, 'D_semantic': D_sem}
    phases = []                                                        # training_loop.py:360-373
    for name, lr, interval in (('G', 0.0025, G_REG_INTERVAL), ('D', 0.002, D_REG_INTERVAL), ('D_semantic', 0.002, D_REG_INTERVAL)):
        mb = interval / (interval + 1)
        # training_loop.py:362-368 builds torch.optim.Adam from opt_kwargs (betas, eps, lr); ``fused=True`` is that optimizer's one-pass device implementation (one
        # multi-tensor launch over p, g, m, v per step instead of the ~8 element-wise passes of the default "foreach" form — the 336 MB generator pays each pass
        # at HBM rate); same update rule, fp32 state.  P3D_BENCH_FUSED_ADAM=0: the default form.
        opt = torch.optim.Adam(nets[name].parameters(), lr=lr * mb, betas=[0 ** mb, 0.99 ** mb], eps=1e-8, fused=FUSED_ADAM and device.type == 'cuda')
        phases += [dict(name=name + 'main', module=nets[name], opt=opt, interval=1), dict(name=name + 'reg', module=nets[name], opt=opt, interval=interval)]
    n = args.batch
    g = torch.Generator().manual_seed(99 + int(os.environ.get('RANK', 0)))
    mask = torch.randint(0, info['sem'], [n, 1, info['res'], info['res']], generator=g, dtype=torch.uint8) if info['data_type'] == 'seg' else \
        torch.rand([n, 1, info['res'], info['res']], generator=g) * 2 - 1

    def cams(k0):
        return torch.tensor(np.stack([configs.orbit_camera(7 * k + k0, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in range(n)]), dtype=torch.float32).to(device)
    batch = {'image': (torch.rand(n, 3, info['res'], info['res'], generator=g) * 2 - 1).to(device), 'mask': mask.to(device), 'pose': cams(3)}
    for i, ph in enumerate(phases):                                    # training_loop.py:501-507: every phase gets its own z and gen_c
        ph['gen_z'], ph['gen_c'] = torch.randn(n, 512, generator=g).to(device), cams(11 + 5 * i)
    loss_kw = dict(configs.AFHQ_SEG_LOSS, lambda_lpips=0, neural_rendering_resolution_initial=args.train_nrr)
    loss = Pix2Pix3DLoss(device=device, G=G, D=D, D_semantic=D_sem, augment_pipe=None, lpips=None, **loss_kw)
    return dict(G=G, G_ema=G_ema, nets=nets, phases=phases, loss=loss, batch=batch, world=world, flat={})


def _finish_phase(st, ph):
    """training_loop.py:528-543: flat gradient all-reduce of the phase's module, then its optimizer step."""
    from pix2pix3d_amd import dp
    flat = dp.allreduce_gradients(ph['module'], world_size=st['world'], out=st['flat'])
    ph['opt'].step()
    return flat.numel() * 4 if flat is not None else 0


def train_iteration(st, timers, only=None, on_phase=None):
    """One iteration with EVERY phase of training_loop.py:487-543 (what the loop does on an iteration where both lazy regularisers fire), driven exactly
    as the loop drives them: zero_grad, requires_grad_(True), ``loss.accumulate_gradients(phase, batch, gen_z, gen_c, gain = interval, cur_nimg)``
    (pix2pix3d_amd/training/loss.py, the restatement of loss.py:509-1003 pinned by tests/test_loss_phases.py), requires_grad_(False), flat gradient
    all-reduce, Adam step — for Gmain (incl. the D_semantic term and the cross-view block: four generator passes), Greg, Dmain, Dreg, D_semanticmain,
    D_semanticreg; then the G_ema update (training_loop.py:545-557).  ``timers``: dict of lists of events, one mark after every phase."""
    sizes = {}

    def mark(key):
        e = torch.cuda.Event(enable_timing=True); e.record()
        timers.setdefault(key, []).append(e)
    mark('t0')
    for ph in st['phases']:
        if only is None or ph['name'] in only:
            ph['opt'].zero_grad(set_to_none=True)
            ph['module'].requires_grad_(True)
            st['loss'].accumulate_gradients(phase=ph['name'], batch=st['batch'], gen_z=ph['gen_z'], gen_c=ph['gen_c'], gain=ph['interval'], cur_nimg=200000)
            ph['module'].requires_grad_(False)
            sizes[ph['name']] = _finish_phase(st, ph)
        mark(ph['name'])
        if on_phase is not None:
            on_phase(ph['name'])
    with torch.no_grad():                                                  # G_ema (training_loop.py:545-557), beta for batch 4 x world / ema_kimg = batch * 10 / 32
        batch_size = st['batch']['image'].shape[0] * st['world']
        beta = 0.5 ** (batch_size / max(batch_size * 10 / 32 * 1000, 1e-8))
        torch._foreach_lerp_(list(st['G_ema'].parameters()), list(st['G'].parameters()), 1.0 - beta)
        for b_ema, b in zip(st['G_ema'].buffers(), st['G'].buffers()):
            b_ema.copy_(b)
    mark('ema')
    return sizes


def train_summary(timers, sizes, world, batch, wall_ms):
    def avg(a, b):
        return float(np.mean([x.elapsed_time(y) for x, y in zip(timers[a], timers[b])]))
    marks = ('t0',) + PHASE_ORDER + ('ema',)
    ph = {b: avg(a, b) for a, b in zip(marks[:-1], marks[1:])}
    lazy = (ph['Gmain'] + ph['Greg'] / G_REG_INTERVAL + ph['Dmain'] + ph['Dreg'] / D_REG_INTERVAL + ph['D_semanticmain'] + ph['D_semanticreg'] / D_REG_INTERVAL + ph['ema'])
    four = ph['Gmain'] + ph['Greg'] + ph['Dmain'] + ph['Dreg'] + ph['ema']
    return {'what': 'BASELINE config 3 per GPU (train_scripts/afhq_seg.sh: --dis_mask=True, random_c_prob 0.5, gamma 5, lambda_d_semantic 0.1, lambda_cross_view 1e-4, only_raw_recons), every phase of '
                    'training_loop.py in one iteration, each driven through Pix2Pix3DLoss.accumulate_gradients as the loop does: Gmain (mapping incl. the label-map Encoder + synthesis + D and D_semantic on '
                    'the generated pair, fwd + bwd; then the cross-view block: no-grad render from gen_c -> arg-max label map -> differentiated render -> no-grad reconstruction -> smooth-L1 backward: FOUR '
                    'generator passes) + Greg (density regularisation on the fused point kernels) + Dmain (one no-grad generator pass, generated + real) + Dreg (R1 double backward) + D_semanticmain + '
                    f'D_semanticreg (the same two on image + 6 label channels), each with its flat gradient all-reduce and Adam step, then G_ema; batch {batch}/GPU, 128^2 rays x 48+48; every convolution '
                    'forward / data gradient / weight gradient on libp3d_hip.so; fp32 products: ' + ('forward / data gradient as six bf16 MFMAs of three-piece register splits (bf16x6: the exact '
                    'kernels\' error against fp64, P3D_F32_BF16X6=0 = the f32-input MFMA), weight gradients on the f32-input MFMA' if _x6_on() else 'all on the f32-input MFMA') + '.  Not in it: LPIPS (lambda_lpips 0: a pretrained VGG that does not exist offline) and the augmentation pipe (--aug=noaug)',
            'ms_per_iteration': round(wall_ms, 2), 'img_per_s': round(batch * world / (wall_ms * 1e-3), 2),
            'phase_ms': {k: round(v, 2) for k, v in ph.items()},
            'lazy_schedule': {'G_reg_interval': G_REG_INTERVAL, 'D_reg_interval': D_REG_INTERVAL, 'ms_per_iteration': round(lazy, 2), 'img_per_s': round(batch * world / (lazy * 1e-3), 2),
                              'note': 'amortised as the loop runs it: Gmain + Greg / 4 + Dmain + Dreg / 16 + D_semanticmain + D_semanticreg / 16 + ema'},
            'four_phase_ms': {'value': round(four, 2), 'note': "round 3's workload for comparison: Gmain + Greg + Dmain + Dreg + ema of THIS run (Gmain here carries the D_semantic term and the cross-view block, "
                                                               'which round 3 did not have)'},
            'optimizer': 'torch.optim.Adam(fused=True): one multi-tensor launch per step' if FUSED_ADAM else 'torch.optim.Adam (default foreach form)',
            'allreduce': {'bytes_per_phase': sizes, 'note': ('world 1: concatenate + nan_to_num + scatter only' if world == 1 else 'RCCL ring over xGMI') + '; inside the phase times'}}


def train_roofline():
    """Which resource the largest kernels of the six-phase iteration keep busiest, from the committed per-kernel counter passes of that workload
    (profiles/kernel_pmc_train6.json, taken by tests/gpu_pmc_kernels.py train6 over `bench.py --train-step`): not live — a training iteration launches
    ~3 000 kernels of ~60 kinds, the counters need one profiled run per group."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'kernel_pmc_train6.json')))
    except (OSError, ValueError):
        return None
    rows = d.get('kernels') or []
    total = sum(r['total_ms'] for r in rows)
    out = []
    for r in rows[:4]:
        res = {k: r.get(k) for k in ('valu_issue', 'mfma_pipe', 'lds_array', 'hbm_frac') if r.get(k) is not None}
        if not res:
            continue
        bound = max(res, key=res.get)
        out.append({'kernel': r['kernel'], 'share_of_kernel_time': round(r['total_ms'] / total, 3), 'launches_per_profiled_run': r['launches'], 'avg_us': round(r['avg_us'], 1),
                    'bound': {'valu_issue': 'valu_issue', 'mfma_pipe': 'mfma', 'lds_array': 'lds', 'hbm_frac': 'hbm'}[bound], 'frac': round(res[bound], 3),
                    'all': {k: round(v, 3) for k, v in res.items()}, 'waves_per_simd': round(r.get('waves_per_simd') or 0.0, 2)})
    return {'source': 'profiles/kernel_pmc_train6.json (rocprofv3 --kernel-trace --pmc passes over bench.py --train-step; utilisation over each kernel\'s life at the clock the pass held)',
            'dominant': out[0] if out else None, 'next': out[1:]}


def _x6_on():
    from pix2pix3d_amd.torch_utils.ops import modconv
    return bool(modconv.f32_x6)


def train_arithmetic_floor(st, args, phase_ms):
    """One more (untimed) iteration with every native convolution logging its multiply-add FLOPs and arithmetic class, cut per phase: what the phase would
    take if its matrix work ran at the quoted peaks — exact-fp32 MFMA 157.3 TFLOP/s, fp16 / bf16 MFMA 2.5 PFLOP/s (each bf16x3 product = three bf16 MFMAs, each bf16x6
    product — the default of the fp32 forward / data-gradient convolutions — six) — and at
    what the matrix pipe sustains on real operands (mfma_real_data_ceiling).  The decoder MLPs of the fused renderer (exact fp32 MFMA in training: forward, and
    in the backward the recomputation + data gradient + weight gradient) are added from the sample counts; element-wise work, the gathers and the optimizer are
    not arithmetic the matrix pipes do and are left out: a FLOOR, to read the measured phase times against."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    cuts, log, fwd = {}, [], []
    _lib.kernel_events['conv_flops'] = log
    _lib.kernel_events['render_forward'] = fwd                        # one event pair per fused forward launch (inference and training forwards alike)
    marks = [(0, dict(rmod.backward_calls), 0)]

    def on_phase(name):
        marks.append((len(log), dict(rmod.backward_calls), len(fwd)))
        cuts[name] = (marks[-2], marks[-1])
    try:
        train_iteration(st, {}, on_phase=on_phase)
        torch.cuda.synchronize()
    finally:
        _lib.kernel_events.pop('conv_flops', None)
        _lib.kernel_events.pop('render_forward', None)
    ceiling = mfma_real_data_ceiling()
    ceil_tf = ceiling['tflops'] if ceiling else None
    samples = args.batch * args.train_nrr * args.train_nrr * 96          # config 3: 48 + 48 samples per ray
    out = {}
    for name, ((i0, b0, l0), (i1, b1, l1)) in cuts.items():
        fl = {'float32': 0.0, 'float16': 0.0, 'bf16x3': 0.0, 'bf16x6': 0.0}
        for d, f in log[i0:i1]:
            key = d if d in ('bf16x3', 'bf16x6') else ('float16' if 'float16' in d else 'float32')
            fl[key] += f
        n_bwd = b1.get('fused', 0) - b0.get('fused', 0)
        n_fwd = l1 - l0                                                  # (the density regularisation's point queries are 2 x 1000 points per image: not counted)
        mlp = samples * MLP_FLOP_PER_SAMPLE * (COARSE_FACTOR * n_fwd + 3.0 * n_bwd)
        f32 = fl['float32'] + mlp
        half = fl['float16'] + 3.0 * fl['bf16x3'] + 6.0 * fl['bf16x6']
        floor_ms = (f32 / (F32_MFMA_PEAK_TF * 1e12) + half / 2.5e15) * 1e3
        rec = {'tflop': {'f32_convs': round(fl['float32'] / 1e12, 3), 'f32_decoder_mlps': round(mlp / 1e12, 3), 'f16_convs': round(fl['float16'] / 1e12, 3),
                         'bf16x3_convs_fp32_equivalent': round(fl['bf16x3'] / 1e12, 3), 'bf16x6_convs_fp32_equivalent': round(fl['bf16x6'] / 1e12, 3)},
               'floor_ms_at_quoted_peaks': round(floor_ms, 2), 'measured_ms': phase_ms.get(name),
               'measured_over_floor': round(phase_ms[name] / floor_ms, 2) if phase_ms.get(name) and floor_ms > 0 else None}
        if ceil_tf:
            rec['floor_ms_half_precision_at_real_data_ceiling'] = round((f32 / (F32_MFMA_PEAK_TF * 1e12) + half / (ceil_tf * 1e12)) * 1e3, 2)
        out[name] = rec
    return {'what': 'matrix-pipe floor per phase: FLOPs of every native convolution (forward, data gradient, weight gradient) and of the renderer\'s decoder MLPs by arithmetic '
                    'class / the pipe\'s rate for that class; exact fp32 MFMA 157.3 TFLOP/s (weight gradients, decoder MLPs), fp16 / bf16 MFMA 2.5 PFLOP/s quoted, with a bf16x6 product counted as six and a bf16x3 product as three (' + (f'{ceil_tf:.0f} TFLOP/s on real operands' if ceil_tf else 'no probe record') + ')',
            'phases': out}


def run_train(args, device, world, dist, iters, warm):
    st = train_setup(args, device, world)
    timers = {}
    for _ in range(warm):
        sizes = train_iteration(st, {})
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        sizes = train_iteration(st, timers)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    summary = train_summary(timers, sizes, world, args.batch, elapsed / iters * 1e3)
    summary['roofline'] = train_roofline()
    try:
        summary['arithmetic_floor'] = train_arithmetic_floor(st, args, summary['phase_ms'])
    except Exception as e:                                           # noqa: BLE001 - an accounting extra must not cost the line
        summary['arithmetic_floor'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    if world > 1:                                                        # the exchange alone: one more all-reduce of each flat vector, timed
        from pix2pix3d_amd import dp
        bus = {}
        for name, mod in st['nets'].items():
            mod.requires_grad_(True)
            for p_ in mod.parameters():
                p_.grad = torch.zeros_like(p_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dp.allreduce_gradients(mod, world_size=world)
            e0.record(); flat = dp.allreduce_gradients(mod, world_size=world); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            bus[name] = {'bytes': flat.numel() * 4, 'ms': round(ms, 3), 'bus_GBps': round(2 * (world - 1) / world * flat.numel() * 4 / (ms * 1e-3) / 1e9, 1)}
        summary['allreduce']['timed_alone'] = bus
    del st
    torch.cuda.empty_cache()
    return summary, elapsed


XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0                # per GPU: 7 point-to-point links x ~153 GB/s (MI355X_MICROARCH.md / the task statement)


def collective_info(dist, world, rank, device, sizes=(336402180, 125438976), reps=5, strict=True):
    """N > 1 only: what the collective library saw, so that a multi-GPU line is self-evidencing — backend and RCCL version, every rank's device
    (index, name, PCI bus id: N distinct devices or the line is not a scaling measurement), the environment knobs that change the transport
    (the reference's scripts export NCCL_P2P_DISABLE=1, train_scripts/afhq_seg.sh:2: refused here, it would take the exchange off xGMI), and the flat
    fp32 all-reduce ALONE at the two message sizes of the training phases (G: 336 MB, D / D_semantic: 125 MB) with its bus bandwidth
    (2 (N-1)/N x bytes / time) next to the xGMI figures."""
    p2p_off = os.environ.get('NCCL_P2P_DISABLE', '0').strip()
    assert not strict or p2p_off in ('', '0'), 'NCCL_P2P_DISABLE is set: RCCL would route the gradient exchange through host memory instead of xGMI (unset it; the reference scripts set it for their own hardware)'
    from pix2pix3d_amd import dp
    backend = dist.get_backend()
    props = torch.cuda.get_device_properties(device)
    mine = {'rank': rank, 'device_index': device.index, 'name': props.name, 'pci_bus_id': getattr(props, 'pci_bus_id', None), 'pci_device_id': getattr(props, 'pci_device_id', None),
            'uuid': str(getattr(props, 'uuid', '')) or None, 'host': os.uname().nodename}
    devices = [None] * world
    dist.all_gather_object(devices, mine)
    version = None
    if backend == 'nccl':
        try:
            version = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                          # noqa: BLE001
            version = None
    timed = {}
    for nbytes in sizes:
        flat = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
        dp._all_reduce(flat, None)                                 # warm-up (communicator set-up, buffer registration)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dp._all_reduce(flat, None)
        e1.record(); torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=device)
        if backend == 'nccl':
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        timed[str(nbytes)] = {'ms': round(ms, 3), 'bus_GBps': round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1)}
        del flat
    return {'p2p_disabled': p2p_off not in ('', '0'),
            'backend': backend + (' (= RCCL on ROCm)' if backend == 'nccl' else ' (staged through host memory: the one-GPU test configuration, not a measurement)'),
            'version': version, 'ranks': world, 'devices': devices, 'distinct_devices': len({(d['host'], d['pci_bus_id'], d['device_index']) for d in devices}),
            'env': {k: os.environ.get(k) for k in ('NCCL_P2P_DISABLE', 'NCCL_DEBUG', 'NCCL_ALGO', 'NCCL_PROTO', 'NCCL_IB_DISABLE', 'NCCL_SOCKET_IFNAME', 'RCCL_MSCCL_ENABLE',
                                                   'HSA_ENABLE_IPC_MODE_LEGACY', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')},
            'allreduce_alone': timed,
            'xgmi': {'links_per_gpu': XGMI_LINKS, 'GBps_per_link': XGMI_LINK_GBS, 'ring_bound_GBps': XGMI_LINK_GBS, 'all_links_GBps': XGMI_LINKS * XGMI_LINK_GBS,
                     'note': 'a ring all-reduce is bound by ONE link per direction (bus bandwidth <= ~153 GB/s); direct all-to-all algorithms can use all seven (<= ~1071 GB/s)'}}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    local = local % torch.cuda.device_count()                      # (ranks > devices only in the one-GPU RCCL smoke run, tools/sessions/gpu_rccl_smoke.sh)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)          # RCCL over xGMI
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    from pix2pix3d_amd.training.volumetric_rendering import renderer as rmod
    _lib.lib()
    conv2d_gradfix.enabled = True                                   # training_loop.py:281
    rmod.fused_policy = 'require'
    torch.backends.cudnn.benchmark = bool(args.miopen_find)         # training_loop.py:280 sets True; find-mode costs ~100 s of warm-up per fresh box

    rccl = None
    if world > 1:
        # the training exchange must run over xGMI (strict: refuses NCCL_P2P_DISABLE); the inference line has no data-path collective, so there the switch is
        # recorded (rccl.p2p_disabled) instead of refused, and nothing in this object may cost the line itself
        if args.train_step:
            rccl = collective_info(dist, world, rank, device, strict=True)
        else:
            try:
                rccl = collective_info(dist, world, rank, device, strict=False)
            except Exception as e:                                   # noqa: BLE001
                rccl = {'error': f'{type(e).__name__}: {e}'[:300]}

    if args.train_step:                                              # BASELINE config 3 as the timed workload
        summary, elapsed = run_train(args, device, world, dist, args.steps, max(args.warmup, 1))
        if rank == 0:
            line = {'metric': 'training img/s (seg2cat 512^2, batch 4/GPU, 128^2 rays x 48+48 samples; Gmain + Greg + Dmain + Dreg + D_semanticmain + D_semanticreg, all-reduce + Adam per phase)',
                    'value': summary['img_per_s'], 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                    'ms_per_step': summary['ms_per_iteration'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                    'dtype': 'f32 (backbone, ray-marcher' + ('; convolution products formed as bf16x6, fp32-accurate' if _x6_on() else '') + ') + f16/f32-acc (super-resolution, discriminator top blocks), as train.py configures', 'data': 'synthetic',
                    'config': {'workload': summary['what'], 'launch': 'eager', 'parallelism': f'dp{world} (batch sharded, flat fp32 gradient all-reduce per phase)'},
                    'train_step': summary, 'rccl': rccl}
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    G_cpu, kw, info, ws_cpu, c_cpu = build(args, device)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, G_cpu, kw, info, ws_cpu, c_cpu)
    G = G_cpu.to(device)
    ws, c = ws_cpu.to(device), c_cpu.to(device)
    nrr = info['nrr']
    syn_kw = dict(noise_mode='const', neural_rendering_resolution=nrr, force_fp32=args.force_fp32)
    from pix2pix3d_amd.torch_utils.ops import modconv as _mc

    def make_step(G_, ws_, c_, kw_):
        def step_():
            with torch.no_grad():
                return G_.synthesis(ws_, c_, **kw_)
        return step_
    step = make_step(G, ws, c, syn_kw)

    def measure(settle_s, step=step, G=G, batch=args.batch, res=info['res']):
        """Warm up, capture the step as one hipGraph, settle, time K steps (barrier + synchronize on both sides, MAX over ranks); then an
        eager pass of the same steps with HIP events around the stages and the instrumented kernels."""
        for _ in range(max(args.warmup, 1)):
            out = step()
        torch.cuda.synchronize()
        # a fresh box may still be loading code objects / growing the allocator's pools: keep warming (untimed) until three
        # consecutive steps agree within 5 %, at most 40 extra steps
        hist = []
        for _ in range(40):
            t0 = time.perf_counter(); step(); torch.cuda.synchronize(); hist.append(time.perf_counter() - t0)
            if len(hist) >= 3 and max(hist[-3:]) < 1.05 * min(hist[-3:]):
                break
        assert out['image'].shape == (batch, 3, res, res)
        launch, graph = 'eager', None
        if not args.no_graph:
            try:                                                         # replay the whole step as one hipGraph
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = step()
                graph.replay()
                torch.cuda.synchronize()
                launch = 'hipgraph'
            except Exception as e:                                       # noqa: BLE001 - report and fall back to eager launches
                graph = None
                launch = f'eager (graph capture failed: {type(e).__name__}: {str(e)[:120]})'
                torch.cuda.synchronize()
        run = graph.replay if graph is not None else step
        # untimed settling: a box that has just booted (or idled) needs a moment of sustained load before its clocks / power state level
        # out.  Replay in chunks of 10 until at least 1.5 s have passed and three consecutive chunks agree within 1.5 % (at most settle_s).
        chunks, t_start = [], time.perf_counter()
        while True:
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            chunks.append(time.perf_counter() - t0)
            spent = time.perf_counter() - t_start
            if spent > settle_s or (spent > min(1.5, settle_s) and len(chunks) >= 3 and max(chunks[-3:]) < 1.015 * min(chunks[-3:])):
                break
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        del graph
        # per-kernel / per-stage HIP-event timing: a second, eager pass over the same steps (events cannot be read back from inside a
        # captured graph); the ray-marcher is one launch per step, so its event pair IS its launch duration.  Events are recorded on
        # torch's current stream == the stream every kernel of the step is launched on.
        stage_events = {'backbone': [], 'render': [], 'sr': []}

        def hook(mod, key):
            def pre(m, a):
                e = torch.cuda.Event(enable_timing=True); e.record(); m._p3d_e0 = e

            def post(m, a, o):
                e = torch.cuda.Event(enable_timing=True); e.record(); stage_events[key].append((m._p3d_e0, e))
            return mod.register_forward_pre_hook(pre), mod.register_forward_hook(post)
        handles = []
        handles += hook(G.backbone.synthesis, 'backbone') + hook(G.renderer, 'render')
        handles += hook(G.superresolution, 'sr') + hook(G.superresolution_semantic, 'sr')
        n_prof = min(args.steps, 10)
        conv_keys = ('conv_f16', 'conv_f32', 'conv_bf16x3') + (('conv_bf16x6',) if _mc.f32_x6 else ())
        mults = {'conv_bf16x3': 3.0, 'conv_bf16x6': 6.0}             # bf16 MFMAs executed per fp32 product
        for k in ('render_forward', 'conv_flops') + conv_keys:
            _lib.kernel_events[k] = []
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        kern = _lib.kernel_events.pop('render_forward')
        render_kernel_ms = sum(a.elapsed_time(b) for a, b in kern) / max(len(kern), 1)
        conv_ms = {k: sum(a.elapsed_time(b) for a, b in _lib.kernel_events.pop(k)) / n_prof for k in conv_keys}
        flops = _lib.kernel_events.pop('conv_flops')
        conv_fl = {'conv_f16': sum(f for d, f in flops if 'float16' in d) / n_prof, 'conv_f32': sum(f for d, f in flops if 'float32' in d) / n_prof,
                   'conv_bf16x3': sum(f for d, f in flops if d == 'bf16x3') / n_prof, 'conv_bf16x6': sum(f for d, f in flops if d == 'bf16x6') / n_prof}
        for h in handles:
            h.remove()
        stage_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / n_prof if v else 0.0) for k, v in stage_events.items()}
        # conv_f32: exact fp32 MFMA kernels vs the 157.3 TF fp32 matrix peak.  conv_bf16x3: the fp32 layers computed as three bf16 MFMAs per
        # product: 'tflops' counts each fp32 multiply-add once (fp32-equivalent), 'frac_of_peak' the 3x bf16 MFMA work it executes vs 2.5 PF
        # (conv_bf16x6, only with modconv.f32_x6: six per product)
        mfma_conv = {k: {'ms_per_step': round(conv_ms[k], 3), 'tflops': round(conv_fl[k] / (conv_ms[k] * 1e-3) / 1e12, 1) if conv_ms[k] > 0 else None,
                         'frac_of_peak': round(conv_fl[k] * mults.get(k, 1.0) / (conv_ms[k] * 1e-3) / 1e12 / (157.3 if k == 'conv_f32' else 2500.0), 3) if conv_ms[k] > 0 else None}
                     for k in conv_keys}
        return dict(elapsed=elapsed, launch=launch, render_kernel_ms=render_kernel_ms, n_render=len(kern), stage_ms=stage_ms, mfma_conv=mfma_conv)

    def roofline(m, mlp_bf3):
        """The ray-marcher's roofline object from its live launch duration + the committed counter passes of this kernel build."""
        samples_per_launch = args.batch * nrr * nrr * args.depth
        render_s = m['render_kernel_ms'] * 1e-3
        tap_gbs = samples_per_launch * BYTES_PER_SAMPLE / render_s / 1e9 if render_s > 0 else 0.0
        pmc = pmc_record(args, nrr, exact=not mlp_bf3)
        mfma_floor_ms = samples_per_launch * MLP_FLOP_PER_SAMPLE * COARSE_FACTOR * (3.0 / (2500.0e12) if mlp_bf3 else 1.0 / (F32_MFMA_PEAK_TF * 1e12)) * 1e3
        r = {'kernel': 'render_forward_kernel (fused tri-plane ray-marcher), decoder MLPs as ' + ('bf16x3 (3 bf16 MFMAs per fp32 product)' if mlp_bf3 else 'exact f32-input MFMA')}
        bind = pmc.get('binding') if pmc else None
        if bind and bind.get('busy_units_per_launch') and render_s > 0:
            ach = bind['busy_units_per_launch'] / render_s
            r.update({'bound': bind['resource'], 'achieved': round(ach / 1e9, 2), 'peak': round(bind['peak_units_per_s'] / 1e9, 2), 'unit': 'G ' + bind['unit'],
                      'frac': round(ach / bind['peak_units_per_s'], 4),
                      'binding': {'source': 'profiles/' + ('render_pmc.json' if mlp_bf3 else 'render_pmc_exact_fp32.json') + f" (kernel sources sha16 {pmc['kernel_src_sha16']} == this tree)",
                                  'counter': bind['counter'], 'busy_units_per_launch': bind['busy_units_per_launch'],
                                  'utilisation_in_pmc_pass': round(bind['utilisation_in_pmc_pass'], 4),
                                  'all_resources_in_pmc_pass': {k: round(v, 4) for k, v in pmc['derived'].get('utilisation', {}).items()},
                                  'wave_time_split': {k: round(v, 3) for k, v in pmc['derived'].get('wave_time_split', {}).items()},
                                  'note': 'the resource of this kernel closest to its peak in the counter passes; frac = its busy units per launch / the LIVE launch duration / its peak at the 2.4 GHz '
                                          'maximum clock (a lower bound on the utilisation at the clock the run actually held)'}})
        else:
            r.update({'bound': 'hbm', 'achieved': round(tap_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(tap_gbs / HBM_PEAK_GBS, 4), 'binding': None,
                      'bound_note': 'no counter passes of THIS kernel build under profiles/ (hash mismatch): only the tap-bytes convention is available, which the taps being served by L1 / L2 lets exceed 1'})
        r.update({'traffic': pmc.get('traffic_bytes_per_launch') if pmc else None,
                  'tap_bytes': {'achieved': round(tap_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(tap_gbs / HBM_PEAK_GBS, 4), 'bytes_per_unit': BYTES_PER_SAMPLE,
                                'note': "SURVEY 8(d)'s convention: 12 taps x 128 B + ray I/O per ray-sample vs the HBM peak; the taps are served by L1 / L2 (traffic << these bytes), so this is a "
                                        'throughput figure, not a bound'},
                  'mfma_frac': round(mfma_floor_ms / m['render_kernel_ms'], 4) if m['render_kernel_ms'] > 0 else None, 'mfma_floor_ms': round(mfma_floor_ms, 4),
                  'ms_per_launch': round(m['render_kernel_ms'], 4), 'launches_timed': m['n_render'], 'units_per_launch': samples_per_launch})
        return r

    main_m = measure(8.0)
    exact = None
    if not args.no_exact_fp32:                                       # the same loop with every bf16x3 switch off: exact fp32 MFMA in the backbone and the decoder
        prev = (_mc.split_bf16, rmod.mlp_bf16x3, _mc.f32_x6)
        _mc.split_bf16, rmod.mlp_bf16x3, _mc.f32_x6 = False, False, False
        try:
            torch.cuda.empty_cache()
            exact_m = measure(3.0)
            exact = {'value': round(args.batch * world * args.steps / exact_m['elapsed'], 3), 'unit': 'img/s', 'ms_per_step': round(exact_m['elapsed'] / args.steps * 1e3, 3),
                     'launch': exact_m['launch'], 'stage_ms': {k: round(v, 3) for k, v in exact_m['stage_ms'].items()}, 'mfma_conv': exact_m['mfma_conv'],
                     'roofline': roofline(exact_m, False),
                     'what': 'P3D_BF16X3=0 P3D_MLP_BF16X3=0 P3D_F32_BF16X6=0: every fp32 convolution and the decoder MLPs on the f32-input MFMA (exact fp32 products, the arithmetic class the reference '
                             'insists on, training_loop.py:278-280); super-resolution heads unchanged'}
            # the same fp32-accurate leg with the backbone's products formed on the bf16 matrix pipe (P3D_F32_BF16X6: three-piece splits, six MFMAs
            # per product; plain fp32 tensors and weights — DESIGN 2.4c) and, since round 6, layer 1 of the decoder MLPs the same way (P3D_MLP_L1X6,
            # render_forward_kernel<.., L1X6>: DESIGN 2.1); the decoder's layer 2 stays on the f32-input MFMA
            _mc.f32_x6 = True
            try:
                torch.cuda.empty_cache()
                x6_m = measure(3.0)
                exact['backbone_as_bf16x6'] = {'value': round(args.batch * world * args.steps / x6_m['elapsed'], 3), 'unit': 'img/s',
                                               'ms_per_step': round(x6_m['elapsed'] / args.steps * 1e3, 3), 'launch': x6_m['launch'],
                                               'stage_ms': {k: round(v, 3) for k, v in x6_m['stage_ms'].items()}, 'mfma_conv': x6_m['mfma_conv'],
                                               'what': 'P3D_BF16X3=0 P3D_MLP_BF16X3=0 alone (P3D_F32_BF16X6 and P3D_MLP_L1X6 at their default, 1): the fp32 convolutions and layer 1 of the decoder MLPs as six bf16 MFMAs per '
                                                       'product (hi/mid/lo pieces, error class of the exact kernels: tests/test_conv_gpu.py::test_bf16x6_formulation_of_the_fp32_convolution, '
                                                       'tests/test_render_gpu.py::test_layer1_as_bf16x6_is_fp32_accurate); decoder layer 2 on the f32-input MFMA'}
            except Exception as e:                                   # noqa: BLE001
                exact['backbone_as_bf16x6'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            finally:
                _mc.f32_x6 = False
        except Exception as e:                                       # noqa: BLE001 - the headline line must still be printed
            exact = {'error': f'{type(e).__name__}: {e}'[:300]}
        finally:
            _mc.split_bf16, rmod.mlp_bf16x3, _mc.f32_x6 = prev

    # BASELINE.json's other single-GPU shapes, each timed the same way (own warm-up, own hipGraph, short settling): configs[1] at its own 48+48
    # samples, configs[3]'s per-GPU share (edge2car, batch 16 over 2 GPUs: train.py:451-461) and configs[4]'s (seg2face, batch 8 over 4 GPUs:
    # train.py:425-437).  configs[0] is the reference's CPU case (cpu_baseline), configs[2] the training iteration (train_step).
    other = None
    if not args.no_configs and world == 1 and args.dataset == 'seg2cat' and args.batch == 4 and args.depth == 128 and not args.force_fp32:      # (N > 1: the scaling run times configs[1]'s headline shape only)
        import copy as _copy
        other = {}
        G = G.cpu()
        for key, ds_name, nb, depth in (('configs[1] seg2cat 512^2, 48+48 samples, batch 4', 'seg2cat', 4, 96),
                                        ('configs[3] edge2car 128^2, 64^2 rays x 64+64 samples, batch 8 per GPU (16 over 2 GPUs)', 'edge2car', 8, 128),
                                        ('configs[4] seg2face 512^2 (19 label channels), 48+48 samples, batch 2 per GPU (8 over 4 GPUs)', 'seg2face', 2, 96)):
            try:
                torch.cuda.empty_cache()
                a2 = _copy.copy(args)
                a2.dataset, a2.batch, a2.depth = ds_name, nb, depth
                G2_cpu, kw2, info2, ws2, c2 = build(a2, device)
                G2 = G2_cpu.to(device)
                nrr2 = info2['nrr']
                m2 = measure(2.0, step=make_step(G2, ws2.to(device), c2.to(device), dict(noise_mode='const', neural_rendering_resolution=nrr2, force_fp32=False)),
                             G=G2, batch=nb, res=info2['res'])
                spl = nb * nrr2 * nrr2 * depth
                other[key] = {'value': round(nb * world * args.steps / m2['elapsed'], 2), 'unit': 'img/s', 'ms_per_step': round(m2['elapsed'] / args.steps * 1e3, 3), 'launch': m2['launch'],
                              'stage_ms': {k: round(v, 3) for k, v in m2['stage_ms'].items()},
                              'ray_marcher': {'ms_per_launch': round(m2['render_kernel_ms'], 4), 'ray_samples_per_launch': spl,
                                              'ray_samples_per_s': round(spl / (m2['render_kernel_ms'] * 1e-3), 1) if m2['render_kernel_ms'] > 0 else None},
                              'mfma_conv': m2['mfma_conv']}
                del G2, G2_cpu, m2
            except Exception as e:                                   # noqa: BLE001 - the headline line must still be printed
                other[key] = {'error': f'{type(e).__name__}: {e}'[:300]}
        G = G.to(device)

    train = None
    if not args.no_train_step:                                       # short: 2 warm-up + 3 timed iterations
        G = G.cpu()
        torch.cuda.empty_cache()
        try:
            train, _ = run_train(args, device, world, dist, 3, 2)
        except Exception as e:                                       # noqa: BLE001 - the headline line must still be printed
            train = {'error': f'{type(e).__name__}: {e}'[:300]}
        if 'error' not in train:                                     # the same iteration with the generator's fp32 training convolutions as bf16x3 (opt-in)
            from pix2pix3d_amd.training import triplane as _tp
            prev_tp, _tp.train_products_bf16x3 = _tp.train_products_bf16x3, True
            try:
                t2, _ = run_train(args, device, world, dist, 3, 1)
                train['generator_bf16x3'] = {'what': 'P3D_TRAIN_G_BF16X3=1: forward and data-gradient convolutions of the GENERATOR (backbone, label-map Encoder, fp32 part of the SR '
                                                     'heads) as three bf16 MFMAs per product — the arithmetic its inference passes use; discriminators and every weight gradient exact fp32',
                                             'ms_per_iteration': t2['ms_per_iteration'], 'img_per_s': t2['img_per_s'], 'phase_ms': t2['phase_ms'],
                                             'lazy_schedule_ms': t2['lazy_schedule']['ms_per_iteration']}
            except Exception as e:                                   # noqa: BLE001
                train['generator_bf16x3'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            finally:
                _tp.train_products_bf16x3 = prev_tp
            prev6, _mc.f32_x6 = _mc.f32_x6, False                    # the iteration with every fp32 product on the f32-input MFMA (the default forms them as bf16x6: same error class)
            try:
                t3, _ = run_train(args, device, world, dist, 3, 1)
                train['f32_input_mfma'] = {'what': 'P3D_F32_BF16X6=0: every fp32 forward and data-gradient convolution (generator, discriminators, frozen passes) on v_mfma_f32_32x32x2_f32 instead of '
                                                   'six bf16 MFMAs per product of three-piece splits (the default; weight gradients are on the f32-input MFMA either way)',
                                           'ms_per_iteration': t3['ms_per_iteration'], 'img_per_s': t3['img_per_s'], 'phase_ms': t3['phase_ms'],
                                           'lazy_schedule_ms': t3['lazy_schedule']['ms_per_iteration']}
            except Exception as e:                                   # noqa: BLE001
                train['f32_input_mfma'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            finally:
                _mc.f32_x6 = prev6

    if rank == 0:
        ceiling = mfma_real_data_ceiling()
        if ceiling:                                                  # the same classes against what the pipe can do on real data at the clock it then holds
            ceiling['frac_of_it'] = {k: (round(v['tflops'] * {'conv_bf16x3': 3.0, 'conv_bf16x6': 6.0}.get(k, 1.0) / ceiling['tflops'], 3) if v.get('tflops') else None)
                                     for k, v in main_m['mfma_conv'].items() if k != 'conv_f32'}
        bb = ('f32 tensors + f32 accumulation; the backbone convolutions (3x3 and the 1x1 ToRGB) form each product as 3 bf16 MFMAs of (hi, lo) splits ("bf16x3", <= 5e-6 of the '
              'output range vs fp64 per layer; P3D_BF16X3=0 = exact f32 MFMA, timed as `exact_fp32`)') if _mc.split_bf16 else 'f32 (exact f32 MFMA)'
        rm = 'f32 gather / sampling / compositing, decoder MLPs as bf16x3 (P3D_MLP_BF16X3=0 = exact f32 MFMA)' if rmod.mlp_bf16x3 else 'f32'
        dtype_desc = f'backbone: {bb}; ray-marcher: {rm}; super-resolution: ' + ('f32' if args.force_fp32 else 'f16 storage / f32 accumulation (the reference GPU config)')
        elapsed, stage_ms = main_m['elapsed'], main_m['stage_ms']
        ms_per_step = elapsed / args.steps * 1e3
        imgs = args.batch * world * args.steps
        samples_per_launch = args.batch * nrr * nrr * args.depth
        render_s = main_m['render_kernel_ms'] * 1e-3
        line = {
            'metric': f'rendered img/s ({info["res"]}^2, {args.depth} depth)',
            'value': round(imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype_desc,
            'data': 'synthetic',
            'config': {'workload': f'{args.dataset} G.synthesis: batch {args.batch}/GPU, 256^2x96 tri-planes, {nrr}^2 rays x {args.depth // 2}+{args.depth // 2} samples, '
                                   f'two {info["sr"]} SR heads -> {info["res"]}^2 image + label map', 'launch': main_m['launch'], 'parallelism': f'replicas x{world} (images sharded, no collective)'},
            'ray_samples_per_s': round(samples_per_launch / render_s, 1) if render_s > 0 else None,
            'stage_ms': {k: round(v, 3) for k, v in stage_ms.items()},
            'conv_tflops': round(FLOP_PER_IMG * args.batch / ((stage_ms['backbone'] + stage_ms['sr']) * 1e-3) / 1e12, 2) if stage_ms['backbone'] + stage_ms['sr'] > 0 else None,
            'roofline': roofline(main_m, bool(rmod.mlp_bf16x3)),
            'mfma_conv': main_m['mfma_conv'],
            'mfma_real_data_ceiling': ceiling,
            'exact_fp32': exact,
            'configs': other,
            'cpu_baseline': cpu,
            'train_step': train,
            'rccl': rccl,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
