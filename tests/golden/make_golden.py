"""Generate golden input/output vectors by running the REFERENCE implementation on CPU.

Run in the authoring container only (it needs the read-only checkout at /root/reference, which does
not exist on the GPU box):

    python tests/golden/make_golden.py [group ...]      # groups: ops renderer semrenderer model model_full flrelu train train_full greg checkpoint variants discriminator api srheads architectures helpers loss_phases discriminator_full encoder_variants

The reference and this repo own the same top-level module names, so this script must never import
``pix2pix3d_amd``; it puts /root/reference first on sys.path and imports the reference's modules
directly.  Outputs are small .npz files committed next to this script; tests load them with numpy
only.  Everything is seeded, so re-running reproduces the files bit for bit (same torch build).
"""
import os
import sys

import numpy as np

REF = os.environ.get('P3D_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import torch  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)')


# ---------------------------------------------------------------------------------------------------------
def group_ops():
    from torch_utils.ops import bias_act, upfirdn2d, conv2d_resample
    from training.networks_stylegan2 import modulated_conv2d

    # --- bias_act: every activation, forward + first- and second-order gradients via autograd of _bias_act_ref
    g = torch.Generator().manual_seed(1234)
    out = {}
    acts = list(bias_act.activation_funcs.keys())
    for i, act in enumerate(acts):
        x = (torch.randn([2, 5, 6, 3], generator=g, dtype=torch.float64) * 2).requires_grad_(True)
        b = torch.randn([5], generator=g, dtype=torch.float64).requires_grad_(True)
        dy = torch.randn([2, 5, 6, 3], generator=g, dtype=torch.float64)
        ddx = torch.randn([2, 5, 6, 3], generator=g, dtype=torch.float64)
        alpha = 0.3 if act == 'lrelu' else None
        gain = [None, 1.7][i % 2]
        clamp = [None, 1.1][(i // 2) % 2]
        y = bias_act._bias_act_ref(x, b, dim=1, act=act, alpha=alpha, gain=gain, clamp=clamp)
        dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
        d2 = torch.autograd.grad(dx, x, ddx, allow_unused=True)[0] if dx.requires_grad else None
        out[f'{act}.x'], out[f'{act}.b'], out[f'{act}.dy'], out[f'{act}.ddx'] = x, b, dy, ddx
        out[f'{act}.alpha'] = np.float64(-1 if alpha is None else alpha)
        out[f'{act}.gain'] = np.float64(-1 if gain is None else gain)
        out[f'{act}.clamp'] = np.float64(-1 if clamp is None else clamp)
        out[f'{act}.y'], out[f'{act}.dx'], out[f'{act}.db'] = y, dx, db
        out[f'{act}.d2'] = torch.zeros_like(x) if d2 is None else d2
    # bias along the last dim of a 2-D tensor (FullyConnectedLayer use) and no bias
    x = torch.randn([7, 9], generator=g, dtype=torch.float64)
    b = torch.randn([9], generator=g, dtype=torch.float64)
    out['fc.x'], out['fc.b'] = x, b
    out['fc.y'] = bias_act._bias_act_ref(x, b, dim=1, act='lrelu')
    out['nob.y'] = bias_act._bias_act_ref(x, None, act='swish', gain=0.5, clamp=0.4)
    save('ops_bias_act', **out)

    # --- upfirdn2d
    cases = [  # (filter taps, up, down, padding, flip, gain)
        ([1, 3, 3, 1], 1, 1, [1, 1, 1, 1], False, 4.0),
        ([1, 3, 3, 1], 2, 1, [2, 1, 2, 1], False, 4.0),
        ([1, 3, 3, 1], 1, 2, [1, 1, 1, 1], False, 1.0),
        ([1, 3, 3, 1], 2, 2, [3, 0, 1, 2], True, 1.5),
        ([1, 2, 1], [2, 1], [1, 3], [2, 2, 0, 4], False, 1.0),
        ([1, 4, 6, 4, 1, 2, 3, 5], 2, 1, [4, 3, 4, 3], False, 2.0),           # 8 taps -> separable
        ([1, 4, 6, 4, 1, 2, 3, 5], 1, 2, [3, 3, 3, 3], True, 1.0),
        ([1, 3, 3, 1], 1, 1, [-1, 2, 3, -2], False, 1.0),                     # negative padding crops
        (None, 3, 2, [0, 1, 2, 0], False, 1.0),
    ]
    out = {'num_cases': np.int64(len(cases))}
    for i, (taps, up, down, pad, flip, gain) in enumerate(cases):
        f = upfirdn2d.setup_filter(taps) if taps is not None else None
        x = torch.randn([2, 3, 9, 8], generator=g, dtype=torch.float64)
        y = upfirdn2d._upfirdn2d_ref(x.float(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        out[f'{i}.x'], out[f'{i}.y'] = x.float(), y
        out[f'{i}.f'] = f if f is not None else np.zeros([0], np.float32)
        out[f'{i}.up'], out[f'{i}.down'] = np.array(upfirdn2d._parse_scaling(up)), np.array(upfirdn2d._parse_scaling(down))
        out[f'{i}.pad'], out[f'{i}.flip'], out[f'{i}.gain'] = np.array(pad), np.int64(flip), np.float64(gain)
    # helpers
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    x = torch.randn([1, 2, 6, 6], generator=g)
    out['h.x'], out['h.f'] = x, f
    out['h.up'] = upfirdn2d.upsample2d(x, f, impl='ref')
    out['h.down'] = upfirdn2d.downsample2d(x, f, impl='ref')
    out['h.filt'] = upfirdn2d.filter2d(x, f, impl='ref')
    out['h.f_sep'] = upfirdn2d.setup_filter([1, 4, 6, 4, 1, 2, 3, 5], gain=2.0, flip_filter=True)
    save('ops_upfirdn2d', **out)

    # --- conv2d_resample + modulated_conv2d (CPU => F.conv2d / F.conv_transpose2d underneath)
    out = {}
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    out['f'] = f
    rcases = [(3, 1, 1, 1, True), (3, 2, 1, 1, False), (3, 1, 2, 1, True), (1, 2, 1, 0, True), (1, 1, 2, 0, True), (3, 1, 1, [2, 0, 1, 1], True)]
    out['num_resample'] = np.int64(len(rcases))
    for i, (k, up, down, pad, flipw) in enumerate(rcases):
        x = torch.randn([2, 4, 8, 8], generator=g)
        w = torch.randn([5, 4, k, k], generator=g)
        y = conv2d_resample.conv2d_resample(x, w, f=f, up=up, down=down, padding=pad, flip_weight=flipw)
        out[f'r{i}.x'], out[f'r{i}.w'], out[f'r{i}.y'] = x, w, y
        out[f'r{i}.cfg'] = np.array([k, up, down, int(flipw)])
        out[f'r{i}.pad'] = np.array(pad if isinstance(pad, list) else [pad] * 4)
    mcases = [(3, 1, True, True), (3, 2, True, True), (1, 1, False, True), (3, 1, True, False), (3, 2, True, False)]
    out['num_mod'] = np.int64(len(mcases))
    for i, (k, up, demod, fused) in enumerate(mcases):
        x = torch.randn([2, 6, 8, 8], generator=g)
        w = torch.randn([5, 6, k, k], generator=g)
        s = torch.randn([2, 6], generator=g) + 1
        res = 8 * up
        noise = torch.randn([res, res], generator=g) * 0.1 if k == 3 else None
        y = modulated_conv2d(x, w, s, noise=noise, up=up, padding=k // 2, resample_filter=f, demodulate=demod,
                             flip_weight=(up == 1), fused_modconv=fused)
        out[f'm{i}.x'], out[f'm{i}.w'], out[f'm{i}.s'], out[f'm{i}.y'] = x, w, s, y
        out[f'm{i}.noise'] = noise if noise is not None else np.zeros([0], np.float32)
        out[f'm{i}.cfg'] = np.array([k, up, int(demod), int(fused)])
    save('ops_conv', **out)


GROUPS = {'ops': group_ops}


# ---------------------------------------------------------------------------------------------------------
class _RandTape:
    """Records every torch.rand_like / torch.rand draw the reference makes (renderer.py:190, :237)."""

    def __enter__(self):
        self.draws, self._rl, self._r = [], torch.rand_like, torch.rand

        def rand_like(t, *a, **k):
            v = self._rl(t, *a, **k); self.draws.append(v.clone()); return v

        def rand(*a, **k):
            v = self._r(*a, **k); self.draws.append(v.clone()); return v
        torch.rand_like, torch.rand = rand_like, rand
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand = self._rl, self._r


def _look_at(radius, yaw, pitch):
    """cam2world of a camera on a sphere looking at the origin (OpenCV convention: +z forward, +y down)."""
    pos = np.array([radius * np.sin(pitch) * np.sin(yaw), radius * np.cos(pitch), radius * np.sin(pitch) * np.cos(yaw)])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, np.array([0., 1., 0.])); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, fwd, pos
    return m.astype(np.float32)


def _decoder_arrays(dec):
    out = {'w1': dec.net[0].weight, 'b1': dec.net[0].bias, 'w2': dec.net[2].weight, 'b2': dec.net[2].bias}
    if hasattr(dec, 'net_semantic'):
        out.update({'w1s': dec.net_semantic[0].weight, 'b1s': dec.net_semantic[0].bias, 'w2s': dec.net_semantic[2].weight, 'b2s': dec.net_semantic[2].bias})
    return out


def group_renderer():
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.triplane import OSGDecoder
    from training.triplane_cond import OSGDecoder_semantic_lateSeparate

    cases = [
        dict(name='seg', nets=2, sem_sigmoid=False, n=2, hw=(20, 16), res=6, focal=4.2647, radius=2.7, lr_mul=1.0,
             opts=dict(depth_resolution=12, depth_resolution_importance=10, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode='softplus')),
        dict(name='car', nets=2, sem_sigmoid=True, n=1, hw=(16, 16), res=7, focal=1.2, radius=1.7, lr_mul=1.0,
             opts=dict(depth_resolution=16, depth_resolution_importance=16, ray_start=0.1, ray_end=2.6, box_warp=1.6, disparity_space_sampling=False, clamp_mode='softplus', white_back=True)),
        dict(name='osg', nets=1, sem_sigmoid=False, n=2, hw=(12, 12), res=5, focal=2.0, radius=2.7, lr_mul=0.5,
             opts=dict(depth_resolution=9, depth_resolution_importance=5, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=True, clamp_mode='softplus')),
        dict(name='auto', nets=2, sem_sigmoid=False, n=1, hw=(16, 16), res=6, focal=1.0, radius=2.0, lr_mul=1.0,
             opts=dict(depth_resolution=10, depth_resolution_importance=8, ray_start='auto', ray_end='auto', box_warp=1.2, disparity_space_sampling=False, clamp_mode='softplus')),
    ]
    for ci, cs in enumerate(cases):
        torch.manual_seed(100 + ci)
        n, (h, w), res = cs['n'], cs['hw'], cs['res']
        planes = torch.randn(n, 3, 32, h, w)
        dopt = {'decoder_lr_mul': cs['lr_mul'], 'decoder_output_dim': 32}
        if cs['nets'] == 2:
            dec = OSGDecoder_semantic_lateSeparate(32, dict(dopt, sigmoid=cs['sem_sigmoid'], semantic_channels=6))
        else:
            dec = OSGDecoder(32, dopt)
        with torch.no_grad():
            for p_ in dec.parameters():
                if p_.ndim == 1:
                    p_.copy_(torch.randn_like(p_) * 0.3 / cs['lr_mul'])
        c2w = torch.tensor(np.stack([_look_at(cs['radius'], 0.3 + 0.9 * i, 1.4 - 0.2 * i) for i in range(n)]))
        K = torch.tensor([[cs['focal'], 0.01 * ci, 0.5], [0, cs['focal'] * 1.05, 0.48], [0, 0, 1]], dtype=torch.float32).repeat(n, 1, 1)
        ray_o, ray_d = RaySampler()(c2w, K, res)
        rend = ImportanceRenderer()
        rec = {}
        orig_imp, orig_uni = rend.sample_importance, rend.unify_samples

        def imp(z, wts, k):
            rec['z_coarse'], rec['w_coarse'] = z.clone(), wts.clone()
            out = orig_imp(z, wts, k); rec['z_fine'] = out.clone(); return out

        def uni(*a):
            out = orig_uni(*a); rec['z_all'] = out[0].clone(); return out
        rend.sample_importance, rend.unify_samples = imp, uni
        with _RandTape() as tape, torch.no_grad():
            feat, depth, wsum = rend(planes, dec, ray_o, ray_d, cs['opts'])
        assert len(tape.draws) == 2
        pts = (torch.rand(n, 40, 3) - 0.5) * 1.3 * cs['opts']['box_warp']
        with torch.no_grad():
            pm = rend.run_model(planes, dec, pts, None, cs['opts'])
        arrays = dict(planes=planes, c2w=c2w, K=K, res=np.int64(res), ray_o=ray_o, ray_d=ray_d, u_coarse=tape.draws[0], u_fine=tape.draws[1],
                      feat=feat, depth=depth, wsum=wsum, pts=pts, pts_rgb=pm['rgb'], pts_sigma=pm['sigma'],
                      nets=np.int64(cs['nets']), sem_sigmoid=np.int64(cs['sem_sigmoid']), lr_mul=np.float64(cs['lr_mul']),
                      opt_keys=np.array(list(cs['opts'].keys())), opt_vals=np.array([str(v) for v in cs['opts'].values()]), **rec)
        arrays.update({'dec_' + k: v for k, v in _decoder_arrays(dec).items()})
        save('renderer_' + cs['name'], **arrays)

    # sample_pdf alone, with the indices it derives (the "bit-exact index work" fixture)
    torch.manual_seed(7)
    rend = ImportanceRenderer()
    rays, sc, sf = 64, 48, 48
    z = torch.sort(torch.rand(rays, sc) * 1.05 + 2.25, dim=1)[0]
    wts = torch.rand(rays, sc - 1) ** 4
    wts[:8] = 0                                  # empty rays: uniform pdf
    wts[8:12, 5:] = 0                            # everything in the first bins
    with _RandTape() as tape, torch.no_grad():
        zf = rend.sample_importance(z.reshape(1, rays, sc, 1), wts.reshape(1, rays, sc - 1, 1), sf)
    save('renderer_importance', z=z, w=wts, u=tape.draws[0], z_fine=zf.reshape(rays, sf))


GROUPS['renderer'] = group_renderer


def group_semrenderer():
    """ImportanceSemanticRenderer (renderer.py:256-438): two plane sets, OSGDecoder on cat(texture, semantic) features for colour and
    OSGDecoder_semantic on the semantic planes for density + labels.  Not selected by train.py any more; recorded for parity of the
    host-side mirror."""
    from training.volumetric_rendering.renderer import ImportanceSemanticRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.triplane import OSGDecoder
    from training.triplane_cond import OSGDecoder_semantic
    cases = [
        dict(name='a', sem_sigmoid=False, n=2, hw=(12, 14), res=5, focal=4.2647, radius=2.7, lr_mul=1.0,
             opts=dict(depth_resolution=10, depth_resolution_importance=8, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode='softplus')),
        dict(name='b', sem_sigmoid=True, n=1, hw=(16, 16), res=6, focal=1.2, radius=1.7, lr_mul=0.5,
             opts=dict(depth_resolution=12, depth_resolution_importance=0, ray_start=0.1, ray_end=2.6, box_warp=1.6, disparity_space_sampling=False, clamp_mode='softplus', white_back=True)),
    ]
    for ci, cs in enumerate(cases):
        torch.manual_seed(300 + ci)
        n, (h, w), res = cs['n'], cs['hw'], cs['res']
        planes_t, planes_s = torch.randn(n, 3, 32, h, w), torch.randn(n, 3, 32, h, w)
        dec_t = OSGDecoder(64, {'decoder_lr_mul': cs['lr_mul'], 'decoder_output_dim': 32})
        dec_s = OSGDecoder_semantic(32, {'decoder_lr_mul': cs['lr_mul'], 'decoder_output_dim': 32, 'sigmoid': cs['sem_sigmoid']})
        with torch.no_grad():
            for d_ in (dec_t, dec_s):
                for p_ in d_.parameters():
                    if p_.ndim == 1:
                        p_.copy_(torch.randn_like(p_) * 0.3 / cs['lr_mul'])
        c2w = torch.tensor(np.stack([_look_at(cs['radius'], 0.2 + 0.8 * i, 1.45 - 0.2 * i) for i in range(n)]))
        K = torch.tensor([[cs['focal'], 0.0, 0.5], [0, cs['focal'], 0.5], [0, 0, 1]], dtype=torch.float32).repeat(n, 1, 1)
        ray_o, ray_d = RaySampler()(c2w, K, res)
        rend = ImportanceSemanticRenderer()
        with _RandTape() as tape, torch.no_grad():
            feat, depth, wsum = rend(planes_t, planes_s, dec_t, dec_s, ray_o, ray_d, cs['opts'])
        pts = (torch.rand(n, 30, 3) - 0.5) * 1.3 * cs['opts']['box_warp']
        with torch.no_grad():
            pm = rend.run_model(planes_t, planes_s, dec_t, dec_s, pts, None, cs['opts'])
        arrays = dict(planes_t=planes_t, planes_s=planes_s, ray_o=ray_o, ray_d=ray_d, u_coarse=tape.draws[0],
                      u_fine=tape.draws[1] if len(tape.draws) > 1 else np.zeros([0], np.float32), feat=feat, depth=depth, wsum=wsum,
                      pts=pts, pts_rgb=pm['rgb'], pts_sigma=pm['sigma'], pts_semantic=pm['semantic'],
                      sem_sigmoid=np.int64(cs['sem_sigmoid']), lr_mul=np.float64(cs['lr_mul']),
                      opt_keys=np.array(list(cs['opts'].keys())), opt_vals=np.array([str(v) for v in cs['opts'].values()]))
        arrays.update({'dect_' + k: v for k, v in _decoder_arrays(dec_t).items()})
        arrays.update({'decs_' + k: v for k, v in _decoder_arrays(dec_s).items()})
        save('semrenderer_' + cs['name'], **arrays)


GROUPS['semrenderer'] = group_semrenderer


# ---------------------------------------------------------------------------------------------------------
def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def _thumb(t, step):
    """Strided thumbnail + an exact central crop: enough to pin a 512^2 image in a few KiB."""
    h = t.shape[-1]
    c0 = h // 2 - 16
    return t[..., ::step, ::step].contiguous(), t[..., c0:c0 + 32, c0:c0 + 32].contiguous()


def group_model():
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    runs = [dict(name='seg2cat', nrr=32, n=1, frames=[7]), dict(name='edge2car', nrr=64, n=2, frames=[3, 40])]
    for run in runs:
        kw = configs.generator_kwargs(run['name'])
        info = configs.dataset_info(run['name'])
        torch.manual_seed(0)
        G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
        weights.seed_module(G, seed=1)
        n = run['n']
        gz = torch.Generator().manual_seed(5)
        ws = torch.randn(n, G.backbone.num_ws, 512, generator=gz)
        rk = kw['rendering_kwargs']
        c = torch.tensor(np.stack([configs.orbit_camera(k, radius=rk['avg_camera_radius'], focal=4.2647 if run['name'] != 'edge2car' else 1.7074,
                                                        pivot=rk['avg_camera_pivot']) for k in run['frames']]))
        # the renderer's two uniform draws are NOT stored (MBs of incompressible floats): they are the first draws of
        # the CPU generator after manual_seed(RENDER_SEED), in shapes [N,M,Sc,1] then [N*M,Sf]; tests re-draw them
        torch.manual_seed(4321)
        with _RandTape() as tape, torch.no_grad():
            out = G.synthesis(ws, c, neural_rendering_resolution=run['nrr'], noise_mode='const')
        assert len(tape.draws) == 2
        arrays = dict(ws=ws, c=c, nrr=np.int64(run['nrr']), render_seed=np.int64(4321), u_coarse_head=tape.draws[0].reshape(-1)[:16], u_fine_head=tape.draws[1].reshape(-1)[:16],
                      image_raw=out['image_raw'], image_depth=out['image_depth'], semantic_raw=out['semantic_raw'])
        step = max(out['image'].shape[-1] // 64, 1)
        arrays['image_thumb'], arrays['image_crop'] = _thumb(out['image'], step)
        arrays['semantic_thumb'], arrays['semantic_crop'] = _thumb(out['semantic'], step)
        arrays['thumb_step'] = np.int64(step)
        # mapping network (G.mapping) on a synthetic conditioning image
        gm = torch.Generator().manual_seed(9)
        z = torch.randn(n, 512, generator=gm)
        if info['data_type'] == 'seg':
            mask = torch.randint(0, info['sem'], [n, 1, info['res'], info['res']], generator=gm)
        else:
            mask = torch.rand([n, 1, info['res'], info['res']], generator=gm) * 2 - 1
        with torch.no_grad():
            wsm = G.mapping(z, c, {'mask': mask, 'pose': c})
        arrays.update(map_z=z, map_mask=mask.to(torch.int16 if info['data_type'] == 'seg' else torch.float32), map_ws=wsm)
        pts = (torch.rand(n, 64, 3, generator=gm) - 0.5) * rk['box_warp']
        with torch.no_grad():
            sm = G.sample_mixed(pts, None, ws, noise_mode='const')
        arrays.update(pts=pts, pts_rgb=sm['rgb'], pts_sigma=sm['sigma'])
        save('model_' + run['name'], **arrays)


GROUPS['model'] = group_model


def group_model_full():
    """The BASELINE configurations at their real size (SURVEY §8 shape table): seg2cat batch 4 at 128^2 rays with 48+48 (cfg 2/3)
    and 64+64 samples (the metric), seg2face (19 label channels, cfg 5) batch 2 at 48+48.  Kept small: strided thumbnails + exact
    central crops of every output (a flipped importance bin or a wrong ray anywhere in the crop / on the thumbnail lattice shows up
    at the 1e-3 level), plus per-image means of every output, which see all 16 384 rays."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    runs = [dict(tag='seg2cat_96', name='seg2cat', n=4, depth=(48, 48), frames=[3, 10, 17, 24]),
            dict(tag='seg2cat_128', name='seg2cat', n=4, depth=(64, 64), frames=[3, 10, 17, 24]),
            dict(tag='seg2face_96', name='seg2face', n=2, depth=(48, 48), frames=[5, 70]),
            # BASELINE configs[3] per GPU (train.py:451-461): edge2car, 64^2 rays x 64+64, batch 8, Hybrid2X heads -> 128^2, white background, sigmoid labels
            dict(tag='edge2car_128', name='edge2car', n=8, depth=(64, 64), nrr=64, focal=1.7074, frames=[3, 18, 33, 48, 63, 78, 93, 108])]
    only = os.environ.get('P3D_GOLDEN_ONLY')                   # e.g. P3D_GOLDEN_ONLY=edge2car_128: record just that run (the others are unchanged)
    for run in runs:
        if only and run['tag'] not in only.split(','):
            continue
        kw = configs.generator_kwargs(run['name'], depth=run['depth'])
        torch.manual_seed(0)
        G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
        weights.seed_module(G, seed=1)
        n, nrr = run['n'], run.get('nrr', 128)
        gz = torch.Generator().manual_seed(5)
        ws = torch.randn(n, G.backbone.num_ws, 512, generator=gz)
        rk = kw['rendering_kwargs']
        c = torch.tensor(np.stack([configs.orbit_camera(k, radius=rk['avg_camera_radius'], focal=run.get('focal', 4.2647), pivot=rk['avg_camera_pivot']) for k in run['frames']]))
        torch.manual_seed(4321)
        with _RandTape() as tape, torch.no_grad():
            out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
        assert len(tape.draws) == 2
        arrays = dict(ws=ws, c=c, nrr=np.int64(nrr), depth=np.array(run['depth']), render_seed=np.int64(4321),
                      u_coarse_head=tape.draws[0].reshape(-1)[:16], u_fine_head=tape.draws[1].reshape(-1)[:16])
        for k in ('image_raw', 'semantic_raw', 'image_depth', 'image', 'semantic'):
            t = out[k]
            step = max(t.shape[-1] // 32, 1)
            arrays[k + '_thumb'], arrays[k + '_crop'] = _thumb(t, step)
            arrays[k + '_step'] = np.int64(step)
            arrays[k + '_mean'] = t.double().mean(dim=[2, 3])
            arrays[k + '_absmax'] = t.abs().max()
            # every pixel of the output is in exactly one 8x8 tile's sum and abs-max (4x4 for the 128^2 renderings): a defect ANYWHERE moves a record
            tile = 8 if t.shape[-1] >= 512 else 4
            v = t.double().reshape(t.shape[0], t.shape[1], t.shape[2] // tile, tile, t.shape[3] // tile, tile)
            arrays[k + '_tile_sum'], arrays[k + '_tile_max'] = v.sum(dim=(3, 5)).float(), v.abs().amax(dim=(3, 5)).float()
            arrays[k + '_tile'] = np.int64(tile)
        save('model_full_' + run['tag'], **arrays)


GROUPS['model_full'] = group_model_full


def group_flrelu():
    from torch_utils.ops import filtered_lrelu, upfirdn2d
    g = torch.Generator().manual_seed(77)
    cases = [dict(up=2, down=2, fu=12, fd=12, pad=[9, 10, 9, 10], clamp=None, flip=False), dict(up=1, down=1, fu=1, fd=1, pad=0, clamp=0.8, flip=False),
             dict(up=2, down=1, fu=8, fd=1, pad=[3, 4, 3, 4], clamp=1.2, flip=True), dict(up=1, down=2, fu=1, fd=6, pad=[2, 3, 2, 3], clamp=None, flip=False)]
    out = {'num': np.int64(len(cases))}
    for i, cs in enumerate(cases):
        x = torch.randn(2, 3, 10, 9, generator=g, dtype=torch.float64).requires_grad_(True)
        b = torch.randn(3, generator=g, dtype=torch.float64)
        fu = upfirdn2d.setup_filter(torch.randn(cs['fu'], generator=g).tolist()) if cs['fu'] > 1 else None
        fd = upfirdn2d.setup_filter(torch.randn(cs['fd'], generator=g).tolist()) if cs['fd'] > 1 else None
        y = filtered_lrelu._filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=cs['up'], down=cs['down'], padding=cs['pad'], gain=1.3, slope=0.15, clamp=cs['clamp'], flip_filter=cs['flip'])
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        gx, = torch.autograd.grad(y, x, gy)
        out.update({f'{i}.x': x, f'{i}.b': b, f'{i}.y': y, f'{i}.gy': gy, f'{i}.gx': gx,
                    f'{i}.fu': fu if fu is not None else np.zeros([0], np.float32), f'{i}.fd': fd if fd is not None else np.zeros([0], np.float32),
                    f'{i}.cfg': np.array([cs['up'], cs['down'], int(cs['flip'])]), f'{i}.pad': np.array(cs['pad'] if isinstance(cs['pad'], list) else [cs['pad']] * 4),
                    f'{i}.clamp': np.float64(-1 if cs['clamp'] is None else cs['clamp'])})
    save('ops_filtered_lrelu', **out)


GROUPS['flrelu'] = group_flrelu


def group_train():
    """Training-mode forward + backward of the reference generator (unfused modconv, tensor-op renderer): gradients of a
    scalar loss w.r.t. a few parameters, with the renderer's uniforms seeded as in group_model.  Two losses: one that also
    differentiates the depth map, and one over images only (what the training losses use) — the second lets the product's
    fused renderer backward, which produces no depth gradient, be checked at model level."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    for fname, with_depth in (('train_seg2cat', True), ('train_seg2cat_nodepth', False)):
        kw = configs.generator_kwargs('seg2cat')
        kw['rendering_kwargs'] = dict(kw['rendering_kwargs'], depth_resolution=8, depth_resolution_importance=8)
        torch.manual_seed(0)
        G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
        weights.seed_module(G, seed=1)
        gz = torch.Generator().manual_seed(5)
        ws = torch.randn(1, G.backbone.num_ws, 512, generator=gz)
        c = torch.tensor(np.stack([configs.orbit_camera(11, radius=2.7, pivot=[0, 0, -0.06])]))
        torch.manual_seed(4321)
        out = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const')
        loss = out['image'].mean() + out['image_raw'].square().mean() + out['semantic'].square().mean() * 0.1
        if with_depth:
            loss = loss + out['image_depth'].mean()
        loss.backward()
        names = ['backbone.synthesis.b4.const', 'backbone.synthesis.b256.conv1.weight', 'backbone.synthesis.b64.conv0.affine.bias',
                 'backbone.synthesis.b256.torgb.weight', 'decoder.net.0.weight', 'decoder.net_semantic.2.bias',
                 'superresolution.block1.conv1.weight', 'superresolution.block0.conv0.bias', 'superresolution_semantic.block1.torgb.bias']
        params = dict(G.named_parameters())
        arrays = dict(ws=ws, c=c, loss=loss.detach(), names=np.array(names), render_seed=np.int64(4321), with_depth=np.int64(with_depth))
        for i, nme in enumerate(names):
            g = params[nme].grad
            arrays[f'g{i}.norm'] = g.norm()
            arrays[f'g{i}.head'] = g.reshape(-1)[:64].clone()
        save(fname, **arrays)


GROUPS['train'] = group_train


def _grad_record(G, heads):
    """Norm of EVERY parameter gradient (name-ordered) + the first 64 entries of the named ones."""
    names = [n for n, _ in G.named_parameters()]
    params = dict(G.named_parameters())
    norms = np.array([float(params[n].grad.double().norm()) if params[n].grad is not None else -1.0 for n in names], dtype=np.float64)
    arrays = {'grad_names': np.array(names), 'grad_norms': norms, 'head_names': np.array(heads)}
    for i, nme in enumerate(heads):
        arrays[f'h{i}'] = params[nme].grad.reshape(-1)[:64].clone()
    return arrays


def group_train_full():
    """BASELINE config 3 at its real size, one Gmain-style pass of the reference on the CPU: seg2cat, batch 2, 128^2 rays x 48+48,
    G.mapping (label map -> Encoder -> ws, loss.py:440) + G.synthesis in training mode (unfused modulation, tensor-op renderer under
    autograd), a scalar loss over image / semantic / image_raw, backward through everything.  Records the loss, the gradient norm of
    every parameter and heads of a dozen gradients (backbone, Encoder of G.mapping, decoder, both SR heads)."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
    info = configs.dataset_info('seg2cat')
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
    weights.seed_module(G, seed=1)
    n, nrr = 2, 128
    gz = torch.Generator().manual_seed(15)
    z = torch.randn(n, 512, generator=gz)
    mask = torch.randint(0, info['sem'], [n, 1, info['res'], info['res']], generator=gz)
    rk = kw['rendering_kwargs']
    c = torch.tensor(np.stack([configs.orbit_camera(k, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in (3, 24)]))
    ws = G.mapping(z, c, {'mask': mask, 'pose': c}, update_emas=False)
    torch.manual_seed(4321)
    with _RandTape() as tape:
        out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
    assert len(tape.draws) == 2
    loss = out['image'].square().mean() + out['semantic'].square().mean() * 0.1 + out['image_raw'].square().mean()
    loss.backward()
    heads = ['backbone.synthesis.b4.const', 'backbone.synthesis.b256.conv1.weight', 'backbone.synthesis.b64.conv0.affine.bias',
             'backbone.synthesis.b256.torgb.weight', 'decoder.net.0.weight', 'decoder.net_semantic.2.bias', 'decoder.net_semantic.0.weight',
             'superresolution.block1.conv1.weight', 'superresolution.block0.conv0.bias', 'superresolution_semantic.block1.torgb.bias']
    enc = [nm for nm, p in G.named_parameters() if nm.startswith('backbone.mapping') and p.grad is not None and p.ndim >= 2]
    heads += [enc[0], enc[len(enc) // 2], enc[-1]]
    arrays = dict(z=z, c=c, mask=mask.to(torch.int16), nrr=np.int64(nrr), depth=np.array([48, 48]), render_seed=np.int64(4321), loss=loss.detach(), ws=ws.detach(),
                  u_coarse_head=tape.draws[0].reshape(-1)[:16], u_fine_head=tape.draws[1].reshape(-1)[:16])
    for k in ('image_raw', 'semantic_raw', 'image', 'semantic'):
        arrays[k + '_mean'] = out[k].detach().double().mean(dim=[2, 3])
    arrays.update(_grad_record(G, heads))
    save('train_full_seg2cat', **arrays)


GROUPS['train_full'] = group_train_full


def group_greg():
    """The density-regularisation phase ('Greg', loss.py:681-706, reg_type 'l1') of the reference on the CPU: ws = G.mapping(...),
    1000 uniform points per image + their perturbed copies, sigma = G.sample_mixed(...)['sigma'], L1 between the halves x density_reg,
    backward.  The points and the perturbation are recorded (the loss draws them with the global generator)."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    kw = configs.generator_kwargs('seg2cat', depth=(48, 48))
    info = configs.dataset_info('seg2cat')
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
    weights.seed_module(G, seed=1)
    n = 2
    gz = torch.Generator().manual_seed(25)
    z = torch.randn(n, 512, generator=gz)
    mask = torch.randint(0, info['sem'], [n, 1, info['res'], info['res']], generator=gz)
    rk = kw['rendering_kwargs']
    c = torch.tensor(np.stack([configs.orbit_camera(k, radius=rk['avg_camera_radius'], pivot=rk['avg_camera_pivot']) for k in (5, 50)]))
    ws = G.mapping(z, c, {'mask': mask, 'pose': c}, update_emas=False)
    torch.manual_seed(777)
    initial = torch.rand((n, 1000, 3)) * 2 - 1
    perturbed = initial + torch.randn_like(initial) * rk['density_reg_p_dist']
    coords = torch.cat([initial, perturbed], dim=1)
    res = G.sample_mixed(coords, torch.randn_like(coords), ws, update_emas=False, noise_mode='const')
    sigma = res['sigma']
    s_i, s_p = sigma[:, :sigma.shape[1] // 2], sigma[:, sigma.shape[1] // 2:]
    loss = torch.nn.functional.l1_loss(s_i, s_p) * rk['density_reg']
    loss.backward()
    heads = ['backbone.synthesis.b4.const', 'backbone.synthesis.b256.conv1.weight', 'backbone.synthesis.b256.torgb.weight', 'backbone.synthesis.b256.torgb.bias',
             'decoder.net_semantic.0.weight', 'decoder.net_semantic.0.bias', 'decoder.net_semantic.2.weight', 'decoder.net_semantic.2.bias']
    enc = [nm for nm, p in G.named_parameters() if nm.startswith('backbone.mapping') and p.grad is not None and p.ndim >= 2]
    heads += [enc[0], enc[-1]]
    arrays = dict(z=z, c=c, mask=mask.to(torch.int16), coords=coords, loss=loss.detach(), sigma=sigma.detach(), rgb_head=res['rgb'].detach()[:, :8], ws=ws.detach())
    arrays.update(_grad_record(G, heads))
    # second record: a loss that also uses the colours (the general autograd contract of run_model)
    for p in G.parameters():
        p.grad = None
    ws2 = G.mapping(z, c, {'mask': mask, 'pose': c}, update_emas=False)
    res2 = G.sample_mixed(coords, None, ws2, update_emas=False, noise_mode='const')
    loss2 = res2['rgb'].square().mean() + res2['sigma'].square().mean() * 1e-3
    loss2.backward()
    rec2 = _grad_record(G, heads)
    arrays.update({'rgbloss': loss2.detach(), 'rgbloss_grad_norms': rec2['grad_norms'], **{f'rgbloss_h{i}': rec2[f'h{i}'] for i in range(len(heads))}})
    save('greg_seg2cat', **arrays)


GROUPS['greg'] = group_greg


def group_legacy_classes():
    """Two reference classes no shipped configuration reaches but checkpoints can name: OSGDecoder_semantic_entangle
    (triplane_cond.py:891-924) and SuperresolutionHybridDeepfp32 (superresolution.py:160-188)."""
    import dnnlib
    from training.triplane_cond import OSGDecoder_semantic_entangle
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    arrays = {}
    gz = torch.Generator().manual_seed(91)
    feats = torch.randn(2, 3, 50, 32, generator=gz)
    for tag, sig in (('raw', False), ('sigmoid', True)):
        dec = OSGDecoder_semantic_entangle(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32, 'sigmoid': sig, 'semantic_channels': 6}).requires_grad_(False)
        weights.seed_module(dec, seed=71)
        out = dec(feats, None)
        arrays[f'dec.{tag}.rgb'], arrays[f'dec.{tag}.sigma'] = out['rgb'], out['sigma']
    arrays['dec.feats_head'] = feats.reshape(-1)[:16].clone()
    torch.manual_seed(0)
    sr = dnnlib.util.construct_class_by_name(class_name='training.superresolution.SuperresolutionHybridDeepfp32', channels=32, img_resolution=256, sr_num_fp16_res=4,
                                             channel_base=32768, channel_max=512, fused_modconv_default='inference_only').eval().requires_grad_(False)
    weights.seed_module(sr, seed=72)
    for tag, side in (('same', 128), ('small', 96)):
        x = torch.randn(1, 32, side, side, generator=gz)
        ws = torch.randn(1, 14, 512, generator=gz)
        with torch.no_grad():
            y = sr(x[:, :3].clone(), x, ws, noise_mode='const')
        arrays[f'sr.{tag}.x_head'], arrays[f'sr.{tag}.ws_head'] = x.reshape(-1)[:16].clone(), ws.reshape(-1)[:16].clone()      # inputs are re-drawn from the seed by the tests
        arrays[f'sr.{tag}.thumb'], arrays[f'sr.{tag}.crop'] = _thumb(y, 8)
    save('legacy_classes', **arrays)


GROUPS['legacy_classes'] = group_legacy_classes


def _tile_large(module, period=512, limit=1024):
    """Replace every tensor above ``limit`` elements by a tiling of its first ``period`` values: the checkpoint then xz-compresses
    to a few hundred KiB (the label-map Encoder alone is 204 MB of fixed 512-channel layers) while every value stays name-seeded."""
    with torch.no_grad():
        for _, t in list(module.named_parameters()) + list(module.named_buffers()):
            if t.numel() > limit and t.is_floating_point():
                flat = t.reshape(-1)
                reps = -(-flat.numel() // period)
                flat.copy_(flat[:period].clone().repeat(reps)[:flat.numel()])


def group_checkpoint():
    """A checkpoint in the reference's wire format (training_loop.py:455-470: pickle of dict(G, D, G_ema, D_semantic, augment_pipe,
    training_set_kwargs) with persistence-decorated networks), written AND read back by the reference (legacy.load_network_pkl), plus
    what the reloaded networks compute on fixed inputs.  tests/test_checkpoint*.py load the same file through pix2pix3d_amd.legacy."""
    import lzma
    import pickle
    import dnnlib
    import legacy
    from training.augment import AugmentPipe
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    torch.manual_seed(0)
    kw = configs.generator_kwargs('edge2car', cbase=1024, cmax=16)
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
    weights.seed_module(G, seed=3); _tile_large(G)
    G_ema = G                                                   # one copy on the wire (pickle memo); both keys must come back
    G.neural_rendering_resolution = 24                          # assigned after construction by the training loop (loss.py)
    dkw = dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=128, img_channels=3,
               channel_base=1024, channel_max=16, num_fp16_res=4, conv_clamp=256, disc_c_noise=0, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2))
    D = dnnlib.util.construct_class_by_name(**dkw).train().requires_grad_(True)
    weights.seed_module(D, seed=4); _tile_large(D)
    ekw = dict(class_name='training.triplane.TriPlaneGenerator', z_dim=512, w_dim=512, c_dim=25, img_resolution=128, img_channels=3,
               mapping_kwargs=dict(num_layers=2), rendering_kwargs=kw['rendering_kwargs'], channel_base=1024, channel_max=16,
               fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None, sr_num_fp16_res=4,
               sr_kwargs=dict(channel_base=1024, channel_max=16, fused_modconv_default='inference_only'))
    E = dnnlib.util.construct_class_by_name(**ekw).eval().requires_grad_(False)
    weights.seed_module(E, seed=5); _tile_large(E)
    aug = AugmentPipe(xflip=1, rotate90=1, xint=1).train().requires_grad_(False)
    snapshot = dict(G=G, D=D, G_ema=G_ema, eg3d=E, augment_pipe=aug, training_set_kwargs=dnnlib.EasyDict(class_name='training.dataset.ImageSegFolderDataset', resolution=128, use_labels=True))
    blob = pickle.dumps(snapshot)
    path = os.path.join(HERE, 'checkpoint_small.pkl.xz')
    with lzma.open(path, 'wb', preset=6) as f:
        f.write(blob)
    print(f'wrote {path}: {len(blob) / 2**20:.1f} MiB pickled, {os.path.getsize(path) / 1024:.1f} KiB compressed')

    with lzma.open(path, 'rb') as f:
        data = legacy.load_network_pkl(f)
    G2, D2, E2 = data['G_ema'].eval().requires_grad_(False), data['D'].eval().requires_grad_(False), data['eg3d']
    assert data['G'] is data['G_ema'] and G2.neural_rendering_resolution == 24
    gz = torch.Generator().manual_seed(11)
    n = 2
    c = torch.tensor(np.stack([configs.orbit_camera(k, radius=1.7, focal=1.7074) for k in (5, 70)]))
    z = torch.randn(n, 512, generator=gz)
    mask = torch.rand([n, 1, 128, 128], generator=gz) * 2 - 1
    arrays = dict(c=c, z=z, mask=mask, render_seed=np.int64(99))
    with torch.no_grad():
        ws = G2.mapping(z, c, {'mask': mask, 'pose': c})
        torch.manual_seed(99)
        out = G2.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const')
        logits = D2({'image': out['image'], 'image_raw': out['image_raw']}, c)
        ws_e = E2.mapping(z, c)
        torch.manual_seed(99)
        out_e = E2.synthesis(ws_e, c, neural_rendering_resolution=16, noise_mode='const')
    arrays.update(ws=ws, image=out['image'], image_raw=out['image_raw'], image_depth=out['image_depth'], semantic=out['semantic'],
                  semantic_raw=out['semantic_raw'], logits=logits, eg3d_ws=ws_e, eg3d_image=out_e['image'], eg3d_image_raw=out_e['image_raw'],
                  eg3d_image_depth=out_e['image_depth'])
    sd = G2.state_dict()
    arrays['param_names'] = np.array(sorted(sd))
    arrays['param_sums'] = np.array([float(sd[k].double().sum()) for k in sorted(sd)])
    arrays['d_param_sums'] = np.array([float(v.double().sum()) for k, v in sorted(D2.state_dict().items())])
    arrays['aug_buffers'] = np.array(sorted(k for k, _ in aug.named_buffers()))
    save('checkpoint_small', **arrays)


GROUPS['checkpoint'] = group_checkpoint


def group_variants():
    """The generator classes outside train.py's current selection (configs.variant_kwargs): mapping + synthesis + point queries of small
    name-seeded instances."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    arrays = {}
    for which in configs.VARIANTS:
        kw = configs.variant_kwargs(which)
        torch.manual_seed(0)
        G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
        weights.seed_module(G, seed=7)
        gz = torch.Generator().manual_seed(21)
        n = 1
        c = torch.tensor(np.stack([configs.orbit_camera(33, radius=1.7, focal=1.7074)]))
        z = torch.randn(n, 512, generator=gz)
        if kw['mapping_kwargs']['in_channels'] > 1:
            mask = torch.randint(0, 6, [n, 1, 128, 128], generator=gz)
        else:
            mask = torch.rand([n, 1, 128, 128], generator=gz) * 2 - 1
        pts = (torch.rand(n, 32, 3, generator=gz) - 0.5) * kw['rendering_kwargs']['box_warp']
        with torch.no_grad():
            ws = G.mapping(z, c, {'mask': mask, 'pose': c})
            torch.manual_seed(77)
            out = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const')
            sm = G.sample_mixed(pts, None, ws, noise_mode='const')
        a = dict(c=c, z=z, mask=mask.to(torch.int16) if mask.dtype == torch.int64 else mask, pts=pts, ws=ws, pts_sigma=sm['sigma'], pts_rgb=sm['rgb'])
        if 'semantic' in sm:
            a['pts_semantic'] = sm['semantic']
        for k, v in out.items():
            if v.shape[-1] > 32:
                a[k + '_thumb'], a[k + '_crop'] = _thumb(v, 4)
            else:
                a[k] = v
        arrays.update({f'{which}.{k}': v for k, v in a.items()})
        print(which, {k: tuple(v.shape) for k, v in out.items()})
    arrays['render_seed'] = np.int64(77)
    save('model_variants', **arrays)


GROUPS['variants'] = group_variants


def discriminator_case(D, conv2d_gradfix, g_in):
    """The 'Dboth' phase on real images (loss.py:327-367): logits, softplus(-logits), R1 gradients w.r.t. both input images under
    ``no_weight_gradients`` with create_graph, (loss_Dreal + r1_penalty * gamma/2).mean().backward().  Shared by the golden run (reference
    modules) and nothing else — the tests restate it against this package's modules."""
    img = {'image': g_in['image'].clone().requires_grad_(True), 'image_raw': g_in['image_raw'].clone().requires_grad_(True)}
    logits = D(img, g_in['c'])
    loss_real = torch.nn.functional.softplus(-logits)
    with conv2d_gradfix.no_weight_gradients():
        g_img, g_raw = torch.autograd.grad(outputs=[logits.sum()], inputs=[img['image'], img['image_raw']], create_graph=True, only_inputs=True)
    r1 = g_img.square().sum([1, 2, 3]) + g_raw.square().sum([1, 2, 3])
    (loss_real + r1 * (10 / 2)).mean().backward()
    return logits, g_img, g_raw, r1


def group_discriminator():
    import dnnlib
    from torch_utils.ops import conv2d_gradfix
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    arrays = {}
    cases = dict(dual=dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=64, img_channels=3, channel_base=1024,
                           channel_max=32, num_fp16_res=0, conv_clamp=None, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={},
                           epilogue_kwargs=dict(mbstd_group_size=2)),
                 dual_clamp=dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=32, img_channels=1, channel_base=512,
                                 channel_max=16, num_fp16_res=0, conv_clamp=0.5, architecture='resnet', epilogue_kwargs=dict(mbstd_group_size=4, mbstd_num_channels=2)),
                 single=dict(class_name='training.dual_discriminator.SingleDiscriminator', c_dim=0, img_resolution=32, img_channels=3, channel_base=512,
                             channel_max=16, num_fp16_res=0, conv_clamp=None))
    for name, kw in cases.items():
        torch.manual_seed(0)
        D = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(True)
        weights.seed_module(D, seed=9)
        gz = torch.Generator().manual_seed(31)
        res, ch = kw['img_resolution'], kw['img_channels']
        g_in = dict(image=torch.randn(4, ch, res, res, generator=gz), image_raw=torch.randn(4, ch, res // 4, res // 4, generator=gz), c=torch.randn(4, 25, generator=gz))
        if name == 'single':
            img = {'image': g_in['image'].clone().requires_grad_(True)}
            logits = D(img, None)
            with conv2d_gradfix.no_weight_gradients():
                g_img, = torch.autograd.grad(outputs=[logits.sum()], inputs=[img['image']], create_graph=True, only_inputs=True)
            r1 = g_img.square().sum([1, 2, 3])
            (torch.nn.functional.softplus(-logits) + r1 * 5).mean().backward()
            g_raw = torch.zeros(1)
        else:
            logits, g_img, g_raw, r1 = discriminator_case(D, conv2d_gradfix, g_in)
        a = dict(image=g_in['image'], image_raw=g_in['image_raw'], c=g_in['c'], logits=logits.detach(), g_img=g_img.detach(), g_raw=g_raw.detach(), r1=r1.detach())
        names = [n for n, p in D.named_parameters() if p.grad is not None]
        a['grad_names'] = np.array(names)
        a['grad_norms'] = np.array([float(dict(D.named_parameters())[n].grad.double().norm()) for n in names])
        first = names[0]
        a['grad_head'] = dict(D.named_parameters())[first].grad.reshape(-1)[:64].clone()
        arrays.update({f'{name}.{k}': v for k, v in a.items()})
        print(name, tuple(logits.shape), len(names), 'grads')
    save('discriminator', **arrays)


GROUPS['discriminator'] = group_discriminator


def group_api():
    """Entry points and switches around the hot path that the other groups leave at their defaults: truncation, w_avg tracking, the plane cache,
    random noise, zeroed / scaled camera conditioning, density noise, G.forward / G.sample, the discriminator's camera noise.  Small
    name-seeded edge2car-style generator (configs.variant_kwargs base), everything seeded so that the draws can be replayed."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    kw = configs.generator_kwargs('edge2car', cbase=1024, cmax=16)
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
    weights.seed_module(G, seed=13)
    gz = torch.Generator().manual_seed(41)
    n = 2
    c = torch.tensor(np.stack([configs.orbit_camera(k, radius=1.7, focal=1.7074) for k in (9, 50)]))
    z = torch.randn(n, 512, generator=gz)
    mask = torch.rand([n, 1, 128, 128], generator=gz) * 2 - 1
    batch = {'mask': mask, 'pose': c}
    pts = (torch.rand(n, 48, 3, generator=gz) - 0.5) * kw['rendering_kwargs']['box_warp']
    a = dict(c=c, z=z, mask=mask, pts=pts)
    with torch.no_grad():
        # (truncation_cutoff cannot be used with the per-layer w_avg of the disentangled mapping networks: triplane_cond.py:591 raises — tested as such)
        a['ws_trunc_all'] = G.mapping(z, c, batch, truncation_psi=0.4)
        # w_avg tracking (training_loop calls mapping with update_emas=True on G_ema copies): one update from the seeded state
        before = G.backbone.mapping.w_avg.clone()
        G.mapping(z, c, batch, update_emas=True)
        a['w_avg_after'] = G.backbone.mapping.w_avg.clone()
        G.backbone.mapping.w_avg.copy_(before)
        ws = G.mapping(z, c, batch)
        a['ws'] = ws
        # G.forward = mapping(z, batch['pose']) + synthesis(ws, c)
        torch.manual_seed(5)
        out = G(z, c, batch, neural_rendering_resolution=16, noise_mode='const')
        a['fwd_image_raw'], a['fwd_semantic_raw'] = out['image_raw'], out['semantic_raw']
        # plane cache: the second call must ignore its ws for the planes (but not for the SR heads)
        torch.manual_seed(5)
        G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const', cache_backbone=True)
        torch.manual_seed(5)
        out2 = G.synthesis(ws.flip(0), c, neural_rendering_resolution=16, noise_mode='const', use_cached_backbone=True)
        a['cached_image_raw'], a['cached_image_thumb'] = out2['image_raw'], out2['image'][..., ::4, ::4]
        G._last_planes = None
        # random noise: per-layer torch.randn draws, replayable from the seed on CPU
        torch.manual_seed(6)
        out3 = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='random')
        a['rand_image_raw'], a['rand_image_thumb'] = out3['image_raw'], out3['image'][..., ::4, ::4]
        # point queries through z, with density noise
        G.rendering_kwargs['density_noise'] = 0.5
        torch.manual_seed(7)
        sm = G.sample(pts, None, z, c, batch, noise_mode='const')
        a['sample_sigma'], a['sample_rgb'] = sm['sigma'], sm['rgb']
        G.rendering_kwargs['density_noise'] = 0
        # camera conditioning switches
        G.rendering_kwargs['c_gen_conditioning_zero'] = True
        a['ws_czero'] = G.mapping(z, c, batch)
        G.rendering_kwargs['c_gen_conditioning_zero'] = False
        G.rendering_kwargs['c_scale'] = 0.25
        a['ws_cscale'] = G.mapping(z, c, batch)
        G.rendering_kwargs['c_scale'] = 1.0
        # discriminator with camera noise
        D = dnnlib.util.construct_class_by_name(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=128, img_channels=3,
                                                channel_base=1024, channel_max=16, num_fp16_res=0, conv_clamp=None, disc_c_noise=0.5).eval().requires_grad_(False)
        weights.seed_module(D, seed=14)
        torch.manual_seed(8)
        a['d_logits_cnoise'] = D({'image': out['image'], 'image_raw': out['image_raw']}, c.clone())
        a['d_image'], a['d_image_raw'] = out['image'], out['image_raw']
    save('model_api', **a)


GROUPS['api'] = group_api


SR_CASES = [  # (class, img_resolution, input side fed, sr_antialias, extra kwargs)
    ('SuperresolutionHybrid8X', 512, 128, True, {}),
    ('SuperresolutionHybrid8X', 512, 64, True, {}),            # resized up to 128 first (antialias has no effect when enlarging)
    ('SuperresolutionHybrid4X', 256, 128, True, {}),
    ('SuperresolutionHybrid4X', 256, 96, True, {}),            # 4X resizes only smaller inputs (:80); a larger one trips block0's shape assert
    ('SuperresolutionHybrid8X', 512, 160, True, {}),           # resized DOWN to 128 with the antialiased kernel
    ('SuperresolutionHybrid8X', 512, 160, False, {}),          # ... and with plain bilinear
    ('SuperresolutionHybrid2X', 128, 64, True, {}),
    ('SuperresolutionHybrid2X_semantic', 128, 48, True, dict(semantic_channels=5)),
    ('SuperresolutionHybrid8XDC', 512, 128, True, {}),
    ('SuperresolutionHybrid8XDC_semantic', 512, 96, False, dict(semantic_channels=6)),
]


def group_srheads():
    """Every super-resolution head class on its own (the model groups only reach 8XDC and 2X), including the input resize in front of it."""
    import dnnlib
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    arrays = {}
    for i, (cls, res, side, aa, extra) in enumerate(SR_CASES):
        torch.manual_seed(0)
        sr = dnnlib.util.construct_class_by_name(class_name='training.superresolution.' + cls, channels=32, img_resolution=res, sr_num_fp16_res=4,
                                                 sr_antialias=aa, channel_base=32768, channel_max=512, fused_modconv_default='inference_only', **extra).eval().requires_grad_(False)
        weights.seed_module(sr, seed=20 + i)
        gz = torch.Generator().manual_seed(60 + i)
        ch = extra.get('semantic_channels', 3)
        x = torch.randn(1, 32, side, side, generator=gz)
        rgb = x[:, :ch].clone()
        ws = torch.randn(1, 14, 512, generator=gz)
        with torch.no_grad():
            y = sr(rgb, x, ws, noise_mode='const')
        assert y.shape == (1, ch, res, res)
        arrays[f'{i}.x_head'], arrays[f'{i}.ws_head'] = x.reshape(-1)[:16].clone(), ws.reshape(-1)[:16].clone()     # inputs are re-drawn from the seed by the tests
        arrays[f'{i}.thumb'], arrays[f'{i}.crop'] = _thumb(y, res // 32)
        print(cls, side, aa, tuple(y.shape))
    save('srheads', **arrays)


GROUPS['srheads'] = group_srheads


ARCH_CASES = dict(
    g_orig=dict(class_name='training.networks_stylegan2.Generator', z_dim=64, c_dim=0, w_dim=64, img_resolution=32, img_channels=3, channel_base=512, channel_max=32,
                architecture='orig', mapping_kwargs=dict(num_layers=2)),
    g_skip=dict(class_name='training.networks_stylegan2.Generator', z_dim=64, c_dim=5, w_dim=64, img_resolution=32, img_channels=3, channel_base=512, channel_max=32,
                architecture='skip', mapping_kwargs=dict(num_layers=3, embed_features=16, layer_features=48)),
    g_resnet=dict(class_name='training.networks_stylegan2.Generator', z_dim=64, c_dim=0, w_dim=64, img_resolution=64, img_channels=2, channel_base=512, channel_max=24,
                  architecture='resnet', mapping_kwargs=dict(num_layers=2), conv_clamp=1.5, use_noise=False),
    d_orig=dict(class_name='training.networks_stylegan2.Discriminator', c_dim=0, img_resolution=32, img_channels=3, architecture='orig', channel_base=512, channel_max=32,
                num_fp16_res=0, conv_clamp=None),
    d_skip=dict(class_name='training.networks_stylegan2.Discriminator', c_dim=5, img_resolution=32, img_channels=3, architecture='skip', channel_base=512, channel_max=32,
                num_fp16_res=0, conv_clamp=None, cmap_dim=12, epilogue_kwargs=dict(mbstd_group_size=None)),
    d_resnet=dict(class_name='training.networks_stylegan2.Discriminator', c_dim=0, img_resolution=64, img_channels=1, architecture='resnet', channel_base=512, channel_max=24,
                  num_fp16_res=0, conv_clamp=0.8, block_kwargs=dict(activation='relu'), epilogue_kwargs=dict(mbstd_num_channels=0)),
)


def group_architectures():
    """The StyleGAN2 generator / discriminator in the three block architectures ('orig', 'skip', 'resnet') and the constructor options pix2pix3D's
    own configurations never flip (conditional mapping with its own feature widths, truncation cutoff, no noise, clamps, relu, no minibatch-std)."""
    import dnnlib
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    arrays = {}
    for name, kw in ARCH_CASES.items():
        torch.manual_seed(0)
        net = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
        weights.seed_module(net, seed=33)
        gz = torch.Generator().manual_seed(71)
        n = 3
        c = torch.randn(n, kw['c_dim'], generator=gz) if kw['c_dim'] > 0 else None
        with torch.no_grad():
            if name.startswith('g_'):
                z = torch.randn(n, kw['z_dim'], generator=gz)
                ws = net.mapping(z, c, truncation_psi=0.7, truncation_cutoff=4)
                img = net.synthesis(ws, noise_mode='const')
                img_none = net.synthesis(ws, noise_mode='none', force_fp32=True)
                arrays.update({f'{name}.z': z, f'{name}.ws': ws, f'{name}.img': img, f'{name}.img_none': img_none})
            else:
                img = torch.randn(n, kw['img_channels'], kw['img_resolution'], kw['img_resolution'], generator=gz)
                arrays.update({f'{name}.img': img, f'{name}.logits': net(img, c)})
        if c is not None:
            arrays[f'{name}.c'] = c
        print(name, 'ok')
    save('architectures', **arrays)


GROUPS['architectures'] = group_architectures


def group_helpers():
    """Small host-side pieces next to the hot path: filtered_resizing (all four modes; loss.py feeds its result to D), sample_from_3dgrid,
    InfiniteSampler's index streams (the rank sharding of the training loop), math_utils."""
    from training.dual_discriminator import filtered_resizing
    from training.volumetric_rendering.renderer import sample_from_3dgrid
    from training.volumetric_rendering import math_utils
    from torch_utils import misc
    from torch_utils.ops import upfirdn2d
    gz = torch.Generator().manual_seed(91)
    a = {}
    img = torch.randn(2, 3, 24, 24, generator=gz)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    a['fr.img'] = img
    for mode in ('antialiased', 'classic', 'none', 0.3):
        a[f'fr.{mode}.up'] = filtered_resizing(img, size=64, f=f, filter_mode=mode)
        if mode != 'classic':               # 'classic' = x2 upsample, resize, /2 decimate: only meaningful upwards
            a[f'fr.{mode}.down'] = filtered_resizing(img, size=10, f=f, filter_mode=mode)
    grid = torch.randn(1, 5, 6, 7, 8, generator=gz)
    coords = torch.rand(3, 40, 3, generator=gz) * 2.4 - 1.2
    a['g3.grid'], a['g3.coords'], a['g3.out'] = grid, coords, sample_from_3dgrid(grid, coords)
    for i, kw in enumerate([dict(rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5), dict(rank=1, num_replicas=3, shuffle=True, seed=7, window_size=0.5),
                            dict(rank=2, num_replicas=4, shuffle=False), dict(rank=0, num_replicas=2, shuffle=True, seed=3, window_size=0)]):
        smp = misc.InfiniteSampler.__new__(misc.InfiniteSampler)     # its __init__ passes the dataset to Sampler.__init__, which torch 2.x no longer takes
        smp.dataset, smp.seed, smp.window_size = list(range(37)), kw.get('seed', 0), kw.get('window_size', 0.5)
        smp.rank, smp.num_replicas, smp.shuffle = kw['rank'], kw['num_replicas'], kw['shuffle']
        it = iter(smp)
        a[f'sampler.{i}'] = np.array([int(next(it)) for _ in range(120)])
    o = torch.randn(2, 50, 3, generator=gz) * 0.3 + torch.tensor([0., 0., 2.5])
    d = torch.nn.functional.normalize(torch.randn(2, 50, 3, generator=gz) * 0.2 - torch.tensor([0., 0., 1.]), dim=-1)
    near, far = math_utils.get_ray_limits_box(o, d, box_side_length=1.3)
    a['mu.o'], a['mu.d'], a['mu.near'], a['mu.far'] = o, d, near, far
    a['mu.linspace'] = math_utils.linspace(near.clamp(-5, 5).nan_to_num(0), far.clamp(-5, 5).nan_to_num(1), 7)
    m = torch.randn(4, 4, generator=gz)
    a['mu.m'], a['mu.tv'] = m, math_utils.transform_vectors(m, torch.randn(50, 4, generator=torch.Generator().manual_seed(92)))
    a['mu.nv'], a['mu.dot'] = math_utils.normalize_vecs(o), math_utils.torch_dot(o, d)
    save('helpers', **a)


GROUPS['helpers'] = group_helpers


# ---------------------------------------------------------------------------------------------------------
def group_loss_phases():
    """The reference's own ``training/loss.py::Pix2Pix3DLoss.accumulate_gradients`` (loss.py:509-1003) on a small G / D / D_semantic triple, every
    phase of training_loop.py:360-373, on the CPU: per-phase gradient norms of every parameter, heads of a few gradients and every statistic the
    loss reports (scores, reconstruction terms, cross-view loss, R1 penalties).  ``lpips`` is loss_phase_driver.py's stand-in; random draws come from det_rng.py.
    Three configurations: the image-pose branch (random_c_prob 0), the random-pose branch (random_c_prob 1: Gmain / Dmain / D_semanticmain), and the
    image-pose branch with the discriminator blur on (blur_sigma 1.5: a 9-tap separable filter on D's input, loss.py:457-466)."""
    import dnnlib
    configs = _load_by_path('p3d_configs', os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pix2pix3d_amd', 'configs.py'))
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    drv = _load_by_path('p3d_loss_phase_driver', os.path.join(HERE, 'loss_phase_driver.py'))
    drv.install_lpips_stub()
    from training import loss as L
    from torch_utils import training_stats
    sink, training_stats.report = drv.make_sink()
    gkw, dkw, dskw = configs.small_train_kwargs()
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**gkw).train().requires_grad_(False)
    D = dnnlib.util.construct_class_by_name(**dkw).train().requires_grad_(False)
    Ds = dnnlib.util.construct_class_by_name(**dskw).train().requires_grad_(False)
    drv.seed_networks(weights, G, D, Ds)
    nets = dict(G=G, D=D, D_semantic=Ds)
    batch, gen_z, gen_c = drv.loss_phase_inputs(configs)
    arrays = dict(gen_z=gen_z, gen_c=gen_c, image=batch['image'], mask=batch['mask'], pose=batch['pose'])
    import time
    for tag, extra, phases, nimg in drv.RUNS:
        loss = L.Pix2Pix3DLoss(device=torch.device('cpu'), G=G, D=D, D_semantic=Ds, augment_pipe=None, **dict(drv.LOSS_KW, **extra))
        t0 = time.time()
        res = drv.run_loss_phases(loss, nets, batch, gen_z, gen_c, sink, phases=phases, cur_nimg=nimg)
        print(tag, f'{time.time() - t0:.1f} s')
        for phase, (names, norms, grads, stats, log) in res.items():
            key = f'{tag}.{phase}'
            arrays.update(drv.phase_record(key, names, norms, grads, stats, log))
            print(' ', key, 'grads', int((norms >= 0).sum()), '/', len(names), 'max norm', float(norms.max()), 'draws', len(log), {k: round(float(np.mean(v)), 5) for k, v in stats.items()})
    save('loss_phases', **arrays)


GROUPS['loss_phases'] = group_loss_phases


def group_discriminator_full():
    """The reference DualDiscriminator of config 3 at its REAL size — 512^2, channel_base 32768, conv_clamp 256, batch 2 — for 3 image channels (D) and
    3 + 6 (D_semantic), in the 'Dboth' phase on real input (loss.py:871-893 / 977-1000, gamma 5): logits, the R1 gradient fields w.r.t. both inputs
    (per-32x32-tile sums and abs-maxima of every channel + norms), the penalty, and the gradient norm of every parameter.  On the CPU the reference
    runs its fp16 blocks in fp32 (networks_stylegan2.py:624-626): this is the fp32 function, clamps included."""
    import dnnlib
    from torch_utils.ops import conv2d_gradfix
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    cases = _load_by_path('p3d_disc_full_cases', os.path.join(HERE, 'disc_full_cases.py'))
    _tile_stats = cases.tile_stats
    arrays = {}
    import time
    for tag, ch in cases.CASES:
        torch.manual_seed(0)
        D = dnnlib.util.construct_class_by_name(**cases.full_discriminator_kwargs(ch)).train().requires_grad_(True)
        weights.seed_discriminator(D, seed=9 + ch)
        img, raw, c = cases.full_discriminator_inputs(ch)
        t0 = time.time()
        x = {'image': img.clone().requires_grad_(True), 'image_raw': raw.clone().requires_grad_(True)}
        logits = D(x, c)
        with conv2d_gradfix.no_weight_gradients():
            g_img, g_raw = torch.autograd.grad(outputs=[logits.sum()], inputs=[x['image'], x['image_raw']], create_graph=True, only_inputs=True)
        r1 = g_img.square().sum([1, 2, 3]) + g_raw.square().sum([1, 2, 3])
        (torch.nn.functional.softplus(-logits) + r1 * (5 / 2)).mean().backward()
        print(tag, f'{time.time() - t0:.1f} s', 'logits', logits.detach().flatten().tolist(), 'r1', r1.detach().tolist())
        a = dict(logits=logits.detach(), r1=r1.detach(), g_img_norm=g_img.detach().double().norm(), g_raw_norm=g_raw.detach().double().norm())
        a['g_img_tile_sum'], a['g_img_tile_max'] = _tile_stats(g_img)
        a['g_raw_tile_sum'], a['g_raw_tile_max'] = _tile_stats(g_raw, 8)
        a['g_img_crop'] = g_img.detach()[:, :, 240:272, 240:272].clone()
        params = dict(D.named_parameters())
        names = [n for n, p in params.items() if p.grad is not None]
        a['grad_names'] = np.array(names)
        a['grad_norms'] = np.array([float(params[n].grad.double().norm()) for n in names])
        for j, nm in enumerate((names[0], names[len(names) // 2], names[-1])):
            a[f'h{j}'] = params[nm].grad.reshape(-1)[:64].clone()
        a['head_names'] = np.array([names[0], names[len(names) // 2], names[-1]])
        arrays.update({f'{tag}.{k}': v for k, v in a.items()})
    save('discriminator_full', **arrays)


GROUPS['discriminator_full'] = group_discriminator_full


ENCODER_KW = dict(img_resolution=64, img_channels=3, architecture='skip', channel_base=1 / 64, channel_max=32, progressive=True, lowres_head=16,
                  model_kwargs=dict(output_mode='W+', num_ws=3, w_dim=8))


def group_encoder_variants():
    """The branches of the reference ``Encoder`` (triplane_cond.py:65-196) that pix2pix3D never configures but that CAN execute: progressive growing with the
    full pyramid (alpha = -1) and entered at the low-resolution head (alpha = 0, image already at 16^2: the first active block gets fromrgb(img) as its
    feature input).  The other progressive paths and predict_camera call `downsample` / `camera_9d_to_16d`, which the reference defines nowhere — recorded
    here as the NameError they raise."""
    from training.triplane_cond import Encoder
    weights = _load_by_path('p3d_weights', os.path.join(HERE, 'weights.py'))
    torch.manual_seed(0)
    enc = Encoder(**ENCODER_KW).eval().requires_grad_(False)
    weights.seed_module(enc, seed=3)
    g = torch.Generator().manual_seed(8)
    full, low = torch.randn(2, 3, 64, 64, generator=g), torch.randn(2, 3, 16, 16, generator=g)
    arrays = dict(full=full, low=low, ws_full=enc(full)['ws'])
    enc.set_alpha(0.0)
    arrays['ws_low'] = enc({'img': low})['ws']
    enc.set_alpha(0.5)                                   # no schedule set: before_res == target_res == the input's size -> no blend, the alpha = 0 path (:140-147)
    arrays['ws_half'] = enc(low)['ws']
    mid = torch.randn(2, 3, 32, 32, generator=g)
    arrays['mid'] = mid
    errs = []

    def blend():
        enc.set_resolution((2, None, 16, 32)); enc.set_alpha(0.25)      # a real blend between the 16^2 head and the 32^2 block
        return enc(mid)
    for make in (lambda: (enc.set_alpha(0.0), enc(full)), blend):
        try:
            make(); errs.append('none')
        except NameError as e:
            errs.append(str(e))
    cam = Encoder(**dict(ENCODER_KW, progressive=False, lowres_head=None, model_kwargs=dict(output_mode='W+', num_ws=3, w_dim=8, predict_camera=True))).eval()
    try:
        cam(full); errs.append('none')
    except NameError as e:
        errs.append(str(e))
    arrays['name_errors'] = np.array(errs)
    arrays['camera_out_dim'] = np.int64(cam.out_dim)
    print(errs)
    save('encoder_variants', **arrays)


GROUPS['encoder_variants'] = group_encoder_variants

if __name__ == '__main__':
    names = sys.argv[1:] or list(GROUPS)
    for nm in names:
        GROUPS[nm]()
