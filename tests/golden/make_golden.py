"""Generate golden input/output vectors by running the REFERENCE implementation on CPU.

Run in the authoring container only (it needs the read-only checkout at /root/reference, which does
not exist on the GPU box):

    python tests/golden/make_golden.py [group ...]      # groups: ops renderer model

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

if __name__ == '__main__':
    names = sys.argv[1:] or list(GROUPS)
    for nm in names:
        GROUPS[nm]()
