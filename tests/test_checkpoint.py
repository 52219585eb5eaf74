"""Checkpoint wire format: a checkpoint written by the REFERENCE (tests/golden/make_golden.py group
``checkpoint``: reference persistence + pickle, read back by the reference's legacy.load_network_pkl to record outputs) loads through
``pix2pix3d_amd.legacy`` into this package's modules and computes what the reference computed; writer round trip; the unpickler's
allow-list."""
import io
import lzma
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from model_cases import replay_uniforms


@pytest.fixture(scope='module')
def ckpt():
    from pix2pix3d_amd import legacy
    with lzma.open(os.path.join(GOLDEN, 'checkpoint_small.pkl.xz'), 'rb') as f:
        return legacy.load_network_pkl(f)


def _uniforms(seed, n, nrr, rk):
    torch.manual_seed(seed)
    m = nrr * nrr
    return torch.rand([n, m, rk['depth_resolution'], 1]), torch.rand([n * m, rk['depth_resolution_importance']])


def test_reference_checkpoint_resolves_to_this_packages_classes(ckpt):
    from pix2pix3d_amd.torch_utils import persistence
    from pix2pix3d_amd.training import triplane, triplane_cond, dual_discriminator
    g = load_golden('checkpoint_small')
    assert type(ckpt['G_ema']) is triplane_cond.TriPlaneSemanticEntangleGenerator
    assert type(ckpt['D']) is dual_discriminator.DualDiscriminator
    assert type(ckpt['eg3d']) is triplane.TriPlaneGenerator                 # same class NAME as triplane_cond's: told apart by module source
    assert ckpt['G'].training and all(p.requires_grad for p in ckpt['G'].parameters())      # flags as pickled
    assert not ckpt['eg3d'].training and not any(p.requires_grad for p in ckpt['eg3d'].parameters())
    assert ckpt['G_ema'].neural_rendering_resolution == 24                  # plain attribute assigned after construction
    assert ckpt['training_set_kwargs'].resolution == 128 and ckpt['training_set_kwargs']['use_labels'] is True
    # a class this package has no mirror of comes back as a parameter holder, buffers intact, forward refusing
    aug = ckpt['augment_pipe']
    assert isinstance(aug, persistence.PickledModule) and type(aug).__name__ == 'AugmentPipe'
    assert sorted(k for k, _ in aug.named_buffers()) == list(g['aug_buffers'])
    with pytest.raises(NotImplementedError):
        aug(torch.zeros(1, 3, 8, 8))
    # every tensor of the checkpoint arrived, bit for bit
    sd = ckpt['G_ema'].state_dict()
    assert sorted(sd) == list(g['param_names'])
    np.testing.assert_array_equal(np.array([float(sd[k].double().sum()) for k in sorted(sd)]), g['param_sums'])
    np.testing.assert_array_equal(np.array([float(v.double().sum()) for _, v in sorted(ckpt['D'].state_dict().items())]), g['d_param_sums'])
    # modules are freshly constructed instances (attributes this package's constructors add are present)
    assert hasattr(ckpt['G_ema'].backbone.synthesis.b8, '_in_div')


def test_reference_checkpoint_computes_the_reference_outputs_cpu(ckpt):
    g = load_golden('checkpoint_small')
    G, D, E = ckpt['G_ema'], ckpt['D'], ckpt['eg3d']
    was_training = G.training
    G.eval(); D.eval()
    try:
        c, z, mask = torch.tensor(g['c']), torch.tensor(g['z']), torch.tensor(g['mask'])
        with torch.no_grad():
            ws = G.mapping(z, c, {'mask': mask, 'pose': c})
            assert rel_err(ws.numpy(), g['ws']) < 1e-4
            ws = torch.tensor(g['ws'])
            with replay_uniforms(*_uniforms(int(g['render_seed']), 2, 16, G.rendering_kwargs)):
                out = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const')
            for k in ('image', 'image_raw', 'image_depth', 'semantic', 'semantic_raw'):
                assert rel_err(out[k].numpy(), g[k]) < 2e-4, k
            logits = D({'image': torch.tensor(g['image']), 'image_raw': torch.tensor(g['image_raw'])}, c)
            assert rel_err(logits.numpy(), g['logits']) < 1e-4
            ws_e = E.mapping(z, c)
            assert rel_err(ws_e.numpy(), g['eg3d_ws']) < 1e-4
            with replay_uniforms(*_uniforms(int(g['render_seed']), 2, 16, E.rendering_kwargs)):
                out_e = E.synthesis(torch.tensor(g['eg3d_ws']), c, neural_rendering_resolution=16, noise_mode='const')
            for k in ('image', 'image_raw', 'image_depth'):
                assert rel_err(out_e[k].numpy(), g['eg3d_' + k]) < 2e-4, k
    finally:
        G.train(was_training); D.train(True)


def test_writer_round_trip_and_force_fp16(ckpt):
    from pix2pix3d_amd import legacy
    E = ckpt['eg3d']
    buf = io.BytesIO()
    legacy.save_network_pkl(dict(G=E, D=ckpt['D'], G_ema=E, augment_pipe=None), buf)
    # the record on the wire is the reference's: REDUCE(persistence._reconstruct_persistent_obj, (meta,)) with these meta keys
    import pickletools
    need = {'type', 'version', 'module_src', 'class_name', 'state', '_reconstruct_persistent_obj'}
    for op, arg, _ in pickletools.genops(buf.getvalue()):
        if isinstance(arg, str):
            need -= {arg, arg.split(' ')[-1]}
        if not need:
            break
    assert not need, need
    back = legacy.load_network_pkl(io.BytesIO(buf.getvalue()))
    assert back['G'] is back['G_ema'] and type(back['G']) is type(E)
    for (k1, v1), (k2, v2) in zip(sorted(E.state_dict().items()), sorted(back['G'].state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2)
    assert back['G'].init_kwargs == E.init_kwargs and back['G'].rendering_kwargs == E.rendering_kwargs
    # force_fp16 (legacy.py:49-60): rebuilt with num_fp16_res=4 / conv_clamp=256, same tensors
    forced = legacy.load_network_pkl(io.BytesIO(buf.getvalue()), force_fp16=True)
    assert forced['G'].init_kwargs['num_fp16_res'] == 4 and forced['G'].init_kwargs['conv_clamp'] == 256
    assert forced['G'].backbone.synthesis.b256.use_fp16 and not E.backbone.synthesis.b256.use_fp16
    assert torch.equal(forced['G'].backbone.synthesis.b64.conv1.weight, E.backbone.synthesis.b64.conv1.weight)


class _Evil:
    def __reduce__(self):
        return os.system, ('echo pwned',)


def test_unpickler_refuses_anything_a_checkpoint_has_no_use_for():
    from pix2pix3d_amd import legacy
    with pytest.raises(pickle.UnpicklingError, match='refusing'):
        legacy.load_network_pkl(io.BytesIO(pickle.dumps(dict(G=_Evil()))))
    blob = pickle.dumps(dict(G=torch.nn.Identity()))
    for name in (b'posix\nsystem', b'builtins\neval', b'torch\nload'):
        forged = pickle.dumps(dict(G=None), protocol=0).replace(b'N', b'c' + name + b'\n(S\'1\'\ntR', 1)
        with pytest.raises(pickle.UnpicklingError):
            legacy.load_network_pkl(io.BytesIO(forged))
    with pytest.raises((AssertionError, KeyError)):
        legacy.load_network_pkl(io.BytesIO(blob))                     # loads, but is not a network checkpoint (no D / G_ema)


def test_pickled_source_is_never_executed(tmp_path):
    """A record whose module_src would create a file if exec'd, as the reference's reader does (persistence.py:207-218)."""
    from pix2pix3d_amd import legacy
    from pix2pix3d_amd.torch_utils import persistence
    from pix2pix3d_amd.training.networks_stylegan2 import FullyConnectedLayer
    marker = tmp_path / 'executed'
    fc = FullyConnectedLayer(4, 3)
    _, (meta,), _ = fc.__reduce__()
    meta = dict(meta, module_src=f"open({str(marker)!r}, 'w').write('x')\n" + meta['module_src'], writer=None, module_name=None)

    class _Rec:
        def __reduce__(self):
            return persistence._reconstruct_persistent_obj, (meta,)
    data = legacy.load_network_pkl(io.BytesIO(pickle.dumps(dict(G=_Rec(), D=torch.nn.Identity(), G_ema=torch.nn.Identity()))))
    assert type(data['G']) is FullyConnectedLayer and torch.equal(data['G'].weight, fc.weight) and not marker.exists()


@pytest.mark.gpu
def test_reference_checkpoint_runs_on_the_hip_path(ckpt):
    """applications/generate_samples.py:77: ``legacy.load_network_pkl(f)['G_ema'].eval().to(device)`` then G.synthesis — here against the
    outputs the reference computed from the same file (1e-3 with fp32 forced, 3e-2 with the fp16 SR heads)."""
    import copy
    g = load_golden('checkpoint_small')
    G = copy.deepcopy(ckpt['G_ema']).eval().requires_grad_(False).to('cuda')
    E = copy.deepcopy(ckpt['eg3d']).to('cuda')
    c, ws = torch.tensor(g['c']).cuda(), torch.tensor(g['ws']).cuda()
    with torch.no_grad():
        wsm = G.mapping(torch.tensor(g['z']).cuda(), c, {'mask': torch.tensor(g['mask']).cuda(), 'pose': c})
        assert rel_err(wsm.cpu().numpy(), g['ws']) < 1e-3
        for force_fp32, tol in ((True, 1e-3), (False, 3e-2)):
            with replay_uniforms(*_uniforms(int(g['render_seed']), 2, 16, G.rendering_kwargs)):
                out = G.synthesis(ws, c, neural_rendering_resolution=16, noise_mode='const', force_fp32=force_fp32)
            for k in ('image_raw', 'image_depth', 'semantic_raw', 'image', 'semantic'):
                assert rel_err(out[k].float().cpu().numpy(), g[k]) < tol, (k, force_fp32)
            with replay_uniforms(*_uniforms(int(g['render_seed']), 2, 16, E.rendering_kwargs)):
                out_e = E.synthesis(torch.tensor(g['eg3d_ws']).cuda(), c, neural_rendering_resolution=16, noise_mode='const', force_fp32=force_fp32)
            for k in ('image_raw', 'image_depth', 'image'):
                assert rel_err(out_e[k].float().cpu().numpy(), g['eg3d_' + k]) < tol, (k, force_fp32)


def test_import_hook_and_cli_resave(tmp_path):
    """persistence.import_hook (persistence.py:151-185) sees every record before its class is resolved; the ``legacy`` CLI re-saves a checkpoint."""
    from pix2pix3d_amd import legacy
    from pix2pix3d_amd.torch_utils import persistence
    from pix2pix3d_amd.training.networks_stylegan2 import FullyConnectedLayer, Conv2dLayer
    seen = []

    def hook(meta):
        seen.append(meta.class_name)
        if meta.class_name == 'FullyConnectedLayer':
            meta.state['activation'] = 'relu'                 # patch the pickled state, the documented use of the hook
        return meta
    persistence.import_hook(hook)
    try:
        fc, conv = FullyConnectedLayer(5, 4, activation='lrelu'), Conv2dLayer(3, 4, kernel_size=3)
        assert persistence.is_persistent(fc) and persistence.is_persistent(FullyConnectedLayer) and not persistence.is_persistent(torch.nn.Identity())
        src = tmp_path / 'in.pkl'
        with open(src, 'wb') as f:
            legacy.save_network_pkl(dict(G=fc, D=conv, G_ema=fc), f)
        with open(src, 'rb') as f:
            data = legacy.load_network_pkl(f)
        assert seen.count('FullyConnectedLayer') == 1 and 'Conv2dLayer' in seen          # G and G_ema are one object on the wire
        assert data['G'].activation == 'relu' and torch.equal(data['G'].weight, fc.weight) and torch.equal(data['D'].weight, conv.weight)
    finally:
        persistence._import_hooks.remove(hook)
    dst = tmp_path / 'out.pkl'
    legacy.main(['--source', str(src), '--dest', str(dst)])
    with open(dst, 'rb') as f:
        again = legacy.load_network_pkl(f)
    assert again['G'].activation == 'lrelu' and torch.equal(again['G'].weight, fc.weight)


@pytest.mark.parametrize('protocol', [2, 3, 4, 5])
def test_everything_a_snapshot_legitimately_holds_loads_in_every_pickle_protocol(protocol):
    """numpy arrays / scalars, EasyDict, tuples, sets, OrderedDict, dtypes, sizes, slices, devices, fp16 / integer tensors, bare Parameters,
    stock torch.nn modules (protocol 2 spells builtins / copyreg / bytes the Python-2 way)."""
    import collections
    from pix2pix3d_amd import legacy, dnnlib
    d = dict(G=torch.nn.Identity(), D=torch.nn.Identity(), G_ema=torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Softplus()),
             training_set_kwargs=dnnlib.EasyDict(a=np.arange(3), b=np.float32(2.5), c=(1, 2), d={1, 2}, e=collections.OrderedDict(x=1), f=torch.float16,
                                                 g=torch.Size([1, 2]), h=slice(1, 2), i=torch.device('cpu')),
             extra=torch.nn.Parameter(torch.ones(2)), t16=torch.ones(2, dtype=torch.float16), ti=torch.arange(3))
    out = legacy.load_network_pkl(io.BytesIO(pickle.dumps(d, protocol=protocol)))
    k = out['training_set_kwargs']
    assert isinstance(k, dnnlib.EasyDict) and np.array_equal(k.a, np.arange(3)) and k.b == np.float32(2.5) and k.c == (1, 2) and k.d == {1, 2}
    assert k.f is torch.float16 and k.g == torch.Size([1, 2]) and k.h == slice(1, 2) and k.i == torch.device('cpu') and list(k.e.items()) == [('x', 1)]
    assert out['t16'].dtype == torch.float16 and torch.equal(out['ti'], torch.arange(3)) and isinstance(out['extra'], torch.nn.Parameter)
    assert isinstance(out['G_ema'][1], torch.nn.Softplus) and out['G_ema'][0].weight.shape == (2, 2)
