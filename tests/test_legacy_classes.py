"""Two reference classes no shipped configuration reaches but old checkpoints can name — OSGDecoder_semantic_entangle (triplane_cond.py:891-924)
and SuperresolutionHybridDeepfp32 (superresolution.py:160-188) — against records from the reference (make_golden.py group ``legacy_classes``)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from model_cases import weights


def _inputs():
    gz = torch.Generator().manual_seed(91)
    feats = torch.randn(2, 3, 50, 32, generator=gz)
    sr_in = {}
    for tag, side in (('same', 128), ('small', 96)):
        x = torch.randn(1, 32, side, side, generator=gz)
        sr_in[tag] = (x, torch.randn(1, 14, 512, generator=gz))
    return feats, sr_in


@pytest.mark.parametrize('device', ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)])
def test_entangled_decoder_matches_reference(device):
    from pix2pix3d_amd.training.triplane_cond import OSGDecoder_semantic_entangle
    g = load_golden('legacy_classes')
    feats, _ = _inputs()
    assert np.array_equal(feats.reshape(-1)[:16].numpy(), g['dec.feats_head'])
    for tag, sig in (('raw', False), ('sigmoid', True)):
        dec = OSGDecoder_semantic_entangle(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32, 'sigmoid': sig, 'semantic_channels': 6}).requires_grad_(False)
        weights.seed_module(dec, seed=71)
        out = dec.to(device)(feats.to(device), None)
        assert rel_err(out['rgb'].cpu().numpy(), g[f'dec.{tag}.rgb']) < 2e-5 and rel_err(out['sigma'].cpu().numpy(), g[f'dec.{tag}.sigma']) < 2e-5


def _sr(device, tol, **kw):
    from pix2pix3d_amd import dnnlib
    g = load_golden('legacy_classes')
    _, sr_in = _inputs()
    torch.manual_seed(0)
    sr = dnnlib.util.construct_class_by_name(class_name='training.superresolution.SuperresolutionHybridDeepfp32', channels=32, img_resolution=256, sr_num_fp16_res=4,
                                             channel_base=32768, channel_max=512, fused_modconv_default='inference_only').eval().requires_grad_(False)
    weights.seed_module(sr, seed=72)
    sr = sr.to(device)
    for tag, (x, ws) in sr_in.items():
        assert np.array_equal(x.reshape(-1)[:16].numpy(), g[f'sr.{tag}.x_head']) and np.array_equal(ws.reshape(-1)[:16].numpy(), g[f'sr.{tag}.ws_head'])
        with torch.no_grad():
            y = sr(x[:, :3].clone().to(device), x.to(device), ws.to(device), noise_mode='const', **kw).float().cpu()
        assert y.shape == (1, 3, 256, 256)
        c0 = 256 // 2 - 16
        scale = np.abs(g[f'sr.{tag}.thumb']).max()
        err = max(np.abs(y[..., ::8, ::8].numpy() - g[f'sr.{tag}.thumb']).max(), np.abs(y[..., c0:c0 + 32, c0:c0 + 32].numpy() - g[f'sr.{tag}.crop']).max()) / scale
        assert err < tol, (tag, err)


def test_deepfp32_head_cpu_matches_reference():
    _sr('cpu', 5e-5)


@pytest.mark.gpu
def test_deepfp32_head_device_matches_reference(hip_lib):
    _sr('cuda', 1e-3, force_fp32=True)
    _sr('cuda', 3e-2)


def test_encoder_progressive_branches_that_the_reference_can_execute():
    """Encoder (triplane_cond.py:65-196) beyond the configuration pix2pix3D instantiates: progressive growing with the full pyramid and entered at the
    low-resolution head match the reference's records; where the reference itself raises NameError (its `downsample` / `camera_9d_to_16d` are defined
    nowhere) this package raises NotImplementedError that says so."""
    import importlib.util, os
    import pytest
    from conftest import ROOT
    from pix2pix3d_amd.training.triplane_cond import Encoder
    spec = importlib.util.spec_from_file_location('p3d_weights', os.path.join(ROOT, 'tests', 'golden', 'weights.py'))
    weights = importlib.util.module_from_spec(spec); spec.loader.exec_module(weights)
    g = load_golden('encoder_variants')
    kw = dict(img_resolution=64, img_channels=3, architecture='skip', channel_base=1 / 64, channel_max=32, progressive=True, lowres_head=16,
              model_kwargs=dict(output_mode='W+', num_ws=3, w_dim=8))
    torch.manual_seed(0)
    enc = Encoder(**kw).eval().requires_grad_(False)
    weights.seed_module(enc, seed=3)
    full, low = torch.tensor(g['full']), torch.tensor(g['low'])
    assert rel_err(enc(full)['ws'].numpy(), g['ws_full']) < 1e-5
    enc.set_alpha(0.0)
    assert rel_err(enc({'img': low})['ws'].numpy(), g['ws_low']) < 1e-5
    assert list(g['name_errors']) == ["name 'downsample' is not defined", "name 'downsample' is not defined", "name 'camera_9d_to_16d' is not defined"]
    with pytest.raises(NotImplementedError, match='downsample'):
        enc(full)                                                    # alpha = 0 with a full-size image: would have to be down-sized
    enc.set_alpha(0.5)                                               # no schedule: no blend, the alpha = 0 path
    assert rel_err(enc(low)['ws'].numpy(), g['ws_half']) < 1e-5
    enc.set_resolution((2, None, 16, 32)); enc.set_alpha(0.25)
    with pytest.raises(NotImplementedError, match='downsample'):
        enc(torch.tensor(g['mid']))                                  # a real blend between the 16^2 head and the 32^2 block
    cam = Encoder(**dict(kw, progressive=False, lowres_head=None, model_kwargs=dict(output_mode='W+', num_ws=3, w_dim=8, predict_camera=True)))
    assert cam.out_dim == int(g['camera_out_dim']) == 8 * 3 + 9
    with pytest.raises(NotImplementedError, match='camera_9d_to_16d'):
        cam(full)
