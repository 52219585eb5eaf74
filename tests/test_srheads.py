"""Every super-resolution head class on its own, incl. the input resize in front of it, against records from the reference
(tests/golden/make_golden.py group ``srheads``; inputs are re-drawn from the recorded seeds)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, record_error
from model_cases import weights

_spec = importlib.util.spec_from_file_location('p3d_make_golden_cases', os.path.join(GOLDEN, 'make_golden.py'))


def _cases():
    # the case table lives next to the generator; read it without importing the module (which would put /root/reference on sys.path)
    src = open(os.path.join(GOLDEN, 'make_golden.py')).read()
    block = src[src.index('SR_CASES = ['):]
    block = block[:block.index('\n]\n') + 3]
    ns = {}
    exec(block, ns)
    return ns['SR_CASES']


SR_CASES = _cases()


def _run(i, device, tol, **kw):
    from pix2pix3d_amd import dnnlib
    cls, res, side, aa, extra = SR_CASES[i]
    g = load_golden('srheads')
    torch.manual_seed(0)
    sr = dnnlib.util.construct_class_by_name(class_name='training.superresolution.' + cls, channels=32, img_resolution=res, sr_num_fp16_res=4,
                                             sr_antialias=aa, channel_base=32768, channel_max=512, fused_modconv_default='inference_only', **extra).eval().requires_grad_(False)
    weights.seed_module(sr, seed=20 + i)
    gz = torch.Generator().manual_seed(60 + i)
    ch = extra.get('semantic_channels', 3)
    x = torch.randn(1, 32, side, side, generator=gz)
    rgb = x[:, :ch].clone()
    ws = torch.randn(1, 14, 512, generator=gz)
    assert np.array_equal(x.reshape(-1)[:16].numpy(), g[f'{i}.x_head']) and np.array_equal(ws.reshape(-1)[:16].numpy(), g[f'{i}.ws_head'])
    sr = sr.to(device)
    with torch.no_grad():
        y = sr(rgb.to(device), x.to(device), ws.to(device), noise_mode='const', **kw).float().cpu()
    assert y.shape == (1, ch, res, res)
    step, c0 = res // 32, res // 2 - 16
    scale = np.abs(g[f'{i}.thumb']).max()
    err = max(np.abs(y[..., ::step, ::step].numpy() - g[f'{i}.thumb']).max(), np.abs(y[..., c0:c0 + 32, c0:c0 + 32].numpy() - g[f'{i}.crop']).max()) / scale
    if device != 'cpu':
        record_error(f'srheads.{i}.{cls}.{side}.' + ('fp32' if kw.get('force_fp32') else 'fp16'), float(err))
    assert err < tol, (cls, side, aa, err)


@pytest.mark.parametrize('i', range(len(SR_CASES)))
def test_sr_head_cpu_path_matches_reference(i):
    _run(i, 'cpu', 5e-5)


def test_4x_head_refuses_larger_inputs_like_the_reference():
    from pix2pix3d_amd.training.superresolution import SuperresolutionHybrid4X
    sr = SuperresolutionHybrid4X(channels=32, img_resolution=256, sr_num_fp16_res=0, sr_antialias=True, channel_base=2048, channel_max=16).eval()
    x = torch.randn(1, 32, 160, 160)
    with pytest.raises(AssertionError):                  # superresolution.py:80 only resizes smaller inputs; block0 then asserts the shape (:261)
        sr(x[:, :3], x, torch.randn(1, 14, 512), noise_mode='const')


@pytest.mark.gpu
@pytest.mark.parametrize('i', range(len(SR_CASES)))
def test_sr_head_device_path_matches_reference(i):
    _run(i, 'cuda', 1e-4, force_fp32=True)         # measured <= 1.2e-5 (profiles/round5_a_parity_errors.json)
    _run(i, 'cuda', 3e-3)                          # fp16 blocks: measured <= 8.6e-4 of the range; 3 x that
