"""The StyleGAN2 generator / discriminator in the three block architectures and with the constructor options pix2pix3D's own configurations
never flip, against records from the reference (tests/golden/make_golden.py group ``architectures``)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from model_cases import weights


def _cases():
    src = open(os.path.join(GOLDEN, 'make_golden.py')).read()
    block = src[src.index('ARCH_CASES = dict('):]
    block = block[:block.index('\n)\n') + 3]
    ns = {}
    exec(block, ns)
    return ns['ARCH_CASES']


ARCH_CASES = _cases()


def _run(name, device, tol, tol_fp16=None):
    from pix2pix3d_amd import dnnlib
    kw = ARCH_CASES[name]
    g = {k.split('.', 1)[1]: v for k, v in load_golden('architectures').items() if k.startswith(name + '.')}
    torch.manual_seed(0)
    net = dnnlib.util.construct_class_by_name(**kw).eval().requires_grad_(False)
    weights.seed_module(net, seed=33)
    net = net.to(device)
    c = torch.tensor(g['c']).to(device) if 'c' in g else None
    with torch.no_grad():
        if name.startswith('g_'):
            ws = net.mapping(torch.tensor(g['z']).to(device), c, truncation_psi=0.7, truncation_cutoff=4)
            assert rel_err(ws.cpu().numpy(), g['ws']) < tol
            ws = torch.tensor(g['ws']).to(device)
            # on a device the top num_fp16_res (default 4) resolutions run in fp16 unless forced (networks_stylegan2.py:423-425)
            assert rel_err(net.synthesis(ws, noise_mode='const').float().cpu().numpy(), g['img']) < (tol_fp16 or tol)
            assert rel_err(net.synthesis(ws, noise_mode='const', force_fp32=True).float().cpu().numpy(), g['img']) < tol
            assert rel_err(net.synthesis(ws, noise_mode='none', force_fp32=True).float().cpu().numpy(), g['img_none']) < tol
        else:
            assert rel_err(net(torch.tensor(g['img']).to(device), c).cpu().numpy(), g['logits']) < tol


@pytest.mark.parametrize('name', list(ARCH_CASES))
def test_architecture_cpu_path_matches_reference(name):
    _run(name, 'cpu', 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(ARCH_CASES))
def test_architecture_device_path_matches_reference(name):
    _run(name, 'cuda', 1e-3, tol_fp16=3e-2)
