"""Model level, CPU: the functional oracle and this package's generator (CPU-tensor path) against outputs recorded
from the reference generator with the same name-seeded weights, latent codes, cameras and uniforms."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from model_cases import build_generator, uniforms, replay_uniforms, compare_outputs


@pytest.mark.parametrize('name', ['seg2cat', 'edge2car'])
def test_generator_cpu_path_matches_reference(name):
    from pix2pix3d_amd import configs
    g = load_golden('model_' + name)
    G = build_generator(name)
    rk = G.rendering_kwargs
    ws, c, nrr = torch.tensor(g['ws']), torch.tensor(g['c']), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, rk)
    with replay_uniforms(u_c, u_f), torch.no_grad():
        out = G.synthesis(ws, c, neural_rendering_resolution=nrr, noise_mode='const')
    compare_outputs(out, g, tol_raw=2e-5, tol_sr=5e-5)
    with torch.no_grad():
        sm = G.sample_mixed(torch.tensor(g['pts']), None, ws, noise_mode='const')
    assert rel_err(sm['rgb'].numpy(), g['pts_rgb']) < 2e-5 and rel_err(sm['sigma'].numpy(), g['pts_sigma']) < 2e-5
    # parameter names are the reference's: the goldens were produced by seeding the REFERENCE by name
    names = {n for n, _ in G.named_parameters()}
    assert 'backbone.synthesis.b256.conv1.affine.weight' in names and 'superresolution_semantic.block1.torgb.weight' in names
    assert 'decoder.net_semantic.2.bias' in names and 'backbone.mapping.embed_mask.projector.weight' in names


@pytest.mark.parametrize('name', ['seg2cat', 'edge2car'])
def test_mapping_network_matches_reference(name):
    g = load_golden('model_' + name)
    G = build_generator(name)
    c = torch.tensor(g['c'])
    mask = torch.tensor(g['map_mask'].astype(np.int64) if name == 'seg2cat' else g['map_mask'])
    with torch.no_grad():
        ws = G.mapping(torch.tensor(g['map_z']), c, {'mask': mask, 'pose': c})
    assert ws.shape == g['map_ws'].shape and rel_err(ws.numpy(), g['map_ws']) < 2e-5


@pytest.mark.parametrize('name', ['seg2cat', 'edge2car'])
def test_model_oracle_matches_reference(name):
    from oracle import model_oracle as M
    from pix2pix3d_amd import configs
    g = load_golden('model_' + name)
    G = build_generator(name)
    sd = {k: v.float() for k, v in G.state_dict().items()}
    ws, c, nrr = torch.tensor(g['ws']), torch.tensor(g['c']), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)
    with torch.no_grad():
        out = M.synthesis(sd, configs.oracle_cfg(name), ws, c, u_c, u_f, nrr=nrr, noise_mode='const')
    compare_outputs(out, g, tol_raw=1e-4, tol_sr=1e-4)
