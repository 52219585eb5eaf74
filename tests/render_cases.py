"""Helpers shared by the renderer tests: load a golden renderer case, rebuild options / decoder."""
import numpy as np

from conftest import load_golden

CASES = ['seg', 'car', 'osg', 'auto']


def _parse(v):
    if v in ('True', 'False'):
        return v == 'True'
    if v == 'auto' or v == 'softplus':
        return v
    f = float(v)
    return int(f) if f.is_integer() and '.' not in v else f


def load_case(name):
    g = load_golden('renderer_' + name)
    opts = {k: _parse(v) for k, v in zip(g['opt_keys'].tolist(), g['opt_vals'].tolist())}
    dec = {k[4:]: g[k] for k in g.files if k.startswith('dec_')}
    dec['lr_mul'] = float(g['lr_mul'])
    dec['semantic_sigmoid'] = bool(g['sem_sigmoid'])
    return g, opts, dec


def make_decoder(g, device='cpu'):
    """Instantiate this package's decoder module with the golden weights."""
    import torch
    from pix2pix3d_amd.training.triplane import OSGDecoder
    from pix2pix3d_amd.training.triplane_cond import OSGDecoder_semantic_lateSeparate
    lr = float(g['lr_mul'])
    if int(g['nets']) == 2:
        dec = OSGDecoder_semantic_lateSeparate(32, {'decoder_lr_mul': lr, 'decoder_output_dim': 32, 'sigmoid': bool(g['sem_sigmoid']), 'semantic_channels': 6})
        pairs = [(dec.net, ''), (dec.net_semantic, 's')]
    else:
        dec = OSGDecoder(32, {'decoder_lr_mul': lr, 'decoder_output_dim': 32})
        pairs = [(dec.net, '')]
    with torch.no_grad():
        for net, sfx in pairs:
            net[0].weight.copy_(torch.tensor(g['dec_w1' + sfx])); net[0].bias.copy_(torch.tensor(g['dec_b1' + sfx]))
            net[2].weight.copy_(torch.tensor(g['dec_w2' + sfx])); net[2].bias.copy_(torch.tensor(g['dec_b2' + sfx]))
    return dec.to(device).requires_grad_(False)
