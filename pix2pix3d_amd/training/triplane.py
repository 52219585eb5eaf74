"""EG3D tri-plane decoder (mirror of training/triplane.py:112-135).  The unconditional ``TriPlaneGenerator``
of that file is not on the pix2pix3D path (train.py:374-380 selects the conditional generators of
``triplane_cond``); the decoder is, via ``triplane_cond.TriPlaneGenerator``."""
import torch

from .networks_stylegan2 import FullyConnectedLayer


def _osg_mlp(n_features, hidden, out_dim, lr_mul):
    return torch.nn.Sequential(FullyConnectedLayer(n_features, hidden, lr_multiplier=lr_mul), torch.nn.Softplus(),
                               FullyConnectedLayer(hidden, out_dim, lr_multiplier=lr_mul))


class OSGDecoder(torch.nn.Module):
    """mean over planes -> FC(32,64) -> softplus -> FC(64, 1+C): density = channel 0, colour = clamped sigmoid of the rest."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = _osg_mlp(n_features, self.hidden_dim, 1 + options['decoder_output_dim'], options['decoder_lr_mul'])

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)                      # [N, M, C]
        n, m, c = x.shape
        y = self.net(x.reshape(n * m, c)).reshape(n, m, -1)
        rgb = torch.sigmoid(y[..., 1:]) * (1 + 2 * 0.001) - 0.001      # MipNeRF-style sigmoid clamping
        return {'rgb': rgb, 'sigma': y[..., 0:1]}
