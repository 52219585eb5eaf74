"""Name-seeded synthetic weights: the same values for the same parameter NAME and SHAPE no matter which
implementation (reference, this package, oracle) owns the tensor or in which order tensors were created.
Used because no checkpoint is available offline.  Standalone (torch + zlib only) so the reference-side golden
generator can import it by path."""
import zlib

import torch


def _scale_and_shift(name, shape):
    leaf = name.split('.')[-1]
    if leaf in ('resample_filter', 'alpha'):
        return None                                              # structural buffers: keep the constructor's value
    if leaf == 'noise_strength':
        return 0.1, 0.0
    if leaf == 'bias':
        return (0.1, 1.0) if '.affine.' in name else (0.1, 0.0)  # style affines are initialised around 1
    if leaf == 'w_avg':
        return 0.1, 0.0
    if leaf == 'weight' and '.mapping.fc' in name:
        return 100.0, 0.0                                        # lr_multiplier 0.01 layers store weights / 0.01
    return 1.0, 0.0


def seeded_tensor(name, shape, seed=0):
    ss = _scale_and_shift(name, shape)
    if ss is None:
        return None
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
    return torch.randn(list(shape), generator=g, dtype=torch.float32) * ss[0] + ss[1]


def seed_module(module, seed=0):
    """Overwrite every floating-point parameter / buffer of ``module`` in place; returns the state dict."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point():
                continue
            v = seeded_tensor(name, t.shape, seed)
            if v is not None:
                t.copy_(v.to(t.device, t.dtype))
    return {k: v.detach() for k, v in module.state_dict().items()}


def seed_discriminator(module, seed=0):
    """seed_module for a discriminator, with its label-mapping network's weights at the scale their lr_multiplier of 0.01 implies (stored = effective /
    0.01; the '.mapping.fc' rule above only matches names with a prefix, i.e. the generator's): logits and R1 penalties come out O(1) instead of 1e-3 /
    1e-9.  (The round-2 golden 'discriminator' predates this and keeps plain seed_module.)"""
    sd = seed_module(module, seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.startswith('mapping.fc') and name.endswith('.weight'):
                p.mul_(100.0)
    return sd
