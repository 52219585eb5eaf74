"""Loading network checkpoints (mirror of legacy.py:24-60 ``load_network_pkl``; line refs are to that file).

A pix2pix3D / EG3D checkpoint is a pickle of ``dict(G=..., D=..., G_ema=..., [D_semantic=...], training_set_kwargs=...,
augment_pipe=...)`` whose networks are persistent objects (``torch_utils/persistence.py``).  The reference unpickles them by
executing the model source embedded in the file.  Here the file is read by a *restricted* unpickler — only the tensor / container /
numpy / EasyDict constructors a checkpoint legitimately names are resolvable, storages are decoded with
``torch.load(weights_only=True)``, and embedded source is never executed — the persistent records are mapped onto this package's
classes by ``persistence._reconstruct_persistent_obj``, and every top-level network is re-instantiated from its recorded
constructor arguments with the checkpoint's parameters and buffers (``persistence.rebuild``), so what comes back runs on the HIP
kernels.  TensorFlow-era StyleGAN2 pickles (:62-325, ``convert_tf_generator``) hold none of the tri-plane networks and are refused.
"""
import copy
import io
import pickle

import numpy as np
import torch

from . import dnnlib
from .torch_utils import misc, persistence


def load_network_pkl(f, force_fp16=False):
    """``f``: binary file object.  Returns the checkpoint dict with ``G`` / ``D`` / ``G_ema`` (and any other persistent network)
    as modules of this package; ``training_set_kwargs`` / ``augment_pipe`` default to None as in the reference (:36-40)."""
    data = _LegacyUnpickler(f).load()
    if isinstance(data, tuple) and len(data) == 3 and all(isinstance(net, _TFNetworkStub) for net in data):
        raise ValueError('TensorFlow StyleGAN2 pickles (legacy.py:27-33) carry no tri-plane generator; convert them with the reference first')
    assert isinstance(data, dict), 'not a network checkpoint'
    for key, value in list(data.items()):
        data[key] = persistence.rebuild(value)
    for optional in ('training_set_kwargs', 'augment_pipe'):              # older snapshots lack them (:36-40)
        data.setdefault(optional, None)
    expected = {'G': torch.nn.Module, 'D': torch.nn.Module, 'G_ema': torch.nn.Module,
                'training_set_kwargs': (dict, type(None)), 'augment_pipe': (torch.nn.Module, type(None))}
    for key, kind in expected.items():
        assert isinstance(data[key], kind), f'checkpoint entry {key!r} has type {type(data[key]).__name__}'

    if force_fp16:                                                                   # :49-60
        for key in ['G', 'D', 'G_ema']:
            old = data[key]
            kwargs = copy.deepcopy(old.init_kwargs)
            fp16_kwargs = kwargs.get('synthesis_kwargs', kwargs)
            fp16_kwargs['num_fp16_res'] = 4
            fp16_kwargs['conv_clamp'] = 256
            if kwargs != old.init_kwargs:
                new = type(old)(*old.init_args, **kwargs).eval().requires_grad_(False)
                misc.copy_params_and_buffers(old, new, require_all=True)
                data[key] = new
    from .torch_utils.ops import modconv          # a process that reloads networks must not be served weight forms derived from the old ones
    modconv.invalidate_caches()
    return data


def save_network_pkl(data, f):
    """The writer ``training_loop.py:455-470`` inlines: ``pickle.dump`` of the snapshot dict (networks already on the CPU)."""
    pickle.dump(data, f)


class _TFNetworkStub(dnnlib.EasyDict):
    pass


def _load_storage_bytes(b):
    """``torch.storage._load_from_bytes`` of a pickled storage, decoded by torch's allow-listed loader rather than a nested pickle."""
    return torch.load(io.BytesIO(b), weights_only=True)


_BUILTINS = {'object', 'set', 'frozenset', 'dict', 'list', 'tuple', 'slice', 'range', 'complex', 'bytearray', 'int', 'float', 'bool', 'str', 'bytes'}
_MIRROR_ROOTS = ('training', 'torch_utils', 'dnnlib')


class _LegacyUnpickler(pickle.Unpickler):
    """legacy.py:68-72 plus an allow-list: a checkpoint may only name what a checkpoint needs."""

    def find_class(self, module, name):
        root = module.split('.')[0]
        if module == 'dnnlib.tflib.network' and name == 'Network':
            return _TFNetworkStub
        if module.endswith('torch_utils.persistence') and name == '_reconstruct_persistent_obj':
            return persistence._reconstruct_persistent_obj
        if root in _MIRROR_ROOTS or root == 'pix2pix3d_amd':
            # non-persistent helper classes pickled by reference (EasyDict, OSGDecoder, ImportanceRenderer, RaySampler, ...)
            target = module if root == 'pix2pix3d_amd' else 'pix2pix3d_amd.' + module
            try:
                obj = dnnlib.util.get_obj_by_name(target + '.' + name)
            except ImportError:
                obj = None
            if isinstance(obj, type) and (issubclass(obj, torch.nn.Module) or issubclass(obj, dict)):
                return obj
            if obj is None and root != 'dnnlib':
                return persistence._holder_class(name)      # a module class this package has no mirror of: parameter holder
            raise pickle.UnpicklingError(f'checkpoint names {module}.{name}, which is not a network class')
        if module == 'torch.storage' and name == '_load_from_bytes':
            return _load_storage_bytes
        if module == 'torch._utils' and name.startswith('_rebuild_'):
            return getattr(torch._utils, name)
        if module == 'torch._tensor' and name == '_rebuild_from_type_v2':
            return torch._tensor._rebuild_from_type_v2
        if module == 'torch':
            obj = getattr(torch, name, None)
            if isinstance(obj, torch.dtype) or name in ('Size', 'device', 'Tensor') or name.endswith('Storage'):
                return obj
        if module in ('torch.nn.parameter', 'torch.nn.modules.container', 'torch.nn.modules.activation', 'torch.nn.modules.linear',
                      'torch.nn.modules.conv', 'torch.nn.modules.normalization', 'torch.nn.modules.batchnorm', 'torch.nn.modules.dropout',
                      'torch.nn.modules.pooling', 'torch.nn.modules.upsampling', 'torch.nn.modules.padding', 'torch.nn.modules.flatten'):
            obj = super().find_class(module, name)
            if isinstance(obj, type):
                return obj
        if module == 'collections' and name in ('OrderedDict', 'defaultdict', 'deque'):
            return super().find_class(module, name)
        if module in ('copyreg', 'copy_reg') and name in ('_reconstructor', '__newobj__'):       # protocol <= 2 writes the Python-2 module names
            return super().find_class(module, name)
        if module in ('builtins', '__builtin__') and name in _BUILTINS:
            return super().find_class(module, name)
        if module == '_codecs' and name == 'encode':                 # how protocol <= 2 spells a bytes object
            return super().find_class(module, name)
        if root == 'numpy':
            if (module in ('numpy.core.multiarray', 'numpy._core.multiarray') and name in ('_reconstruct', 'scalar')) or \
               (module == 'numpy' and name in ('ndarray', 'dtype')) or (module in ('numpy.core.numeric', 'numpy._core.numeric') and name == '_frombuffer'):
                if module.startswith('numpy.core.'):
                    module = module.replace('numpy.core.', 'numpy._core.') if hasattr(np, '_core') else module
                return super().find_class(module, name)
        raise pickle.UnpicklingError(f'checkpoint names {module}.{name}, which network checkpoints have no use for; refusing to resolve it')


def main(argv=None):
    """``python -m pix2pix3d_amd.legacy --source in.pkl --dest out.pkl [--force-fp16]`` — legacy.py:302-321 (re-save a checkpoint
    in this package's writer, i.e. with the mirrors' source embedded instead of the reference's)."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--source', required=True)
    ap.add_argument('--dest', required=True)
    ap.add_argument('--force-fp16', action='store_true')
    args = ap.parse_args(argv)
    with open(args.source, 'rb') as f:
        data = load_network_pkl(f, force_fp16=args.force_fp16)
    with open(args.dest, 'wb') as f:
        save_network_pkl(data, f)


if __name__ == '__main__':
    main()
