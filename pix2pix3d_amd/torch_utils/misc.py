"""Small helpers the model code relies on (reference: torch_utils/misc.py:84 assert_shape, :102
profiled_function, :157 copy_params_and_buffers, :113 InfiniteSampler, :194 check_ddp_consistency)."""
import contextlib
import re
import warnings

import numpy as np
import torch

_constant_cache = dict()


def constant(value, shape=None, dtype=None, device=None, memory_format=None):
    """Cached constant tensor (reference: misc.py:24-50)."""
    value = np.asarray(value)
    shape = None if shape is None else tuple(shape)
    dtype = dtype or torch.get_default_dtype()
    device = device or torch.device('cpu')
    memory_format = memory_format or torch.contiguous_format
    key = (value.shape, value.dtype, value.tobytes(), shape, dtype, device, memory_format)
    t = _constant_cache.get(key)
    if t is None:
        t = torch.as_tensor(value.copy(), dtype=dtype, device=device)
        if shape is not None:
            t, _ = torch.broadcast_tensors(t, torch.empty(shape))
        t = t.contiguous(memory_format=memory_format)
        _constant_cache[key] = t
    return t


def nan_to_num(input, nan=0.0, posinf=None, neginf=None, *, out=None):
    return torch.nan_to_num(input, nan=nan, posinf=posinf, neginf=neginf, out=out)


@contextlib.contextmanager
def suppress_tracer_warnings():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        yield


def assert_shape(tensor, ref_shape):
    """Check ndim and every non-None entry of ref_shape (reference: misc.py:84-97)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is None:
            continue
        if isinstance(ref, torch.Tensor) or isinstance(size, torch.Tensor):
            with suppress_tracer_warnings():
                torch._assert(torch.as_tensor(size) == torch.as_tensor(ref), f'Wrong size for dimension {idx}')
        elif size != ref:
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref}')


def profiled_function(fn):
    """Wrap fn in a record_function scope named after it (reference: misc.py:102-107)."""
    def decorator(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    decorator.__name__ = fn.__name__
    decorator.__wrapped__ = fn
    return decorator


class InfiniteSampler(torch.utils.data.Sampler):
    """Endless, optionally shuffling, rank-sharded index stream (reference: misc.py:113-144):
    rank r of num_replicas yields every index whose running position is == r mod num_replicas."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        pos = 0
        while True:
            i = pos % order.size
            if pos % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            pos += 1


def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


@torch.no_grad()
def copy_params_and_buffers(src_module, dst_module, require_all=False, allow_mismatch=False):
    """Copy same-named tensors src -> dst (reference: misc.py:157-175, incl. the '_semantic' name fallback)."""
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all), f'{name} missing in source module'      # before the fallback, as misc.py:163
        src_name = name
        if src_name not in src and '_semantic' in src_name:
            src_name = src_name.replace('_semantic', '')
        if src_name in src:
            s = src[src_name].detach()
            if s.shape == tensor.shape:
                tensor.copy_(s).requires_grad_(tensor.requires_grad)
            elif not allow_mismatch:
                raise AssertionError(f'shape mismatch for {name}: {tuple(s.shape)} vs {tuple(tensor.shape)}')
    from .ops import modconv                      # derived weight forms (modulated / bf16x3 layouts) of the overwritten parameters must not survive
    modconv.invalidate_caches()


def check_ddp_consistency(module, ignore_regex=None):
    """Assert every rank holds rank 0's values (reference: misc.py:194-205)."""
    assert isinstance(module, torch.nn.Module)
    prefix = type(module).__name__ + '.'
    skip = re.compile(ignore_regex) if ignore_regex is not None else None
    for name, tensor in named_params_and_buffers(module):
        if skip is not None and skip.fullmatch(prefix + name):
            continue
        mine = tensor.detach()
        mine = nan_to_num(mine) if mine.is_floating_point() else mine
        theirs = mine.clone()
        torch.distributed.broadcast(tensor=theirs, src=0)
        assert bool((mine == theirs).all()), prefix + name


def __getattr__(name):
    """Host-side helpers that are not restated here resolve to the reference checkout's own ``torch_utils.misc`` (see dropin.reference_attr)."""
    from .. import dropin
    return dropin.reference_attr('torch_utils.misc', name)
