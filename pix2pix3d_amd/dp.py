"""Data-parallel gradient exchange of the training loop.

The reference does not use DistributedDataParallel: after all backward passes of a phase it concatenates every
``param.grad`` of the phase's module into ONE flat fp32 vector, all-reduces it (SUM), divides by the world size,
sanitises it and scatters it back (training/training_loop.py:531-542).  This module is that step, kept as one flat
message on purpose — on MI355X `backend='nccl'` is RCCL over xGMI, point-to-point links where a ring all-reduce is
per-link bound, so few large messages (336 MB for G, 125 MB for D) are the best shape; nothing is overlapped
because the reference's loop is synchronous at this point and gradient accumulation rounds must all have finished.

Also here: the start-up parameter broadcast (training_loop.py:349-353) and the rank-sharded batch split
(misc.InfiniteSampler, torch_utils/misc.py:113-144) helpers used by the tests and the benchmark.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _gloo_on_device(t, group):
    """gloo (the CPU tests' backend, and the only way to run two ranks on ONE GPU: RCCL refuses a duplicate device) moves host memory; a device
    tensor is staged through the host explicitly, whatever this build's gloo would do with it."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _all_reduce(t, group):
    if _gloo_on_device(t, group):
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)                  # SUM over ranks: RCCL on the GPUs of a node


def allreduce_gradients(module_or_params, world_size=None, group=None, out=None):
    """Average gradients across ranks in place; returns the flat vector that was exchanged (or None).

    Parameters without a gradient are skipped, exactly like the reference's ``if param.grad is not None`` filter, so
    every rank must have the same set of parameters with gradients (true when all ranks run the same phase).
    ``out``: a dict the caller keeps between phases — the flat vector of each (module, size) is then allocated once and reused; the
    gradients are views of it after the call (a second exchange of the same module without a backward in between is detected by
    storage and copied through a temporary).  All gradients of one exchange must share one dtype (the reference's are fp32)."""
    owner = id(module_or_params) if isinstance(module_or_params, torch.nn.Module) else None
    params = module_or_params.parameters() if isinstance(module_or_params, torch.nn.Module) else module_or_params
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    # each parameter's piece of the flat vector is laid out in the PARAMETER's memory order (channels-last weights of the fp16 super-resolution blocks:
    # fp16_channels_last), so the gradient handed back has the parameter's strides — the optimizer's multi-tensor kernels take a list only when every
    # (param, grad, state) triple has equal strides; with contiguous gradients on channels-last weights the whole Gmain step of 194 tensors fell back to
    # per-tensor lerp_ / addcmul_ launches
    orders = [_memory_order(p) for p in params]
    grads = [(p.grad if o is None else p.grad.permute(o)).reshape(-1) for p, o in zip(params, orders)]
    assert all(g.dtype == grads[0].dtype and g.device == grads[0].device for g in grads), 'allreduce_gradients: gradients of one exchange must share dtype and device'
    if out is None:
        flat = torch.cat(grads)
    else:
        key = (owner, sum(g.numel() for g in grads), grads[0].dtype, grads[0].device)
        flat = out.get(key)
        if flat is None:
            flat = out[key] = torch.empty(key[1], dtype=key[2], device=key[3])
        base = flat.untyped_storage().data_ptr()
        if any(g.untyped_storage().data_ptr() == base for g in grads):      # some gradient still is a view of this buffer (two exchanges without a backward in between)
            flat.copy_(torch.cat(grads))
        else:
            torch.cat(grads, out=flat)
    if world_size is None:
        world_size = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world_size > 1:
        _all_reduce(flat, group)                        # SUM over ranks (RCCL on GPUs, gloo in the CPU tests)
        flat /= world_size
    torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
    for p, o, g in zip(params, orders, flat.split([p.numel() for p in params])):
        p.grad = g.reshape(p.shape) if o is None else g.as_strided(p.shape, p.stride())
    return flat


def _memory_order(p):
    """None for a contiguous parameter; otherwise the permutation of its dimensions from the slowest to the fastest in memory (dense, non-overlapping
    tensors only: anything else is treated as contiguous and simply gets a contiguous gradient)."""
    canonical, span = [], 1
    for n in reversed(p.shape):                                # the strides reshape() gives a piece of the flat vector (size-1 dimensions included: the
        canonical.append(span)                                 # multi-tensor kernels compare strides literally)
        span *= max(n, 1)
    if p.dim() < 2 or tuple(reversed(canonical)) == tuple(p.stride()):
        return None
    order = sorted(range(p.dim()), key=lambda d: (-p.stride(d), d))
    span = 1
    for d in reversed(order):                                  # dense in that order?
        if p.shape[d] != 1 and p.stride(d) != span:
            return None
        span *= p.shape[d]
    return order


def broadcast_module(module, src=0, group=None):
    """Make every rank hold rank ``src``'s parameters and buffers (training_loop.py:349-353)."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        t = t.data if t.is_leaf else t
        if _gloo_on_device(t, group):
            host = t.cpu()
            dist.broadcast(host, src=src, group=group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src, group=group)
    from .torch_utils.ops import modconv                  # writes through .data do not bump the version counter the weight caches watch
    modconv.invalidate_caches()


def shard_indices(n_items, rank, world_size):
    """Indices of a length-``n_items`` batch that belong to ``rank`` under the reference's round-robin sharding
    (position % num_replicas == rank, misc.py:139)."""
    return list(range(rank, n_items, world_size))
