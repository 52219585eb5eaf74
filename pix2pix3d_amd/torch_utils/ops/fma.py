"""``fma(a, b, c) = a * b + c`` with a cheap hand-written backward (reference: torch_utils/ops/fma.py:17-60)."""
import torch


def fma(a, b, c):
    from . import bcast
    if bcast.fma_supported(a, b, c):                      # dense device activations: one fused kernel forward, fused reductions backward
        return bcast.fma(a, b, c)
    return _Fma.apply(a, b, c)


def _reduce_to(t, shape):
    """Sum ``t`` over the axes that broadcasting expanded so that the result has ``shape``."""
    lead = t.ndim - len(shape)
    assert lead >= 0
    axes = [i for i in range(t.ndim) if i < lead or (shape[i - lead] == 1 and t.shape[i] > 1)]
    if axes:
        t = t.sum(dim=axes, keepdim=True)
    if lead:
        t = t.reshape(-1, *t.shape[lead + 1:])
    assert t.shape == shape
    return t


class _Fma(torch.autograd.Function):
    """out = a * b + c; every gradient is a product reduced back to its operand's (possibly broadcast) shape (fma.py:30-47)."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        need_a, need_b, need_c = ctx.needs_input_grad
        return (_reduce_to(dout * b, a.shape) if need_a else None,
                _reduce_to(dout * a, b.shape) if need_b else None,
                _reduce_to(dout, ctx.c_shape) if need_c else None)
