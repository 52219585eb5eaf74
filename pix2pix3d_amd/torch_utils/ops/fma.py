"""``fma(a, b, c) = a * b + c`` with a cheap hand-written backward (reference: torch_utils/ops/fma.py:17-60)."""
import torch


def fma(a, b, c):
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
    @staticmethod
    def forward(ctx, a, b, c):
        out = torch.addcmul(c, a, b)
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = db = dc = None
        if ctx.needs_input_grad[0]:
            da = _reduce_to(dout * b, a.shape)
        if ctx.needs_input_grad[1]:
            db = _reduce_to(dout * a, b.shape)
        if ctx.needs_input_grad[2]:
            dc = _reduce_to(dout, ctx.c_shape)
        return da, db, dc
