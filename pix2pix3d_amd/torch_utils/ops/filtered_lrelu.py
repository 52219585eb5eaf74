"""``filtered_lrelu``: bias -> upsampling FIR -> leaky ReLU (gain, clamp) -> downsampling FIR (StyleGAN3).

Mirror of the reference operator API (torch_utils/ops/filtered_lrelu.py:58-118).  No pix2pix3D configuration ever
calls this op (its only caller, networks_stylegan3.SynthesisLayer, is imported but never instantiated — SURVEY §2
note N1), so it is served by the same decomposition the reference itself uses whenever its plugin reports "no
specialised kernel" (filtered_lrelu.py:225-231): this package's native ``bias_act`` and ``upfirdn2d`` kernels in
sequence.  Gradients of every order come from those ops' own autograd Functions, which is equivalent to the
reference's sign-tensor formulation (the lrelu sign/clamp mask is exactly what ``bias_act``'s saved output encodes).
A single fused kernel with the bit-packed sign tensor (filtered_lrelu.cu:143-1103) is not implemented.
"""
import numpy as np
import torch

from . import bias_act
from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl='cuda'):
    """x [N,C,H,W]; fu/fd FIR filters from ``upfirdn2d.setup_filter``; output size
    ``(in*up + pad0 + pad1 - (fu-1) - (fd-1) + (down-1)) // down`` per axis (filtered_lrelu.py:58-118)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ['ref', 'cuda']
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype and tuple(b.shape) == (x.shape[1],)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    px0, px1, py0, py1 = _parse_padding(padding)
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    n, c, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    op_impl = impl if x.device.type == 'cuda' else 'ref'
    y = bias_act.bias_act(x=x, b=b, impl=op_impl)
    y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, impl=op_impl)
    y = bias_act.bias_act(x=y, act='lrelu', alpha=slope, gain=gain, clamp=clamp, impl=op_impl)
    y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter, impl=op_impl)
    assert tuple(y.shape) == (n, c, out_h, out_w) and y.dtype == x.dtype
    return y


_filtered_lrelu_ref = filtered_lrelu


class _Plugin:
    """``filtered_lrelu_plugin`` stand-in (filtered_lrelu.cpp:20-23, :217): reports "no specialised kernel" (-1) so a
    caller written against the reference's protocol takes its generic route; the in-place activation helper is
    served by ``bias_act``."""

    @staticmethod
    def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filters, write_signs):
        return torch.empty([0], device=x.device), torch.empty([0], device=x.device), -1

    @staticmethod
    def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, write_signs):
        y = bias_act.bias_act(x=x, act='lrelu', alpha=slope, gain=gain, clamp=(clamp if clamp >= 0 else None))
        x.copy_(y)
        return torch.empty([0], device=x.device)
