"""``conv2d_resample``: convolution fused with integer up/down-sampling (reference:
torch_utils/ops/conv2d_resample.py:48-143).  Pure routing: picks the cheapest decomposition into
``conv2d_gradfix`` convolutions and ``upfirdn2d`` FIR passes; padding is applied once, relative to the
upsampled image."""
import torch

from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _parse_padding, _get_filter_size


def _get_weight_shape(w):
    return [int(s) for s in w.shape]


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d_gradfix with the correlation/convolution convention made explicit: flip_weight=False means
    a true convolution, i.e. the kernel is mirrored before the (correlating) library call."""
    kh, kw = w.shape[2:]
    if not flip_weight and (kh > 1 or kw > 1):
        w = w.flip([2, 3])
    fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return fn(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    cout, cin_g, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # the FIR's own support is absorbed into the padding so sizes come out as in*up/down
    if up > 1:
        px0, px1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2
        py0, py1 = py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2
        py0, py1 = py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2
    pads = [px0, px1, py0, py1]
    pointwise = (kw == 1 and kh == 1)

    if pointwise and down > 1 and up == 1:            # filter+decimate first: the 1x1 then sees 1/down^2 pixels
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=pads, flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)

    if pointwise and up > 1 and down == 1:            # 1x1 first on the small image, then upsample
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:                          # low-pass at full rate, then a strided conv
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=pads, flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:                                        # stride-`up` transposed conv, then the FIR (gain up^2)
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2).reshape(groups * cin_g, cout // groups, kh, kw)
        px0, px1, py0, py1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:      # plain convolution with symmetric padding
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    # anything else: explicit pad/upsample, convolve, decimate
    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
