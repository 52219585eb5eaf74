"""CPU oracle for the element-wise / FIR / convolution operators of the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pix2pix3d_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.  It restates the
reference's algorithms (file:line cited per function, paths relative to the reference checkout) in
numpy, written independently of both the reference text and the HIP kernels, and is pinned against
outputs of the reference itself: ``tests/golden/make_golden.py`` imports the reference in this
container and records its results for fixed seeded inputs; ``tests/test_oracle_golden.py`` checks
this file against those records.  The reference ships no golden vectors of its own.
"""
import numpy as np

ACT_INDEX = {'linear': 1, 'relu': 2, 'lrelu': 3, 'tanh': 4, 'sigmoid': 5, 'elu': 6, 'selu': 7, 'softplus': 8, 'swish': 9}
ACT_DEFAULTS = {  # name: (def_alpha, def_gain)   torch_utils/ops/bias_act.py:23-33
    'linear': (0, 1), 'relu': (0, np.sqrt(2)), 'lrelu': (0.2, np.sqrt(2)), 'tanh': (0, 1), 'sigmoid': (0, 1),
    'elu': (0, 1), 'selu': (0, 1), 'softplus': (0, 1), 'swish': (0, np.sqrt(2)),
}
_SELU_SCALE = 1.0507009873554804934193349852946
_SELU_ALPHA = 1.6732632423543772848170429916717


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def _act(name, v, alpha):
    """Forward activation, its first derivative and second derivative w.r.t. the pre-activation."""
    if name == 'linear':
        return v, np.ones_like(v), np.zeros_like(v)
    if name == 'relu':
        return np.maximum(v, 0), (v > 0).astype(v.dtype), np.zeros_like(v)
    if name == 'lrelu':
        return np.where(v > 0, v, v * alpha), np.where(v > 0, 1.0, alpha).astype(v.dtype), np.zeros_like(v)
    if name == 'tanh':
        t = np.tanh(v)
        return t, 1 - t * t, -2 * t * (1 - t * t)
    if name == 'sigmoid':
        s = _sigmoid(v)
        return s, s * (1 - s), s * (1 - s) * (1 - 2 * s)
    if name == 'elu':
        e = np.exp(np.minimum(v, 0))
        return np.where(v >= 0, v, e - 1), np.where(v >= 0, 1.0, e), np.where(v >= 0, 0.0, e)
    if name == 'selu':
        e = np.exp(np.minimum(v, 0))
        return (np.where(v >= 0, _SELU_SCALE * v, _SELU_SCALE * _SELU_ALPHA * (e - 1)),
                np.where(v >= 0, _SELU_SCALE, _SELU_SCALE * _SELU_ALPHA * e),
                np.where(v >= 0, 0.0, _SELU_SCALE * _SELU_ALPHA * e))
    if name == 'softplus':
        s = _sigmoid(v)
        return np.logaddexp(v, 0), s, s * (1 - s)
    if name == 'swish':
        s = _sigmoid(v)
        return v * s, s + v * s * (1 - s), s * (1 - s) * (2 + v * (1 - 2 * s))
    raise KeyError(name)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """y = clamp(act(x + b) * gain)   — torch_utils/ops/bias_act.py:93-122 (_bias_act_ref)."""
    x = np.asarray(x)
    a0, g0 = ACT_DEFAULTS[act]
    alpha = a0 if alpha is None else alpha
    gain = g0 if gain is None else gain
    v = x.astype(np.float64)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        v = v + np.asarray(b, np.float64).reshape(shape)
    y = _act(act, v, alpha)[0] * gain
    if clamp is not None and clamp >= 0:
        y = np.clip(y, -clamp, clamp)
    return y.astype(x.dtype)


def bias_act_grads(x, b, dy, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """First-order gradients (dx, db) of bias_act for upstream dy — the quantity the grad=1 form of the
    kernel produces (torch_utils/ops/bias_act.py:170-187, bias_act.cu:55-148): zero where the forward
    output was clamped."""
    x = np.asarray(x)
    a0, g0 = ACT_DEFAULTS[act]
    alpha = a0 if alpha is None else alpha
    gain = g0 if gain is None else gain
    v = x.astype(np.float64)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        v = v + np.asarray(b, np.float64).reshape(shape)
    y, d1, _ = _act(act, v, alpha)
    dx = np.asarray(dy, np.float64) * d1 * gain
    if clamp is not None and clamp >= 0:
        dx = np.where(np.abs(y * gain) < clamp, dx, 0.0)
    db = dx.sum(axis=tuple(i for i in range(x.ndim) if i != dim)) if b is not None else None
    return dx.astype(x.dtype), (None if db is None else db.astype(x.dtype))


def bias_act_second(x, b, dy, ddx, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """d/dx of <ddx, dx(x, dy)> — the grad=2 form (bias_act.py:196-203): dy * ddx * act''(x+b) * gain,
    zero where clamped."""
    x = np.asarray(x)
    a0, g0 = ACT_DEFAULTS[act]
    alpha = a0 if alpha is None else alpha
    gain = g0 if gain is None else gain
    v = x.astype(np.float64)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        v = v + np.asarray(b, np.float64).reshape(shape)
    y, _, d2 = _act(act, v, alpha)
    r = np.asarray(dy, np.float64) * np.asarray(ddx, np.float64) * d2 * gain
    if clamp is not None and clamp >= 0:
        r = np.where(np.abs(y * gain) < clamp, r, 0.0)
    return r.astype(x.dtype)


# ---------------------------------------------------------------------------------------------------------
def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding] * 2
    padding = list(padding)
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return [int(p) for p in padding]


def _two(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def setup_filter(f, normalize=True, flip_filter=False, gain=1, separable=None):
    """torch_utils/ops/upfirdn2d.py:72-116."""
    f = np.atleast_1d(np.asarray(1 if f is None else f, np.float32))
    if separable is None:
        separable = f.ndim == 1 and f.size >= 8
    if f.ndim == 1 and not separable:
        f = np.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f[::-1] if f.ndim == 1 else f[::-1, ::-1]
    return np.ascontiguousarray(f * gain ** (f.ndim / 2), np.float32)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Zero-insert upsample, pad/crop, FIR, decimate — torch_utils/ops/upfirdn2d.py:169-213 (_upfirdn2d_ref).

    Direct polyphase evaluation of  y[o] = gain * sum_k u[o*down + k] * g[k]  per axis, where u is the
    zero-stuffed, padded signal and g the mirrored filter (convolution) unless flip_filter."""
    x = np.asarray(x)
    n, c, h, w = x.shape
    upx, upy = _two(up)
    dnx, dny = _two(down)
    px0, px1, py0, py1 = _pad4(padding)
    f = np.ones([1, 1], np.float32) if f is None else np.asarray(f, np.float32)
    f2 = np.outer(f, f) if f.ndim == 1 else f
    fh, fw = f2.shape
    g = f2 if flip_filter else f2[::-1, ::-1]
    g = g.astype(np.float64) * gain

    uw, uh = w * upx + px0 + px1, h * upy + py0 + py1
    u = np.zeros([n, c, max(uh, 0), max(uw, 0)], np.float64)
    ys, xs = np.arange(h) * upy + py0, np.arange(w) * upx + px0
    my, mx = (ys >= 0) & (ys < uh), (xs >= 0) & (xs < uw)
    u[:, :, ys[my][:, None], xs[mx][None, :]] = x[:, :, my][:, :, :, mx]
    oh, ow = (uh - fh + dny) // dny, (uw - fw + dnx) // dnx
    y = np.zeros([n, c, oh, ow], np.float64)
    for ky in range(fh):
        for kx in range(fw):
            y += g[ky, kx] * u[:, :, ky: ky + (oh - 1) * dny + 1: dny, kx: kx + (ow - 1) * dnx + 1: dnx]
    return y.astype(x.dtype)


def conv2d(x, w, stride=1, padding=0, groups=1):
    """Plain correlation (what torch.nn.functional.conv2d computes), NCHW, float64 accumulation."""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    n, cin, h, wd = x.shape
    cout, cin_g, kh, kw = w.shape
    sy, sx = _two(stride)
    py, px = _two(padding)
    xp = np.pad(x, [(0, 0), (0, 0), (py, py), (px, px)])
    oh, ow = (h + 2 * py - kh) // sy + 1, (wd + 2 * px - kw) // sx + 1
    y = np.zeros([n, cout, oh, ow])
    cog = cout // groups
    for g in range(groups):
        xs = xp[:, g * cin_g:(g + 1) * cin_g]
        ws = w[g * cog:(g + 1) * cog]
        for ky in range(kh):
            for kx in range(kw):
                patch = xs[:, :, ky: ky + (oh - 1) * sy + 1: sy, kx: kx + (ow - 1) * sx + 1: sx]
                y[:, g * cog:(g + 1) * cog] += np.einsum('nchw,oc->nohw', patch, ws[:, :, ky, kx])
    return y


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, flip_weight=True, flip_filter=False):
    """Definition-level restatement of conv2d_resample (torch_utils/ops/conv2d_resample.py:48-143, the generic
    route :138-143 which every fast path must equal): upsample with the FIR (gain up^2), convolve, decimate
    with the FIR.  groups = 1."""
    x = np.asarray(x)
    fnp = None if f is None else np.asarray(f, np.float32)
    fw = 1 if fnp is None else fnp.shape[-1]
    fh = 1 if fnp is None else fnp.shape[0]
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:
        px0, px1, py0, py1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2, py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1, py0, py1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2
    y = upfirdn2d(x.astype(np.float64), fnp if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    wk = np.asarray(w, np.float64)
    if not flip_weight:
        wk = wk[:, :, ::-1, ::-1]
    y = conv2d(y, wk)
    if down > 1:
        y = upfirdn2d(y, fnp, down=down, flip_filter=flip_filter)
    return y.astype(x.dtype)


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None, demodulate=True, flip_weight=True):
    """Per-sample weight modulation + optional demodulation + conv (training/networks_stylegan2.py:34-91),
    evaluated sample by sample with explicitly modulated weights (the ':81-91' formulation)."""
    x = np.asarray(x)
    wt = np.asarray(weight, np.float64)
    s = np.asarray(styles, np.float64)
    outs = []
    for i in range(x.shape[0]):
        wi = wt * s[i][None, :, None, None]
        if demodulate:
            wi = wi / np.sqrt((wi ** 2).sum(axis=(1, 2, 3), keepdims=True) + 1e-8)
        outs.append(conv2d_resample(x[i:i + 1].astype(np.float64), wi, f=resample_filter, up=up, padding=padding, flip_weight=flip_weight))
    y = np.concatenate(outs, 0)
    if noise is not None:
        y = y + np.asarray(noise, np.float64)
    return y.astype(x.dtype)


# ---- filtered_lrelu with the bit-packed sign tensor -------------------------------------------------------------------------------
def _filter2d(f):
    f = np.ones([1, 1], np.float32) if f is None else np.asarray(f, np.float32)
    return np.outer(f, f) if f.ndim == 1 else f


def filtered_lrelu_sizes(x_shape, fu, fd, up, down, padding):
    """Sizes the plugin derives (torch_utils/ops/filtered_lrelu.cpp:62-97): up-sampled extent (cw, ch), output (yw, yh), sign tensor
    (rows sh, active width, bytes per row)."""
    n, c, xh, xw = x_shape
    fu2, fd2 = _filter2d(fu), _filter2d(fd)
    px0, px1, py0, py1 = _pad4(padding)
    cw, ch = xw * up + px0 + px1 - (fu2.shape[1] - 1), xh * up + py0 + py1 - (fu2.shape[0] - 1)
    yw, yh = (cw - (fd2.shape[1] - 1) + down - 1) // down, (ch - (fd2.shape[0] - 1) + down - 1) // down
    sw_active, sh = yw * down - (down - 1) + fd2.shape[1] - 1, yh * down - (down - 1) + fd2.shape[0] - 1
    return dict(cw=cw, ch=ch, yw=yw, yh=yh, sh=sh, sw_active=sw_active, sw_bytes=((sw_active + 15) & ~15) >> 2)


def pack_signs(codes):
    """[..., H, W] array of 2-bit codes -> uint8 [..., H, ceil16(W)/4]: element x in byte x >> 2, bits (x & 3) * 2
    (filtered_lrelu.cu:480-487, 512-523)."""
    h, w = codes.shape[-2:]
    wp = (w + 15) & ~15
    padded = np.zeros(codes.shape[:-1] + (wp,), np.uint8)
    padded[..., :w] = codes
    q = padded.reshape(codes.shape[:-1] + (wp // 4, 4))
    return (q[..., 0] | (q[..., 1] << 2) | (q[..., 2] << 4) | (q[..., 3] << 6)).astype(np.uint8)


def unpack_signs(packed, x, y):
    """2-bit code of sign-tensor element (x, y) per plane; 0 where (x, y) falls outside the tensor (those elements pass unchanged,
    filtered_lrelu.cu:566-576)."""
    sh, swb = packed.shape[-2:]
    ok = (x >= 0) & ((x >> 2) < swb) & (y >= 0) & (y < sh)
    xs, ys = np.where(ok, x, 0), np.where(ok, y, 0)
    byte = packed[..., ys, xs >> 2]
    return np.where(ok, (byte >> ((xs & 3) * 2)) & 3, 0)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False,
                   signs=None, sign_offset=(0, 0), write_signs=False):
    """bias -> up-sampling FIR (gain up^2) -> lrelu * gain, clamp -> down-sampling FIR
    (torch_utils/ops/filtered_lrelu.py:123-148; fused kernel filtered_lrelu.cu:143-1103).

    ``signs`` given: the backward configuration — instead of comparing, each element of the up-sampled grid takes its saved 2-bit
    code at (x + ox, y + oy): & 1 -> times slope, & 2 -> zero, outside the tensor -> unchanged; no clamp (filtered_lrelu.cu:566-576).
    ``write_signs``: also return the packed sign tensor over the plugin's extent: code 1 where the scaled value has its IEEE sign bit
    set, 2 (replacing it) where it was clamped (:497-508).  float64 arithmetic on the inputs' values."""
    x = np.asarray(x)
    n, c, xh, xw = x.shape
    px0, px1, py0, py1 = _pad4(padding)
    sz = filtered_lrelu_sizes(x.shape, fu, fd, up, down, padding)
    v = x.astype(np.float64) + (0 if b is None else np.asarray(b, np.float64).reshape(1, -1, 1, 1))
    # the up-sampled grid out to the sign tensor's extent (zeros beyond the data: what the kernel's zero-filled footprint gives)
    ext_w, ext_h = max(sz['cw'], (sz['sw_active'] + 3) & ~3), max(sz['ch'], sz['sh'])
    fu2 = _filter2d(fu)
    pad_r, pad_b = px1 + (ext_w - sz['cw']), py1 + (ext_h - sz['ch'])
    u = upfirdn2d(v, fu2, up=up, padding=[px0, pad_r, py0, pad_b], flip_filter=flip_filter, gain=1).astype(np.float64)
    u = u * (np.float32(up) * np.float32(up) * np.float32(gain)).astype(np.float64)
    ys, xs = np.meshgrid(np.arange(u.shape[2]), np.arange(u.shape[3]), indexing='ij')
    packed = None
    if signs is not None:
        code = unpack_signs(np.asarray(signs), xs + sign_offset[0], ys + sign_offset[1])
        u = np.where(code & 1, u * slope, u)
        u = np.where(code & 2, 0.0, u)
    else:
        neg = np.signbit(u)
        u = np.where(neg, u * slope, u)
        code = neg.astype(np.uint8)
        if clamp is not None:
            hit = np.abs(u) > clamp
            u = np.clip(u, -clamp, clamp)
            code = np.where(hit, 2, code).astype(np.uint8)
        if write_signs:
            packed = pack_signs(code[..., :sz['sh'], :min(code.shape[-1], (sz['sw_active'] + 3) & ~3)])
            packed = packed[..., :sz['sw_bytes']]
    y = upfirdn2d(u[..., :sz['ch'], :sz['cw']], _filter2d(fd), down=down, flip_filter=flip_filter)
    y = y.astype(x.dtype)
    return (y, packed) if write_signs else y


def filtered_lrelu_backward(dy, fu, fd, x_shape, signs, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, flip_filter=False):
    """d/dx of filtered_lrelu as the SAME op with up <-> down, fu <-> fd, mirrored filters and the saved signs
    (torch_utils/ops/filtered_lrelu.py:240-270)."""
    fu2, fd2 = _filter2d(fu), _filter2d(fd)
    _, _, xh, xw = x_shape
    _, _, yh, yw = np.asarray(dy).shape
    px0, px1, py0, py1 = _pad4(padding)
    pp = [(fu2.shape[1] - 1) + (fd2.shape[1] - 1) - px0, xw * up - yw * down + px0 - (up - 1),
          (fu2.shape[0] - 1) + (fd2.shape[0] - 1) - py0, xh * up - yh * down + py0 - (up - 1)]
    gg = gain * (up ** 2) / (down ** 2)
    off = (-(fu2.shape[1] - 1) + px0, -(fu2.shape[0] - 1) + py0)
    return filtered_lrelu(dy, fu=fd2, fd=fu2, b=None, up=down, down=up, padding=pp, gain=gg, slope=slope, clamp=None, flip_filter=not flip_filter,
                          signs=signs, sign_offset=off)
