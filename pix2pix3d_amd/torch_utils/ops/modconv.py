"""Native inference path of the StyleGAN2 synthesis layers on channels-last activations (fp16 or fp32).

Not a module of the reference: it bundles what the reference spreads over ``modulated_conv2d`` (fused branch,
training/networks_stylegan2.py:34-69, 81-91), ``conv2d_resample`` (:114-136), the noise add and ``bias_act``
(networks_stylegan2.py:319-332) into calls of the MFMA implicit-GEMM kernels of libp3d_hip.so (csrc/conv2d.hip):
``p3d_modulate_weights`` -> ``p3d_conv2d_nhwc`` (+ ``upfirdn2d`` for the x2 layers) and ``p3d_torgb_nhwc_f16``.
``SynthesisLayer`` / ``ToRGBLayer`` use it when ``layer_supported`` says so (device tensor, channels_last, inference,
per-sample "fused" modulation, enough pixels to fill 128-row MFMA tiles); every other case keeps the generic route.
"""
import ctypes
import os

import torch

from ... import _lib
from . import upfirdn2d, bias_act

enabled = True
prefetch_styles = True       # run every layer's style affine + weight modulation ahead of the convolutions on a second stream
_plan = {}                   # id(layer) -> (styles, (wmod, route tag) or None, event), filled by SynthesisNetwork.forward
_side = {}


def side_stream(device):
    st = _side.get(device)
    if st is None:
        st = _side[device] = torch.cuda.Stream(device=device)
    return st


_plan_seq = [0]              # position of the newest event on the prefetch stream (monotonic for the life of the process)
after_prefetch = []          # callables a network's prefetch_styles runs once its own plan is issued (the plans of networks LATER in the step: triplane._render)
_ahead = {}                  # id(network) -> (the ws it was planned for, the plan's keys): plans issued ahead, picked up by that network's forward
_plan_own_until = [0]        # waits for positions up to this one are for the entry's OWN event (the current network's first layer, then its ToRGB group); later ones for everything issued
_plan_latest = [None, -1]    # (event, position): the newest event on the prefetch stream
_plan_waited = {}            # consuming stream (its handle) -> position on the prefetch stream (the sequence number of an event) it already waits behind


def take_plan(layer):
    """Pop this layer's prefetched (styles, premodulated weights) and make the current stream wait for them.
    Entries carry the POSITION of their event on the prefetch stream (prefetch_styles numbers them; layers that launched nothing new share the event of the
    last one that did): a layer whose position the stream already waits behind adds no second wait.  In the captured step every such wait is an edge
    between two branches of the graph — idle device in front of the layer's first kernel whether or not the event fired long ago (27 of them per step,
    5.8 us each under the profiler: profiles/round5_u_step_trace.txt; same-box A/B of the elision: +0.4 ... +1.1 %, profiles/round6_a_*).
    The waits for a network's first layer and for its ToRGB group (issued right behind it: prefetch_styles) are for those events themselves — the first layers
    must not stand behind every modulation of the step; any LATER one is for everything issued so far (``_plan_latest``: by then — the first per-image-weight
    layer, hundreds of microseconds into the network — the prefetch stream has long run dry), so a network costs three edges, and the heads whose plans were
    issued ahead none."""
    hit = _plan.pop(id(layer), None)
    if hit is None:
        return None
    seq = hit[3] if len(hit) > 3 else None
    cur = torch.cuda.current_stream()
    waited = _plan_waited.get(cur.cuda_stream, -1)
    if seq is None or not plan_wait_elision or seq > waited:
        ev = hit[2]
        if seq is not None and plan_wait_elision and plan_wait_latest and seq > _plan_own_until[0] and _plan_latest[0] is not None:
            ev, seq = _plan_latest
        cur.wait_event(ev)
        if seq is not None:
            _plan_waited[cur.cuda_stream] = seq
    return hit[0], hit[1]


def plan_joined(stream):
    """True when ``stream`` already waits behind everything on the prefetch stream (finish_prefetch then adds no further edge)."""
    return plan_wait_elision and _plan_latest[0] is not None and _plan_waited.get(stream.cuda_stream, -1) >= _plan_latest[1]


plan_wait_elision = os.environ.get('P3D_PLAN_WAIT_ELISION', '1') != '0'      # take_plan skips waits its stream already stands behind; 0 = one wait per layer (A/B)
plan_wait_latest = os.environ.get('P3D_PLAN_WAIT_LATEST', '1') != '0'        # ... and a stream's SECOND wait on a plan is for everything issued so far (0 = the layer's own event)
premodulate_rgb = os.environ.get('P3D_PREMODULATE_RGB', '1') != '0'          # ToRGB layers' weight modulation on the prefetch stream too (premodulate_torgb)
sr_prefetch_ahead = os.environ.get('P3D_SR_PREFETCH_AHEAD', '1') != '0'      # the super-resolution heads' plans issued from inside the backbone's forward (superresolution.prefetch_ahead)


_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_lib.register('p3d_modulate_weights', ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp])
_lib.register('p3d_conv2d_nhwc', ctypes.c_int, [_vp] * 3 + [ctypes.c_int] + [_vp] * 4 + [_i32] * 5 + [_i64, _i32, _i32, _i32, _f32, _f32, _vp])
_lib.register('p3d_conv2d_nhwc_ws', ctypes.c_int, [_vp] * 3 + [ctypes.c_int] + [_vp] * 4 + [_i32] * 5 + [_i64, _i32, _i32, _i32, _f32, _f32, _vp, _i64, _vp])
_lib.register('p3d_conv2d_nhwc_workspace', _i64, [ctypes.c_int] + [_i32] * 5 + [_i64, _i32, _i32])

_lib.register('p3d_conv3x3_torgb_f16', ctypes.c_int, [_vp] * 8 + [_i32, _f32] + [_i32] * 5 + [ctypes.c_int64, _i32, _f32, _f32, _vp])
_lib.register('p3d_conv3x3_torgb_split', ctypes.c_int, [_vp] * 11 + [_i32, _f32] + [_i32] * 5 + [ctypes.c_int64, _i32, _f32, _f32, _vp])
_lib.register('p3d_conv2d_nhwc_scaled', ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp] + [_i32] * 5 + [ctypes.c_int64] + [_i32] * 3 + [_f32, _f32, _vp, ctypes.c_int64, _vp])
_lib.register('p3d_demod_coefs', ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp])
_lib.register('p3d_conv2d_nhwc_scaled_in', ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int] + [_vp] * 6 + [_i32] * 5 + [ctypes.c_int64, _i32, _i32, _i32, _f32, _f32, _vp, ctypes.c_int64, _vp])
_lib.register('p3d_up2_fir_bf16x3', ctypes.c_int, [_vp] * 8 + [_i32] * 5 + [ctypes.c_int64, _f32, _i32, _f32, _f32, _vp])
_lib.register('p3d_up2_fir_f16', ctypes.c_int, [_vp] * 8 + [_i32] * 5 + [ctypes.c_int64, _f32, _i32, _f32, _f32, _vp])
_lib.register('p3d_fir4_bias_act_nhwc', ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int] + [_i32] * 9 + [_f32, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _vp])
_lib.register('p3d_fc_forward', ctypes.c_int, [_vp] * 4 + [_i32] * 3 + [_i64, _f32, _f32, _i32, _f32, _f32, _f32, _vp])
_lib.register('p3d_im2col3x3', ctypes.c_int, [_vp, _vp] + [_i32] * 6 + [_i64] * 4 + [_vp])
_lib.register('p3d_noise_bias_act', ctypes.c_int, [_vp] * 5 + [_i32] * 4 + [_f32, _f32, _f32, _vp])

min_pixels = 1               # every layer takes this module (the vendor conv library is never entered: its choices for the small
                             # layers — naive kernels on a fresh box — cost milliseconds)
gemm_max_pixels = int(os.environ.get('P3D_GEMM_MAX_PIXELS', 0))
# Layers whose input has at most this many pixels per image run as im2col / col2im + one batched LIBRARY GEMM per layer.  0 (default):
# no such layer — the low-resolution layers (4^2 .. 32^2, K = 4608) take the MFMA kernel too, with its K steps dealt out to many
# work-groups (split-K, csrc/conv2d.hip): same speed as the library GEMM route on the benchmark (402 vs 402 img/s), no vendor GEMM
# and no im2col / col2im launches.  The GEMM route is kept behind the environment variable for A/B measurements only.

_zero_pages = {}


def _zeros_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(256, dtype=torch.float32, device=device)
    return z
_lib.register('p3d_torgb_nhwc_f16', ctypes.c_int, [_vp] * 5 + [_i32] * 4 + [_f32, _i32, _vp])
_lib.register('p3d_conv2d_nhwc_bf16x3_io_plan', ctypes.c_int, [_i32] * 5 + [ctypes.c_int64, _i32, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_int64)])
_lib.register('p3d_conv2d_nhwc_bf16x3_io', ctypes.c_int, [_vp] * 7 + [_i32] * 5 + [ctypes.c_int64, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp, ctypes.c_int64, _vp])
_lib.register('p3d_torgb_wide_split', ctypes.c_int, [_vp] * 6 + [_i32] * 5 + [_f32, _vp])
_lib.register('p3d_fir4_bias_act_nhwc_split', ctypes.c_int, [_vp] * 3 + [_i32] * 9 + [_f32, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _vp])


def _is_nhwc(x, dtypes=(torch.float16, torch.float32)):
    x = _raw(x)
    return x.is_cuda and x.dtype in dtypes and x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last)


def _is_nhwc_f16(x):
    return _is_nhwc(x, (torch.float16,))


def _no_grad_needed(*tensors):
    return not (torch.is_grad_enabled() and any(t is not None and _raw(t).requires_grad for t in tensors))


def _dense_dev(x):
    x = _raw(x)
    return x.is_cuda and x.dtype in (torch.float16, torch.float32) and x.ndim == 4 and (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last))


gemm_max_pixels_up = int(os.environ.get('P3D_GEMM_MAX_PIXELS_UP', 0))     # same switch for the x2 layers (see gemm_max_pixels)


def is_small(x, up=1):
    """Images this small run as one batched GEMM (any dense layout); larger ones need channels_last for the MFMA kernel."""
    return x.shape[2] * x.shape[3] <= (gemm_max_pixels if up == 1 else gemm_max_pixels_up)


def layer_supported(x, weight, styles, noise_mode, fused_modconv, up):
    """True when the native kernels cover this SynthesisLayer call."""
    if not enabled or not fused_modconv or up not in (1, 2) or not _dense_dev(x):
        return False
    if tuple(weight.shape[2:]) != (3, 3) or noise_mode == 'random' or not (is_small(x, up) or _is_nhwc(x)):
        return False
    return _no_grad_needed(x, weight, styles)


def torgb_supported(x, weight, styles, fused_modconv):
    if not enabled or not fused_modconv or tuple(weight.shape[2:]) != (1, 1) or not _dense_dev(x) or not _no_grad_needed(x, weight, styles):
        return False
    if _is_nhwc_f16(x) and x.shape[1] in (64, 128, 256) and weight.shape[0] <= 32 and (x.shape[2] * x.shape[3]) % 4 == 0:
        return True                                   # skinny streaming kernel
    if is_small(x):
        return True                                   # batched GEMM
    return _is_nhwc(x) and x.shape[1] % (64 if x.dtype == torch.float16 else 32) == 0      # 1x1 through the MFMA kernel


BF16X3 = 'bf16x3'            # dtype tag: fp32 tensors whose products run as three bf16 MFMAs of (hi, lo) splits (csrc/conv2d.hip)
DTYPE_F32_BF16X3 = 3         # p3d_dtype code of that formulation (include/p3d_hip.h)
DTYPE_F32_BF16X6 = 4         # P3D_F32_BF16X6: plain fp32 tensors and weights, every product as SIX bf16 MFMAs of three-piece splits made in registers (fp32-accurate)
f32_x6 = os.environ.get('P3D_F32_BF16X6', '1') == '1'      # the fp32 convolutions that would run on the f32-input MFMA run as bf16x6 instead — inference layers here (when bf16x3 is off),
                             # training-mode forward / data gradient in conv2d_gradfix; ignored wherever bf16x3 is selected.  On by default since the whole GPU suite passed
                             # under it with unchanged bounds and its error against fp64 is the exact kernels' (DESIGN 2.4c); '0' = every product on v_mfma_f32_32x32x2_f32
wgrad_x6 = os.environ.get('P3D_WGRAD_BF16X6', '1') == '1'     # with f32_x6: the fp32 WEIGHT gradients of whole 128 x 128 tiles (both sides > 64 channels) the same way
fuse_up2_f32_min_res = int(os.environ.get('P3D_FUSE_UP2_F32_MIN_RES', 1 << 30))      # fp32 (bf16x3) x2 layers from this input resolution up take the one-kernel form.
                             # OFF by default: measured SLOWER than transposed conv + FIR (256->128 @128^2: 331 vs 302 us, 512->256 @64^2: 313 vs 234 us, batch 4) — its fp32
                             # tile needs 131 KB of LDS, i.e. one 4-wave block per CU with nothing to hide the staging behind; kept, with its parity test, as the starting point
fuse_up2 = os.environ.get('P3D_FUSE_UP2', '1') != '0'       # fp16 x2 layers: transposed conv + FIR + epilogue in one kernel (csrc/up2_fir.hip); 0 = the two-kernel form
split_bf16 = os.environ.get('P3D_BF16X3', '1') != '0'      # use it for the fp32 layers that are bound by the fp32 matrix rate
split_bf16_min_pixels = int(os.environ.get('P3D_BF16X3_MIN_PIXELS', 16))       # every fp32 3x3 layer (measured: 4096 -> 462, 1024 -> 472, 256 -> 474, 16 -> 476 img/s)
split_activations = os.environ.get('P3D_SPLIT_ACTS', '1') != '0'               # bf16x3 inference: activations stay split between layers (SplitActs)


class SplitActs:
    """fp32 activations [N, C, H, W] of a bf16x3 inference pass, held channels-last as the K rows the matrix cores consume: per pixel and 32 channels
    [32 x bf16 hi | 32 x bf16 lo] in the 128 bytes of 32 floats (hi = bf16(v), lo = bf16(v - hi)).  Written by the producing layer's epilogue
    (p3d_conv2d_nhwc_bf16x3_io with y_split, p3d_fir4_bias_act_nhwc_split), read by the next layer's kernel without the per-tap split in registers
    (13-21 % of those kernels, profiles/round3_ae_*).  Deliberately NOT a tensor: anything that is not one of the consuming kernels has to call
    ``dense()`` = hi + lo, within 2^-17 of the value; a bf16x3 consumer of that works with a (hi, lo) pair that stands for the same number."""
    __slots__ = ('t',)

    def __init__(self, t):
        assert t.dtype == torch.float32 and t.ndim == 4 and t.shape[1] % 32 == 0 and t.is_contiguous(memory_format=torch.channels_last)
        self.t = t                                         # storage; its float VALUES are meaningless

    shape = property(lambda self: self.t.shape)
    ndim = property(lambda self: 4)
    dtype = property(lambda self: torch.float32)
    device = property(lambda self: self.t.device)
    is_cuda = property(lambda self: self.t.is_cuda)
    requires_grad = False

    def is_contiguous(self, memory_format=torch.contiguous_format):
        return memory_format == torch.channels_last

    def dense(self):
        n, c, h, w = self.t.shape
        rows = self.t.permute(0, 2, 3, 1).reshape(n, h, w, c // 32, 32).view(torch.bfloat16).reshape(n, h, w, c // 32, 2, 32).float()
        v = (rows[..., 0, :] + rows[..., 1, :]).reshape(n, h, w, c)
        return v.permute(0, 3, 1, 2)                       # [N, C, H, W] with channels-last strides


def _raw(x):
    return x.t if isinstance(x, SplitActs) else x


def accepts_split_input(n, ci, in_pixels, up):
    """Will synthesis_layer / torgb run a layer with this input (per-image pixels, batch) on a kernel that reads SplitActs?  (The per-image bf16x3
    routes do; the GEMM route of tiny images, the shared-weight form — it scales x first — and the one-kernel fp32 x2 layer do not.)"""
    if not (enabled and split_bf16 and split_activations) or ci % 32 != 0 or in_pixels < split_bf16_min_pixels:
        return False
    if in_pixels <= (gemm_max_pixels if up == 1 else gemm_max_pixels_up):
        return False
    if shared_weight_max_pixels > 0 and 1 < n <= 16 and in_pixels <= shared_weight_max_pixels:
        return False
    return not (up == 2 and fuse_up2 and fuse_up2_f32_min_res <= 1 << 20)


def use_split_bf16(x, ci):
    return split_bf16 and x.dtype == torch.float32 and x.shape[2] * x.shape[3] >= split_bf16_min_pixels and ci % 32 == 0


shared_weight_max_pixels = int(os.environ.get('P3D_SHARED_W_MAX_PIXELS', 1024))    # fp32 layers whose input has at most this many pixels per image (<= 32^2)
                                                                                  # take the shared-weight form at batch > 1; 0 = always per-image weights


def use_shared_weights(x, weight, styles):
    """The low-resolution fp32 layers of a batch are bound by their weights: N modulated copies of a 9.4 MB tensor are written and read back
    for a few KB of activations.  The unfused form of the same function (networks_stylegan2.py:70-79: x * styles -> convolution with the
    UNMODULATED weights -> * demodulation coefficients) reads one weight tensor for the whole batch, with the batch folded into the GEMM
    rows; the scaling passes over the (tiny) activations instead."""
    return (shared_weight_max_pixels > 0 and x.shape[0] > 1 and x.shape[0] <= 16 and x.dtype == torch.float32 and x.shape[2] * x.shape[3] <= shared_weight_max_pixels
            and use_split_bf16(x, weight.shape[1]) and weight.shape[1] % 4 == 0 and tuple(weight.shape[2:]) == (3, 3))


def shared_split_weights(weight):
    """The unmodulated weights in the bf16x3 K-row layout, [1][Co][9][Ci]; once per weight version."""
    def make():
        ones = torch.ones([1, weight.shape[1]], dtype=torch.float32, device=weight.device)
        return modulate_weights(weight, ones, demodulate=False, dtype=BF16X3)
    return _cached_weight(weight, 'bf16x3_shared', make)


def demod_coefs(weight, styles):
    """rsqrt(sum over (i, taps) of (w * s)^2 + 1e-8) as [N, Co] fp32, from the per-(o, i) tap sums of w^2 (cached per weight version)."""
    w2 = _cached_weight(weight, 'w2_tapsum', lambda: weight.detach().float().square().sum(dim=[2, 3]).contiguous())
    s32 = styles.detach().float().contiguous()
    n, ci = s32.shape
    d = torch.empty([n, weight.shape[0]], dtype=torch.float32, device=weight.device)
    code = _lib.lib().p3d_demod_coefs(_lib.ptr(s32), _lib.ptr(w2), _lib.ptr(d), n, ci, weight.shape[0], _lib.stream_of(d))
    _lib.check(code, 'demod_coefs')
    return d


class _DemodJob(ctypes.Structure):
    _fields_ = [('styles', ctypes.c_void_p), ('w2', ctypes.c_void_p), ('d', ctypes.c_void_p), ('ci', ctypes.c_int32), ('co', ctypes.c_int32)]


DEMOD_MAX_JOBS = 24
_lib.register('p3d_demod_coefs_multi', ctypes.c_int, [ctypes.c_void_p, _i32, _i32, _vp])


def demod_coefs_many(pairs):
    """demod_coefs for several (weight, styles) pairs with the same row count in ONE launch (bit-identical to one call each)."""
    if len(pairs) == 1 or len(pairs) > DEMOD_MAX_JOBS or len({st.shape[0] for _, st in pairs}) != 1:
        return [demod_coefs(w, st) for w, st in pairs]
    arr, keep, outs = (_DemodJob * len(pairs))(), [], []
    n = pairs[0][1].shape[0]
    for k, (weight, styles) in enumerate(pairs):
        w2 = _cached_weight(weight, 'w2_tapsum', lambda weight=weight: weight.detach().float().square().sum(dim=[2, 3]).contiguous())
        s32 = styles.detach().float().contiguous()
        d = torch.empty([n, weight.shape[0]], dtype=torch.float32, device=weight.device)
        keep.append((w2, s32))
        outs.append(d)
        arr[k] = _DemodJob(_lib.ptr(s32), _lib.ptr(w2), _lib.ptr(d), s32.shape[1], weight.shape[0])
    _lib.check(_lib.lib().p3d_demod_coefs_multi(ctypes.cast(arr, ctypes.c_void_p), len(pairs), n, _lib.stream_of(outs[0])), 'demod_coefs_multi')
    return outs


def scale_input(x, styles):
    """x * styles[:, :, None, None] on a dense device tensor, one launch (csrc/bcast_ops.hip), no autograd (inference route)."""
    from . import bcast
    geo = bcast.layout(x)
    assert geo is not None
    return bcast._scale_fma(x, geo, styles.detach().float().contiguous(), None)


def modulate_weights(weight, styles, demodulate=True, pre_scale=1.0, dtype=torch.float16, oihw=False):
    """weight [O,I,kh,kw] fp32, styles [N,I] -> ``dtype`` [N][O][kh*kw][I] (or [N][O][I][kh*kw] with ``oihw``), demodulation folded in.
    ``dtype=BF16X3``: an fp32-typed tensor of the same shape whose K rows hold [32 x bf16 hi | 32 x bf16 lo] per 32 input channels."""
    o, i, kh, kw = weight.shape
    n = styles.shape[0]
    w32 = weight.detach()
    if w32.dtype != torch.float32 or not w32.is_contiguous():      # (channels-last parameters of the fp16 blocks: one copy per weight version, not per call)
        w32 = _cached_weight(weight, 'f32_contiguous', lambda: weight.detach().float().contiguous())
    s32 = styles.detach().float().contiguous()
    split = dtype == BF16X3
    out = torch.empty([n, o, i, kh * kw] if oihw else [n, o, kh * kw, i], dtype=torch.float32 if split else dtype, device=weight.device)
    code = _lib.lib().p3d_modulate_weights(_lib.ptr(w32), _lib.ptr(s32), _lib.ptr(out), DTYPE_F32_BF16X3 if split else _lib.DTYPE_CODE[dtype], n, o, i, kh * kw,
                                           int(demodulate), float(pre_scale), int(oihw), _lib.stream_of(out))
    _lib.check(code, 'modulate_weights')
    return out


def _pad_channels(x, wmod, transposed=False):
    """Zero-pad Ci to a whole 128-byte K row where the kernel needs one (the x2 fp16 kernel works on 64-byte rows: the 32-channel
    SR input goes in as it is)."""
    mult = 64 if x.dtype == torch.float16 else 32
    ci = x.shape[1]
    if ci % mult == 0:
        return x, wmod
    if transposed and x.dtype == torch.float16 and ci % 32 == 0 and wmod.shape[1] % 128 == 0 and x.shape[2] >= 32 and x.shape[3] >= 32:
        return x, wmod
    pad = mult - ci % mult
    xp = torch.empty([x.shape[0], ci + pad, x.shape[2], x.shape[3]], dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
    xp[:, :ci] = x
    wp = torch.zeros([*wmod.shape[:3], ci + pad], dtype=wmod.dtype, device=wmod.device)
    wp[..., :ci] = wmod
    return xp, wp


fuse_input_scale = os.environ.get('P3D_FUSE_INPUT_SCALE', '1') != '0'      # shared-weight layers: x * styles inside the convolution kernel (p3d_conv2d_nhwc_scaled_in); 0 = a pass of its own


def conv2d(x, wmod, transposed=False, bias=None, noise=None, noise_strength=None, act=0, gain=1.0, clamp=-1.0, down=1, split=False, out_scale=None, out_split=False,
           in_scale=None):
    """x NHWC [N,Ci,H,W] (channels_last strides), wmod [N or 1][Co][k*k][Ci] of the same dtype -> NHWC, same dtype.
    k*k = 9: 3x3 "same" correlation, or (transposed) the stride-2 transposed conv [N,Co,2H+1,2W+1], or (down=2) the valid
    stride-2 correlation [N,Co,(H-3)//2+1,(W-3)//2+1]; k*k = 1: 1x1.
    bf16x3 (``split``): x may be a SplitActs; ``out_split`` asks for one back (granted by the 3x3 halo-slab kernel; otherwise a tensor)."""
    x_split = isinstance(x, SplitActs)
    if x_split or out_split:
        assert split and out_scale is None
        return _conv2d_split_io(x, wmod, transposed, bias, noise, noise_strength, act, gain, clamp, down, out_split)
    assert _is_nhwc(x) and wmod.dtype == x.dtype and wmod.is_contiguous() and wmod.shape[2] in (1, 9)
    assert down in (1, 2) and not (transposed and down == 2)
    x, wmod = _pad_channels(x, wmod, transposed)
    n, ci, h, w = x.shape
    co, k = wmod.shape[1], (3 if wmod.shape[2] == 9 else 1)
    oh, ow = (2 * h + 1, 2 * w + 1) if transposed else (((h - k) // 2 + 1, (w - k) // 2 + 1) if down == 2 else (h, w))
    y = torch.empty([n, co, oh, ow], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    assert wmod.shape[0] in (1, n)
    stride = 0 if wmod.shape[0] == 1 else wmod.shape[1] * wmod.shape[2] * wmod.shape[3]
    b32 = None if bias is None else bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    mode = 1 if transposed else (2 if down == 2 else 0)
    code_dtype = DTYPE_F32_BF16X3 if split else _lib.DTYPE_CODE[x.dtype]          # split: wmod came from modulate_weights(dtype=BF16X3)
    x6 = f32_x6 and not split and x.dtype == torch.float32 and ci % 32 == 0
    if x6:
        code_dtype = DTYPE_F32_BF16X6
    assert not split or x.dtype == torch.float32
    nbytes = int(_lib.lib().p3d_conv2d_nhwc_workspace(code_dtype, n, h, w, ci, co, stride, k, mode))
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device) if nbytes > 0 else None          # split-K partial tiles (low-resolution layers)
    with _lib.kernel_timer('conv_bf16x3' if split else ('conv_f16' if x.dtype == torch.float16 else ('conv_bf16x6' if x6 else 'conv_f32')), x):
        code = None
        if in_scale is not None:                           # [N, Ci] fp32 on the activations as they enter the matrix cores (shared-weight form: the styles)
            assert out_scale is not None and split and tuple(in_scale.shape) == (n, ci) and in_scale.dtype == torch.float32 and in_scale.is_contiguous()
            code = _lib.lib().p3d_conv2d_nhwc_scaled_in(_lib.ptr(x), _lib.ptr(wmod), _lib.ptr(y), code_dtype, _lib.ptr(in_scale), _lib.ptr(out_scale), _lib.ptr(b32), _lib.ptr(nz),
                                                        _lib.ptr(ns), _lib.ptr(_zeros_page(x.device)), n, h, w, ci, co, stride, k, mode, int(act), float(gain), float(clamp),
                                                        _lib.ptr(work), nbytes, _lib.stream_of(x))
            if code == _lib.P3D_ERR_UNSUPPORTED:           # more scale rows per tile than the kernel's table holds: scale x in a pass of its own
                x, code = scale_input(x, in_scale), None
        if code is not None:
            pass
        elif out_scale is not None:                        # [N, Co] fp32 on the accumulator (shared-weight form: the demodulation coefficients)
            assert out_scale.dtype == torch.float32 and out_scale.is_contiguous() and tuple(out_scale.shape) == (n, co) and x.dtype == torch.float32
            code = _lib.lib().p3d_conv2d_nhwc_scaled(_lib.ptr(x), _lib.ptr(wmod), _lib.ptr(y), code_dtype, _lib.ptr(out_scale), _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns),
                                                     _lib.ptr(_zeros_page(x.device)), n, h, w, ci, co, stride, k, mode, int(act), float(gain), float(clamp),
                                                     _lib.ptr(work), nbytes, _lib.stream_of(x))
        else:
            code = _lib.lib().p3d_conv2d_nhwc_ws(_lib.ptr(x), _lib.ptr(wmod), _lib.ptr(y), code_dtype, _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns),
                                                 _lib.ptr(_zeros_page(x.device)), n, h, w, ci, co, stride, k, mode, int(act), float(gain), float(clamp),
                                                 _lib.ptr(work), nbytes, _lib.stream_of(x))
    _lib.check(code, 'conv2d_nhwc')
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:                                  # bench.py: FLOPs of the launches it is timing (2*Ci*Co*k*k per output / input pixel)
        log.append(('bf16x3' if split else ('bf16x6' if x6 else str(x.dtype)), 2.0 * n * ci * co * k * k * (oh * ow if down == 2 else h * w)))
    return y


_io_plans = {}


def _split_io_plan(*key):
    """(y_split granted, split-K scratch bytes) of the route p3d_conv2d_nhwc_bf16x3_io takes for these sizes: asked once per geometry, so that a layer
    whose grid cannot hand back a split result is launched once, with the flag and the scratch its route really uses."""
    hit = _io_plans.get(key)
    if hit is None:
        nbytes = ctypes.c_int64(0)
        granted = int(_lib.lib().p3d_conv2d_nhwc_bf16x3_io_plan(*key, ctypes.byref(nbytes)))
        _lib.check(min(granted, 0), 'conv2d_nhwc_bf16x3_io_plan')
        hit = _io_plans[key] = (granted, int(nbytes.value))
    return hit


def _conv2d_split_io(x, wmod, transposed, bias, noise, noise_strength, act, gain, clamp, down, out_split):
    """conv2d's bf16x3 form with the activations split on one or both sides (p3d_conv2d_nhwc_bf16x3_io)."""
    x_split = isinstance(x, SplitActs)
    xt = _raw(x)
    assert _is_nhwc(xt, (torch.float32,)) and wmod.dtype == torch.float32 and wmod.is_contiguous() and wmod.shape[2] in (1, 9) and xt.shape[1] % 32 == 0
    assert down in (1, 2) and not (transposed and down == 2)
    n, ci, h, w = xt.shape
    co, k = wmod.shape[1], (3 if wmod.shape[2] == 9 else 1)
    oh, ow = (2 * h + 1, 2 * w + 1) if transposed else (((h - k) // 2 + 1, (w - k) // 2 + 1) if down == 2 else (h, w))
    y = torch.empty([n, co, oh, ow], dtype=torch.float32, device=xt.device, memory_format=torch.channels_last)
    assert wmod.shape[0] in (1, n) and wmod.shape[3] == ci
    stride = 0 if wmod.shape[0] == 1 else wmod.shape[1] * wmod.shape[2] * wmod.shape[3]
    b32 = None if bias is None else bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    mode = 1 if transposed else (2 if down == 2 else 0)
    y_split, nbytes = _split_io_plan(n, h, w, ci, co, stride, k, mode, int(x_split), int(bool(out_split) and co % 32 == 0 and k == 3 and mode == 0))
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=xt.device) if nbytes > 0 else None          # scratch of the route that will run, if it has any
    with _lib.kernel_timer('conv_bf16x3', xt):
        code = _lib.lib().p3d_conv2d_nhwc_bf16x3_io(_lib.ptr(xt), _lib.ptr(wmod), _lib.ptr(y), _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns), _lib.ptr(_zeros_page(xt.device)),
                                                    n, h, w, ci, co, stride, k, mode, int(act), float(gain), float(clamp), int(x_split), y_split,
                                                    _lib.ptr(work), nbytes, _lib.stream_of(xt))
    _lib.check(code, 'conv2d_nhwc_bf16x3_io')
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append(('bf16x3', 2.0 * n * ci * co * k * k * (oh * ow if down == 2 else h * w)))
    return SplitActs(y) if y_split else y


conv3x3 = conv2d


_plain_weights = {}


def _cached_weight(weight, tag, make):
    """Derived form of an (inference-time constant) weight tensor, rebuilt when the tensor object, its storage or its version
    changes.  The entry holds a weak reference to the tensor itself: ``id()`` and ``data_ptr()`` are both recycled once a module
    is freed, so a key made of them alone can hand a new layer another layer's weights."""
    import weakref
    slot = (id(weight), tag)
    hit = _plain_weights.get(slot)
    if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version):
        return hit[2]
    value = make()
    if len(_plain_weights) > 4096:                       # dead entries of freed modules
        for k in [k for k, v in _plain_weights.items() if v[0]() is None]:
            del _plain_weights[k]
    _plain_weights[slot] = (weakref.ref(weight), (weight.data_ptr(), weight._version), value)
    return value


def invalidate_caches():
    """Drop every derived weight form.  The cache notices in-place updates of a parameter through its autograd version counter; writes that
    bypass it (``param.data.copy_()``, ``dist.broadcast(param.data)``) do not bump it — code that overwrites parameters that way (dp.broadcast_module,
    misc.copy_params_and_buffers, the checkpoint loader) calls this afterwards."""
    _plain_weights.clear()
    for hook in _invalidate_hooks:
        hook()


_invalidate_hooks = []                                   # other modules' caches of parameter-derived tensors (networks_stylegan2: b4's batch of constants)


def on_invalidate(hook):
    """Register a zero-argument callable that drops another module's parameter-derived cache whenever ``invalidate_caches()`` runs."""
    _invalidate_hooks.append(hook)
    return hook


def plain_layer_supported(x, weight, up, down, activation):
    """Conv2dLayer calls (networks_stylegan2.py:135-188) the native kernels cover: inference on the device, 1x1 / 3x3, down in {1, 2}."""
    if not enabled or up != 1 or down not in (1, 2) or not _dense_dev(x) or activation not in ('linear', 'lrelu'):
        return False
    k = weight.shape[2]
    if weight.shape[2] != weight.shape[3] or k not in (1, 3):
        return False
    if x.shape[2] * x.shape[3] <= gemm_max_pixels and (x.dtype != torch.float32 or (x.shape[2] * x.shape[3]) % (4 * down * down) != 0):
        return False                                     # the small (GEMM) route is fp32 and wants hw % 4 == 0 after decimation
    return _no_grad_needed(x, weight)


def _plain_small(x, weight, bias, weight_gain, resample_filter, down, padding, act, act_gain, clamp):
    """Low-resolution Conv2dLayer: one library GEMM with shared weights, [Co, Ci*k*k] @ [N, Ci*k*k, OH*OW] (im2col in one launch)."""
    co, ci, k, _ = weight.shape
    n, _, h, w = x.shape
    wm = _cached_weight(weight, ('gemm', float(weight_gain)),
                        lambda: (weight.detach().float() * float(weight_gain)).reshape(co, ci * k * k).contiguous())
    fw = resample_filter.shape[-1]
    p0, p1 = padding + (fw - down + 1) // 2, padding + (fw - down) // 2
    if k == 1:
        if down == 2:
            x = upfirdn2d.upfirdn2d(x, resample_filter, down=down, padding=[p0, p1, p0, p1])
        oh, ow = x.shape[2], x.shape[3]
        cols = x.contiguous().reshape(n, ci, oh * ow)
    elif down == 1:
        oh, ow = h, w
        cols = im2col3x3(x)
    else:
        x = upfirdn2d.upfirdn2d(x, resample_filter, padding=[p0, p1, p0, p1])
        oh, ow = (x.shape[2] - 3) // 2 + 1, (x.shape[3] - 3) // 2 + 1
        cols = im2col3x3(x, pad=0, stride=2)
    y = torch.matmul(wm, cols).reshape(n, co, oh, ow)
    return noise_bias_act(y, bias, None, None, act, act_gain, clamp)


def plain_layer(x, weight, bias, weight_gain, resample_filter, down, padding, act, act_gain, clamp):
    """Conv2dLayer.forward on the MFMA kernels: weight * gain re-laid tap-major once per weight version (shared across the
    batch), then conv2d_resample's routing (conv2d_resample.py:96-136): plain "same" conv; down = 2 with a 3x3: low-pass FIR
    at full rate then the valid stride-2 conv; down = 2 with a 1x1: FIR + decimate, then the 1x1.  Bias, activation, gain
    and clamp ride in the conv epilogue.  Input any dense layout, output channels-last."""
    co, ci, k, _ = weight.shape
    if x.shape[2] * x.shape[3] <= gemm_max_pixels:
        return _plain_small(x, weight, bias, weight_gain, resample_filter, down, padding, act, act_gain, clamp)
    split = use_split_bf16(x, ci)                          # fp32 layers (the label-map Encoder of G.mapping, fp32 discriminator blocks) as bf16x3 too;
                                                           # narrow inputs (the 6-channel label map: K = 6 products, no averaging) stay exact fp32
    wdt = BF16X3 if split else x.dtype
    wmod = _cached_weight(weight, ('mfma', wdt, float(weight_gain)),
                          lambda: modulate_weights(weight, torch.ones([1, ci], dtype=torch.float32, device=weight.device), demodulate=False,
                                                   pre_scale=float(weight_gain), dtype=wdt))
    x = x.contiguous(memory_format=torch.channels_last)
    act_idx = {'linear': 0, 'lrelu': 1}[act]
    clampv = -1.0 if clamp is None else float(clamp)
    if down == 1:
        assert padding == k // 2
        return conv2d(x, wmod, bias=bias, act=act_idx, gain=act_gain, clamp=clampv, split=split)
    fw = resample_filter.shape[-1]
    p0, p1 = padding + (fw - down + 1) // 2, padding + (fw - down) // 2
    if k == 1:
        x = upfirdn2d.upfirdn2d(x, resample_filter, down=down, padding=[p0, p1, p0, p1])
        return conv2d(x, wmod, bias=bias, act=act_idx, gain=act_gain, clamp=clampv, split=split)
    x = upfirdn2d.upfirdn2d(x, resample_filter, padding=[p0, p1, p0, p1])
    return conv2d(x, wmod, bias=bias, act=act_idx, gain=act_gain, clamp=clampv, down=2, split=split)


def fc_supported(x, weight, bias, activation):
    """FullyConnectedLayer calls the one-launch kernel covers: a few fp32 rows on the device, inference."""
    return (enabled and x.is_cuda and x.ndim == 2 and x.dtype == torch.float32 and 1 <= x.shape[0] <= 16
            and x.shape[0] * ((x.shape[1] + 3) // 4 * 4) <= 16384 and activation in ('linear', 'lrelu') and weight.dtype == torch.float32
            and _no_grad_needed(x, weight, bias))          # (in_features that are not a multiple of 4 — the 25 camera parameters — are zero-padded in fc())


def fc(x, weight, bias, weight_gain, bias_gain, activation='linear', out_scale=1.0):
    """act((x @ weight.T) * weight_gain + bias * bias_gain) * def_gain * out_scale in one launch (networks_stylegan2.py:113-127)."""
    n, out_f = x.shape[0], weight.shape[0]
    x32 = x.detach()
    in_f = x.shape[1]
    if in_f % 4 != 0:                                      # whole float4 rows for the kernel: zero columns change nothing
        pad = 4 - in_f % 4
        x32 = torch.nn.functional.pad(x32, [0, pad])
        w32 = _cached_weight(weight, ('fc_pad', pad), lambda: torch.nn.functional.pad(weight.detach().float(), [0, pad]).contiguous())
        in_f += pad
    else:
        w32 = weight.detach().contiguous()
    if x32.stride(1) != 1 or x32.stride(0) % 4 != 0 or x32.data_ptr() % 16 != 0:
        x32 = x32.contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    y = torch.empty([n, out_f], dtype=torch.float32, device=x.device)
    act_gain = bias_act.activation_funcs[activation].def_gain
    code = _lib.lib().p3d_fc_forward(_lib.ptr(x32), _lib.ptr(w32), _lib.ptr(b32), _lib.ptr(y), n, in_f, out_f, x32.stride(0) if n > 1 else in_f, float(weight_gain), float(bias_gain),
                                     {'linear': 1, 'lrelu': 3}[activation], 0.2, float(act_gain), float(out_scale), _lib.stream_of(x))
    _lib.check(code, 'fc_forward')
    return y


class _FcJob(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('b', ctypes.c_void_p), ('y', ctypes.c_void_p), ('x_row_stride', ctypes.c_int64),
                ('in_features', ctypes.c_int32), ('out_features', ctypes.c_int32), ('weight_gain', ctypes.c_float), ('bias_gain', ctypes.c_float),
                ('act', ctypes.c_int32), ('alpha', ctypes.c_float), ('act_gain', ctypes.c_float), ('out_scale', ctypes.c_float)]


FC_MAX_JOBS = 40
_lib.register('p3d_fc_multi', ctypes.c_int, [ctypes.c_void_p, _i32, _i32, _vp])


def fc_multi(jobs):
    """Several FullyConnectedLayer evaluations in one launch.  ``jobs``: list of (x [n, in], layer, out_scale) with ``layer`` a
    FullyConnectedLayer every one of which passes fc_supported and all x with the same row count; returns the list of outputs."""
    n = jobs[0][0].shape[0]
    assert all(x.shape[1] % 4 == 0 for x, _, _ in jobs)
    outs, keep = [], []
    arr = (_FcJob * len(jobs))()
    for k, (x, layer, out_scale) in enumerate(jobs):
        x32 = x.detach()
        if x32.stride(1) != 1 or x32.stride(0) % 4 != 0 or x32.data_ptr() % 16 != 0:
            x32 = x32.contiguous()
        w32 = layer.weight.detach().contiguous()
        b32 = None if layer.bias is None else layer.bias.detach().float().contiguous()
        y = torch.empty([n, w32.shape[0]], dtype=torch.float32, device=x.device)
        keep.append((x32, w32, b32))
        outs.append(y)
        arr[k] = _FcJob(_lib.ptr(x32), _lib.ptr(w32), _lib.ptr(b32), _lib.ptr(y), x32.stride(0) if n > 1 else x.shape[1], x.shape[1], w32.shape[0],
                        float(layer.weight_gain), float(layer.bias_gain), {'linear': 1, 'lrelu': 3}[layer.activation], 0.2,
                        float(bias_act.activation_funcs[layer.activation].def_gain), float(out_scale))
    code = _lib.lib().p3d_fc_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), n, _lib.stream_of(outs[0]))
    _lib.check(code, 'fc_multi')
    return outs


def im2col3x3(x, pad=1, stride=1):
    """[N, C, H, W] fp32 in any dense layout -> [N, C*9, OH*OW] (pad 1, stride 1: = F.unfold(x, 3, padding=1)), one launch for the batch."""
    n, c, h, w = x.shape
    oh, ow = (h + 2 * pad - 3) // stride + 1, (w + 2 * pad - 3) // stride + 1
    cols = torch.empty([n, c * 9, oh * ow], dtype=torch.float32, device=x.device)
    code = _lib.lib().p3d_im2col3x3(_lib.ptr(x), _lib.ptr(cols), n, c, h, w, pad, stride, x.stride(0), x.stride(1), x.stride(2), x.stride(3), _lib.stream_of(x))
    _lib.check(code, 'im2col3x3')
    return cols


def noise_bias_act(y, bias, noise, noise_strength, act, act_gain, clamp):
    """In place on a contiguous fp32 [N, C, H, W]: + noise * strength, + bias, activation, gain, clamp (networks_stylegan2.py:326-332)."""
    n, c, h, w = y.shape
    b32 = None if bias is None else bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    code = _lib.lib().p3d_noise_bias_act(_lib.ptr(y), _lib.ptr(y), _lib.ptr(nz), _lib.ptr(ns), _lib.ptr(b32), n, c, h * w, {'linear': 1, 'lrelu': 3}[act], 0.2,
                                         float(act_gain), -1.0 if clamp is None else float(clamp), _lib.stream_of(y))
    _lib.check(code, 'noise_bias_act')
    return y


def _small_layer(x, weight, styles, up, wm=None):
    """Per-sample modulated 3x3 conv (or its stride-2 transposed form) for images too small for the MFMA tiles: the
    classic lowering to ONE batched GEMM per layer.  up == 1: im2col then W[n] @ cols[n]; up == 2: (W[n]^T arranged
    [Co*9, Ci]) @ x[n] then col2im (F.fold, stride 2) -> [N, Co, 2H+1, 2W+1].  Any dense layout in, NCHW out."""
    n, ci, h, w = x.shape
    co = weight.shape[0]
    if up == 1:
        if wm is None:
            wm = modulate_weights(weight, styles, demodulate=True, dtype=x.dtype, oihw=True)
        wm = wm.reshape(n, co, ci * 9)
        cols = im2col3x3(x) if x.dtype == torch.float32 else torch.nn.functional.unfold(x.contiguous(), kernel_size=3, padding=1)   # [N, Ci*9, H*W]
        return torch.bmm(wm, cols).reshape(n, co, h, w)
    if wm is None:
        wm = modulate_weights(weight, styles, demodulate=True, dtype=x.dtype, oihw=False)   # [N, Co, 9, Ci]
    cols = torch.bmm(wm.reshape(n, co * 9, ci), x.contiguous().reshape(n, ci, h * w))      # [N, Co*9, H*W]
    return torch.nn.functional.fold(cols, output_size=(2 * h + 1, 2 * w + 1), kernel_size=3, stride=2)


def premodulate_many(items):
    """premodulate for a list of (weight, styles, up, in_pixels, dtype) — ``in_pixels`` None (a ToRGB layer) gives None — with the demodulation coefficients of
    all the shared-weight layers among them computed by ONE launch.
    A GENERATOR: each layer's product is launched when it is asked for, so that the caller can record the event a layer waits on right behind that layer's own
    work (the shared-weight coefficients — the low-resolution layers, which run first — are all issued at the first request)."""
    shared = [k for k, (w, st, up, px, dt) in enumerate(items) if px is not None and _premod_route(w, st, up, px, dt) == 'shared']
    ds = dict(zip(shared, demod_coefs_many([(items[k][0], items[k][1]) for k in shared]))) if shared else {}
    for k in shared:
        shared_split_weights(items[k][0])                  # (warm the per-weight cache off the critical path; all of them HERE, so that a later shared layer launches nothing)
    for k, (w, st, up, px, dt) in enumerate(items):
        if px is None:
            yield None
        elif k in ds:
            yield (ds[k], ('shared', up, BF16X3))
        else:
            yield premodulate(w, st, up, px, dt)


def premodulate_launches(items):
    """For the list premodulate_many takes: does asking for item k launch anything?  (A ToRGB layer: nothing.  A shared-weight layer: everything of all of them
    at the FIRST one.  Anything else: its own modulation.)  prefetch_styles gives the layers that launch nothing the event of the last one that did."""
    out, first_shared = [], True
    for w, st, up, px, dt in items:
        if px is None:
            out.append(False)
        elif _premod_route(w, st, up, px, dt) == 'shared':
            out.append(first_shared)
            first_shared = False
        else:
            out.append(True)
    return out


def _premod_route(weight, styles, up, in_pixels, dtype):
    """'gemm' / 'shared' / 'mfma': which of premodulate's three products a layer gets."""
    if in_pixels <= (gemm_max_pixels if up == 1 else gemm_max_pixels_up):
        return 'gemm'
    if split_bf16 and dtype == torch.float32 and in_pixels >= split_bf16_min_pixels and weight.shape[1] % 32 == 0:
        if (shared_weight_max_pixels > 0 and 1 < styles.shape[0] <= 16 and in_pixels <= shared_weight_max_pixels and weight.shape[1] % 4 == 0
                and tuple(weight.shape[2:]) == (3, 3)):
            return 'shared'
    return 'mfma'


def premodulate_torgb(weight, styles, pixels, dtype):
    """A ToRGB layer's modulated weights (no demodulation: networks_stylegan2.py:355-359) in the form the layer's route will ask for — bf16x3 K rows for the wide
    fp32 image of a block with ``pixels`` pixels (torgb(), torgb_wide_skip), plain fp32 otherwise (the fused ToRGB of the fp16 heads) — with the tag the consumers
    compare: a 3-6 us launch per ToRGB layer that otherwise sits in line in front of the layer (eleven per step)."""
    ci = weight.shape[1]
    split = split_bf16 and dtype == torch.float32 and pixels >= split_bf16_min_pixels and ci % 32 == 0
    wtag = BF16X3 if split else torch.float32
    return modulate_weights(weight, styles, demodulate=False, dtype=wtag), ('rgb', wtag)


def premodulate(weight, styles, up, in_pixels, dtype):
    """The modulated weights synthesis_layer will want for a layer whose input has ``in_pixels`` pixels per image, in the layout of
    the route it will take; returns (tensor, route tag).  Used to run every layer's modulation ahead of the convolutions on a
    second stream (SynthesisNetwork.forward)."""
    small = in_pixels <= (gemm_max_pixels if up == 1 else gemm_max_pixels_up)
    if small:
        return modulate_weights(weight, styles, demodulate=True, dtype=dtype, oihw=(up == 1)), ('gemm', up, dtype)
    if split_bf16 and dtype == torch.float32 and in_pixels >= split_bf16_min_pixels and weight.shape[1] % 32 == 0:
        if (shared_weight_max_pixels > 0 and 1 < styles.shape[0] <= 16 and in_pixels <= shared_weight_max_pixels and weight.shape[1] % 4 == 0
                and tuple(weight.shape[2:]) == (3, 3)):
            shared_split_weights(weight)                   # (warm the per-weight cache off the critical path)
            return demod_coefs(weight, styles), ('shared', up, BF16X3)
        return modulate_weights(weight, styles, demodulate=True, dtype=BF16X3), ('mfma', up, BF16X3)
    return modulate_weights(weight, styles, demodulate=True, dtype=dtype), ('mfma', up, dtype)


def synthesis_layer(x, weight, styles, bias, up, resample_filter, noise_const=None, noise_strength=None, act='lrelu', act_gain=1.0, clamp=None, pre=None, rgb=None,
                    out_split=False):
    """Whole SynthesisLayer body after the style affine: modulated 3x3 conv (x2 up when ``up == 2``) + noise + bias + act.
    ``pre`` = (modulated weights, route tag) from ``premodulate`` — used when the tag matches the route taken here.
    x may be a SplitActs (bf16x3 inference); ``out_split`` asks for the result as one (granted where the producing kernel can)."""
    small = is_small(x, up)
    split = (not small) and use_split_bf16(x, weight.shape[1])
    if isinstance(x, SplitActs) and not (split and accepts_split_input(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], up)):
        x = x.dense()
    out_split = bool(out_split) and split and split_activations and act in ('linear', 'lrelu')
    wtag = BF16X3 if split else x.dtype
    wpre = pre[0] if pre is not None and pre[1] == ('gemm' if small else 'mfma', up, wtag) else None
    if small:
        y = _small_layer(x, weight, styles, up, wpre)
        if up == 2:
            y = upfirdn2d.upfirdn2d(y, resample_filter, padding=[1, 1, 1, 1], gain=4)
        if y.dtype == torch.float32 and act in ('linear', 'lrelu') and (y.shape[2] * y.shape[3]) % 4 == 0 and y.is_contiguous():
            return noise_bias_act(y, bias, noise_const, noise_strength, act, act_gain, clamp)
        if noise_const is not None:
            y = y.add_((noise_const * noise_strength).to(y.dtype))
        return bias_act.bias_act(y, (None if bias is None else bias.to(y.dtype)), act=act, gain=act_gain, clamp=clamp)
    act_idx = {'linear': 0, 'lrelu': 1}.get(act)
    clampv = -1.0 if clamp is None else float(clamp)
    if split and use_shared_weights(x, weight, styles):
        d = pre[0] if pre is not None and pre[1] == ('shared', up, BF16X3) else demod_coefs(weight, styles)
        wsh = shared_split_weights(weight)
        if fuse_input_scale and x.shape[1] % 32 == 0:      # x * styles happens inside the convolution (bit-identical to the separate pass)
            xs, isc = x, styles.detach().float().contiguous()
        else:
            xs, isc = scale_input(x, styles), None
        if up == 1 and act_idx is not None:
            return conv2d(xs, wsh, bias=bias, noise=noise_const, noise_strength=noise_strength, act=act_idx, gain=act_gain, clamp=clampv, split=True, out_scale=d, in_scale=isc)
        if up == 1:
            y = conv2d(xs, wsh, noise=noise_const, noise_strength=noise_strength, split=True, out_scale=d, in_scale=isc)
            return bias_act.bias_act(y, (None if bias is None else bias.to(y.dtype)), act=act, gain=act_gain, clamp=clamp)
        y = conv2d(xs, wsh, transposed=True, split=True, out_scale=d, in_scale=isc)
        if act_idx is not None and tuple(resample_filter.shape) == (4, 4) and y.shape[1] % 32 == 0:
            return fir4_bias_act(y, resample_filter, bias, noise_const, noise_strength, act, act_gain, clampv, out_split=out_split)
        y = upfirdn2d.upfirdn2d(y, resample_filter, padding=[1, 1, 1, 1], gain=4)
        if noise_const is not None:
            y = y.add_((noise_const * noise_strength).to(y.dtype))
        return bias_act.bias_act(y, (None if bias is None else bias.to(y.dtype)), act=act, gain=act_gain, clamp=clamp)
    wmod = wpre if wpre is not None else modulate_weights(weight, styles, demodulate=True, dtype=wtag)
    if rgb is not None:                                    # (rgb_weight, rgb_styles, rgb_bias, rgb_clamp, img[, x_dead]): checked by torgb_fusable
        rgb_w, rgb_s, rgb_b, rgb_c, img = rgb[:5]
        rgb_pre = rgb[6] if len(rgb) > 6 else None
        rgb_wmod = rgb_pre[0] if rgb_pre is not None and rgb_pre[1] == ('rgb', torch.float32) else modulate_weights(rgb_w, rgb_s, demodulate=False, dtype=torch.float32)
        return conv3x3_torgb(x, wmod, bias, act_idx, act_gain, clampv, rgb_wmod, rgb_b, rgb_c, img, store_y=not (len(rgb) > 5 and rgb[5]))
    if up == 1 and act_idx is not None:
        return conv2d(x, wmod, bias=bias, noise=noise_const, noise_strength=noise_strength, act=act_idx, gain=act_gain, clamp=clampv, split=split, out_split=out_split)
    if up == 1:
        y = conv2d(x, wmod, noise=noise_const, noise_strength=noise_strength, split=split)
        return bias_act.bias_act(y, (None if bias is None else bias.to(y.dtype)), act=act, gain=act_gain, clamp=clamp)
    if fuse_up2 and act_idx is not None and x.shape[1] % 32 == 0 and wmod.shape[1] % 32 == 0 and (
            x.dtype == torch.float16 or (split and min(x.shape[2], x.shape[3]) >= fuse_up2_f32_min_res)):
        taps = _separable_fir(resample_filter)
        if taps is not None:
            return up2_fir(x, wmod, taps, bias, noise_const, noise_strength, act_idx, act_gain, clampv)
    # x2: stride-2 transposed conv as four polyphase GEMMs, then the 4x4 low-pass with gain 4 (conv2d_resample.py:114-131)
    y = conv2d(x, wmod, transposed=True, split=split)
    if act_idx is not None and tuple(resample_filter.shape) == (4, 4) and y.shape[1] % (64 if y.dtype == torch.float16 else 32) == 0:
        return fir4_bias_act(y, resample_filter, bias, noise_const, noise_strength, act, act_gain, clampv, out_split=out_split)     # FIR + noise + bias + act in one pass
    y = upfirdn2d.upfirdn2d(y, resample_filter, padding=[1, 1, 1, 1], gain=4)
    if noise_const is not None:
        y = y.add_((noise_const * noise_strength).to(y.dtype))
    return bias_act.bias_act(y, (None if bias is None else bias.to(y.dtype)), act=act, gain=act_gain, clamp=clamp)


def _separable_fir(f):
    """The 4x4 low-pass of the x2 layers as (fy, fx) in correlation order with the gain 4 folded in — eight host floats for
    p3d_up2_fir_f16 — or None when ``f`` is not 4x4 / not an outer product (setup_filter of a 1-D filter always is).  One device
    read per filter tensor, cached."""
    if tuple(f.shape) != (4, 4):
        return None

    def make():
        f64 = f.detach().double().cpu()
        r, c, tot = f64.sum(1), f64.sum(0), f64.sum()
        if float(tot) == 0.0 or not torch.allclose(torch.outer(r, c) / tot, f64, rtol=1e-6, atol=1e-9):
            return (None,)
        vals = [4.0 * float(r[3 - k]) / float(tot) for k in range(4)] + [float(c[3 - k]) for k in range(4)]       # upfirdn2d flips f (flip_filter = False)
        return ((ctypes.c_float * 8)(*vals),)
    return _cached_weight(f, 'fir_sep', make)[0]


def up2_fir(x, wmod, taps, bias, noise, noise_strength, act, act_gain, clamp):
    """The whole x2 layer in one launch: x NHWC fp16 [N,Ci,H,W], wmod [N or 1][Co][9][Ci] -> NHWC fp16 [N,Co,2H,2W]
    (csrc/up2_fir.hip: transposed conv, 4x4 FIR, noise, bias, activation, clamp; the (2H+1)^2 intermediate stays in LDS).
    fp32 x with bf16x3-layout weights (modulate_weights(dtype=BF16X3)) takes the bf16x3 build of the same kernel."""
    split = x.dtype == torch.float32
    assert _is_nhwc(x) and wmod.dtype == x.dtype and wmod.is_contiguous() and wmod.shape[2] == 9
    n, ci, h, w = x.shape
    co = wmod.shape[1]
    assert wmod.shape[0] in (1, n) and wmod.shape[3] == ci
    y = torch.empty([n, co, 2 * h, 2 * w], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    stride = 0 if wmod.shape[0] == 1 else co * 9 * ci
    b32 = None if bias is None else bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    with _lib.kernel_timer('conv_bf16x3' if split else 'conv_f16', x):
        fn = _lib.lib().p3d_up2_fir_bf16x3 if split else _lib.lib().p3d_up2_fir_f16
        code = fn(_lib.ptr(x), _lib.ptr(wmod), _lib.ptr(y), _lib.ptr(_zeros_page(x.device)), _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns),
                  ctypes.cast(taps, ctypes.c_void_p), n, h, w, ci, co, stride, 1.0, int(act), float(act_gain), float(clamp), _lib.stream_of(x))
    _lib.check(code, 'up2_fir')
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append(('bf16x3' if split else str(x.dtype), 2.0 * n * ci * co * 9 * h * w))
    return y


def fir4_bias_act(y, f, bias, noise, noise_strength, act, act_gain, clamp, out_split=False):
    """4x4 FIR (pad 1, gain 4) + noise + bias + activation on an NHWC tensor [N,C,2H+1,2W+1] -> [N,C,2H,2W] (``out_split``, fp32: a SplitActs)."""
    n, c, ih, iw = y.shape
    out = torch.empty([n, c, ih - 1, iw - 1], dtype=y.dtype, device=y.device, memory_format=torch.channels_last)
    f32 = f.detach().float().contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    if out_split and y.dtype == torch.float32 and c % 32 == 0:
        code = _lib.lib().p3d_fir4_bias_act_nhwc_split(_lib.ptr(y), _lib.ptr(f32), _lib.ptr(out), n, c, ih, iw, 1, 1, ih - 1, iw - 1, 0, 4.0,
                                                       _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns), {'linear': 1, 'lrelu': 3}[act], 0.2, float(act_gain), float(clamp),
                                                       _lib.stream_of(y))
        _lib.check(code, 'fir4_bias_act_nhwc_split')
        return SplitActs(out)
    code = _lib.lib().p3d_fir4_bias_act_nhwc(_lib.ptr(y), _lib.ptr(f32), _lib.ptr(out), _lib.DTYPE_CODE[y.dtype], n, c, ih, iw, 1, 1, ih - 1, iw - 1, 0, 4.0,
                                             _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns), {'linear': 1, 'lrelu': 3}[act], 0.2, float(act_gain), float(clamp),
                                             _lib.stream_of(y))
    _lib.check(code, 'fir4_bias_act_nhwc')
    return out


fused_torgb_calls = 0        # launches of the fused conv1 + ToRGB kernel (tests)
fuse_torgb_256 = os.environ.get('P3D_FUSE_TORGB_256', '1') != '0'      # ... and at Co = 256 (the first block of a super-resolution head: its ToRGB re-read 134 MB of activations per launch)
fuse_torgb = os.environ.get('P3D_FUSE_TORGB', '1') != '0'      # SynthesisBlock.conv1 + ToRGB + skip-image sum in one launch where the kernel allows (Co = 128, fp16)


def torgb_fusable(x, conv_weight, rgb_weight, img, up, noise_const, act):
    """Can the block's last 3x3 layer also produce its ToRGB contribution (csrc/conv2d.hip: p3d_conv3x3_torgb_f16)?  x is that layer's INPUT."""
    if not (fuse_torgb and enabled and up == 1 and noise_const is None and act in ('linear', 'lrelu') and img is not None and _is_nhwc_f16(x)):
        return False
    n, ci, h, w = x.shape
    if n * ((h + 15) // 16) * ((w + 15) // 16) < 192:         # too few 16 x 16 patches to fill the chip: the dispatcher would take the split-K kernel
        return False
    co = conv_weight.shape[0]                                  # 128: one channel block per work-group; 256 (fuse_torgb_256): each work-group walks both blocks of its patch
    return (co in ((128, 256) if fuse_torgb_256 else (128,)) and tuple(conv_weight.shape[2:]) == (3, 3) and ci % 64 == 0 and h >= 32 and w >= 32 and rgb_weight.shape[0] <= 8
            and tuple(rgb_weight.shape[1:]) == (co, 1, 1) and img.dtype == torch.float32 and img.is_contiguous() and tuple(img.shape) == (n, rgb_weight.shape[0], h, w)
            and not img.requires_grad and _no_grad_needed(x, conv_weight, rgb_weight))


def conv3x3_torgb(x, wmod, bias, act, gain, clamp, rgb_wmod, rgb_bias, rgb_clamp, img, store_y=True):
    """3x3 'same' modulated conv (fp16 NHWC, Co = 128 or 256) + epilogue, and img += clamp(ToRGB(y) + rgb_bias) from the same launch.
    ``store_y=False``: the layer's activations have no reader but this ToRGB (the last block of a super-resolution head) — they are not written
    (268 MB per launch at 512^2, batch 4) and None is returned."""
    n, ci, h, w = x.shape
    co = wmod.shape[1]
    y = torch.empty([n, co, h, w], dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if store_y else None
    stride = 0 if wmod.shape[0] == 1 else co * 9 * ci
    b32 = None if bias is None else bias.detach().float().contiguous()
    rb32 = None if rgb_bias is None else rgb_bias.detach().float().contiguous()
    rw = rgb_wmod.reshape(n, -1, co)
    assert rw.dtype == torch.float32 and rw.is_contiguous() and wmod.dtype == torch.float16 and wmod.is_contiguous()
    with _lib.kernel_timer('conv_f16', x):
        code = _lib.lib().p3d_conv3x3_torgb_f16(_lib.ptr(x), _lib.ptr(wmod), _lib.ptr(y), _lib.ptr(b32), _lib.ptr(_zeros_page(x.device)), _lib.ptr(rw), _lib.ptr(rb32),
                                                _lib.ptr(img), rw.shape[1], -1.0 if rgb_clamp is None else float(rgb_clamp), n, h, w, ci, co, stride,
                                                int(act), float(gain), float(clamp), _lib.stream_of(x))
    _lib.check(code, 'conv3x3_torgb_f16')
    global fused_torgb_calls
    fused_torgb_calls += 1
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append((str(x.dtype), 2.0 * n * ci * co * 9 * h * w))
    return y


fuse_wide_torgb = os.environ.get('P3D_FUSE_WIDE_TORGB', '1') != '0'      # the backbone's wide ToRGB + skip-image sum in one launch on split activations (csrc/torgb_split.hip)


def _filter_host(f):
    """The 4x4 resampling filter as 16 host floats (one device read per filter tensor, cached), or None."""
    if tuple(f.shape) != (4, 4):
        return None
    return _cached_weight(f, 'fir_host16', lambda: ((ctypes.c_float * 16)(*[float(v) for v in f.detach().float().cpu().reshape(-1)]),))[0]


def torgb_wide_skip_supported(x, weight, prev, f):
    """p3d_torgb_wide_split takes it: SplitActs of 128 / 256 channels, 32 / 64 / 96 outputs, rows that are whole 32-pixel tiles, and (if given) a
    channels-last fp32 predecessor image at half the resolution."""
    if not (fuse_wide_torgb and enabled and isinstance(x, SplitActs) and tuple(weight.shape[2:]) == (1, 1) and not torch.is_grad_enabled()):
        return False
    n, ci, h, w = x.shape
    co = weight.shape[0]
    if ci not in (128, 256) or co not in (32, 64, 96) or w % 32 != 0:
        return False
    if prev is None:
        return True
    return (f is not None and tuple(f.shape) == (4, 4) and prev.dtype == torch.float32 and tuple(prev.shape) == (n, co, h // 2, w // 2) and h % 2 == 0
            and prev.is_cuda and prev.is_contiguous(memory_format=torch.channels_last) and not prev.requires_grad)


def torgb_wide_skip(x, weight, styles, bias, clamp, prev, f, pre=None):
    """ToRGB of a SplitActs (wide image: the backbone's tri-planes) + the block's skip-image sum ``upsample2d(prev, f) + y`` in one launch -> fp32 NHWC."""
    n, ci, h, w = x.shape
    co = weight.shape[0]
    wmod = pre[0] if pre is not None and pre[1] == ('rgb', BF16X3) else modulate_weights(weight, styles, demodulate=False, dtype=BF16X3)
    y = torch.empty([n, co, h, w], dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    b32 = None if bias is None else bias.detach().float().contiguous()
    fh = None if prev is None else _filter_host(f)
    with _lib.kernel_timer('conv_bf16x3', x.t):
        code = _lib.lib().p3d_torgb_wide_split(_lib.ptr(x.t), _lib.ptr(wmod), _lib.ptr(b32), _lib.ptr(y), _lib.ptr(prev), None if fh is None else ctypes.cast(fh, ctypes.c_void_p),
                                               n, h, w, ci, co, -1.0 if clamp is None else float(clamp), _lib.stream_of(x.t))
    _lib.check(code, 'torgb_wide_split')
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append(('bf16x3', 2.0 * n * ci * co * h * w))
    return y


fuse_conv_wide_torgb = os.environ.get('P3D_FUSE_CONV_WIDE_TORGB', '1') != '0'      # the LAST backbone block: conv1 + wide ToRGB + skip-image sum in one launch, x never stored
                             # (csrc/conv2d.hip: conv3x3_r2_bf16x3_kernel<TR>); 0 = conv1, then torgb_wide_skip
conv_wide_torgb_calls = 0


def conv3x3_torgb_wide_supported(x, conv_weight, rgb_weight, prev, f, up, act):
    """p3d_conv3x3_torgb_split takes it: a SplitActs into a 3x3 'same' layer of 128 output channels whose result nobody but the block's wide ToRGB reads (the caller
    vouches for that), 32 / 64 / 96 image channels, enough 16 x 16 patches to fill the chip, and (if given) a channels-last fp32 predecessor image at half the resolution."""
    if not (fuse_conv_wide_torgb and fuse_wide_torgb and enabled and split_bf16 and split_activations and isinstance(x, SplitActs) and up == 1 and act in ('linear', 'lrelu')
            and not torch.is_grad_enabled()):
        return False
    n, ci, h, w = x.shape
    co, rco = conv_weight.shape[0], rgb_weight.shape[0]
    if (co != 128 or tuple(conv_weight.shape[1:]) != (ci, 3, 3) or tuple(rgb_weight.shape[1:]) != (co, 1, 1) or rco not in (32, 64, 96) or ci % 32 != 0 or h < 32 or w < 32
            or (h | w) & 1 or n * ((h + 15) // 16) * ((w + 15) // 16) < 192 or use_shared_weights(x.t, conv_weight, None) or not use_split_bf16(x.t, ci)):
        return False
    if prev is None:
        return True
    return (f is not None and tuple(f.shape) == (4, 4) and prev.dtype == torch.float32 and tuple(prev.shape) == (n, rco, h // 2, w // 2)
            and prev.is_cuda and prev.is_contiguous(memory_format=torch.channels_last) and not prev.requires_grad)


def conv3x3_torgb_wide(x, wmod, bias, noise, noise_strength, act, gain, clamp, rgb_wmod, rgb_bias, rgb_clamp, prev, f):
    """SplitActs x -> the block's fp32 NHWC image: clamp(ToRGB(act(conv3x3(x) + noise + bias) * gain) + rgb_bias) + upsample2d(prev, f), one launch; the layer's own
    activations are never written.  wmod / rgb_wmod: modulate_weights(dtype=BF16X3) of the 3x3 layer (demodulated) and of the ToRGB (not)."""
    n, ci, h, w = x.shape
    co, rco = wmod.shape[1], rgb_wmod.shape[1]
    assert wmod.dtype == torch.float32 and wmod.is_contiguous() and tuple(wmod.shape[1:]) == (co, 9, ci) and wmod.shape[0] in (1, n)
    assert rgb_wmod.dtype == torch.float32 and rgb_wmod.is_contiguous() and tuple(rgb_wmod.shape) == (n, rco, 1, co)
    img = torch.empty([n, rco, h, w], dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    stride = 0 if wmod.shape[0] == 1 else co * 9 * ci
    b32 = None if bias is None else bias.detach().float().contiguous()
    rb32 = None if rgb_bias is None else rgb_bias.detach().float().contiguous()
    nz = None if noise is None else noise.detach().float().contiguous()
    ns = None if noise is None else noise_strength.detach().float().reshape(1).contiguous()
    fh = None if prev is None else _filter_host(f)
    with _lib.kernel_timer('conv_bf16x3', x.t):
        code = _lib.lib().p3d_conv3x3_torgb_split(_lib.ptr(x.t), _lib.ptr(wmod), _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(ns), _lib.ptr(_zeros_page(x.device)), _lib.ptr(rgb_wmod),
                                                  _lib.ptr(rb32), _lib.ptr(img), _lib.ptr(prev), None if fh is None else ctypes.cast(fh, ctypes.c_void_p), rco,
                                                  -1.0 if rgb_clamp is None else float(rgb_clamp), n, h, w, ci, co, stride, int(act), float(gain), float(clamp), _lib.stream_of(x.t))
    _lib.check(code, 'conv3x3_torgb_split')
    global conv_wide_torgb_calls
    conv_wide_torgb_calls += 1
    log = _lib.kernel_events.get('conv_flops')
    if log is not None:
        log.append(('bf16x3', 2.0 * n * ci * co * 9 * h * w + 2.0 * n * co * rco * h * w))
    return img


def torgb_accumulates(x, weight, out):
    """True when torgb(..., out=out) adds into ``out`` inside its own kernel (the fp16 streaming route) rather than with a separate add."""
    n, ci, h, w = x.shape
    return (_is_nhwc_f16(x) and ci in (64, 128, 256) and weight.shape[0] <= 32 and (h * w) % 4 == 0 and out.dtype == torch.float32 and out.is_contiguous()
            and tuple(out.shape) == (n, weight.shape[0], h, w) and not out.requires_grad)


def torgb(x, weight, styles, bias, clamp=None, out=None, pre=None):
    """ToRGB: 1x1 modulated conv without demodulation + bias (+ clamp).  fp16 activations with a handful of output
    channels take the streaming kernel (-> fp32 NCHW, optionally accumulated into ``out``); wide outputs (the 96-channel
    tri-plane image of the backbone) go through the MFMA kernel as a 1x1 conv and stay channels-last."""
    n, ci, h, w = x.shape
    co = weight.shape[0]
    if isinstance(x, SplitActs) and not (not is_small(x) and use_split_bf16(x, ci)):
        x = x.dense()
    if is_small(x) and not (_is_nhwc_f16(x) and ci in (64, 128, 256) and co <= 32 and (h * w) % 4 == 0):
        wm = modulate_weights(weight, styles, demodulate=False, dtype=x.dtype).reshape(n, co, ci)
        y = torch.bmm(wm, x.contiguous().reshape(n, ci, h * w)).reshape(n, co, h, w)
        y = bias_act.bias_act(y, None if bias is None else bias.to(y.dtype), clamp=clamp)
        return y if out is None else out.add_(y)
    if not (x.dtype == torch.float16 and ci in (64, 128, 256) and co <= 32 and (h * w) % 4 == 0):
        split = use_split_bf16(x, ci)                      # the wide fp32 ToRGB (96 tri-plane channels) as bf16x3 too: as exact-fp32 MFMA it ran at a
        wtag = BF16X3 if split else x.dtype                # third of that pipe's peak, 0.29 ms per step
        wmod = pre[0] if pre is not None and pre[1] == ('rgb', wtag) else modulate_weights(weight, styles, demodulate=False, dtype=wtag)
        y = conv2d(x, wmod, bias=bias, clamp=-1.0 if clamp is None else float(clamp), split=split)
        return y if out is None else out.add_(y)
    w32 = weight.detach().float().reshape(co, ci).contiguous()
    s32 = styles.detach().float().contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    acc = out is not None
    y = out if acc else torch.empty([n, co, h, w], dtype=torch.float32, device=x.device)
    assert y.is_contiguous() and y.dtype == torch.float32
    code = _lib.lib().p3d_torgb_nhwc_f16(_lib.ptr(x), _lib.ptr(w32), _lib.ptr(s32), _lib.ptr(b32), _lib.ptr(y), n, h * w, ci, co,
                                         -1.0 if clamp is None else float(clamp), int(acc), _lib.stream_of(x))
    _lib.check(code, 'torgb_nhwc_f16')
    return y
