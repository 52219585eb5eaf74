"""CPU oracle for the whole ``G.synthesis`` path: backbone -> renderer -> super-resolution head(s).

TEST INFRASTRUCTURE ONLY — see oracle/ops_oracle.py for the rules.  A functional (no nn.Module) fp32
restatement in torch CPU ops of the reference's generator forward, driven by a flat ``{name: tensor}``
state dict with the reference's parameter names:
    training/triplane_cond.py:1020-1061   TriPlaneSemanticEntangleGenerator.synthesis  (and :656-700)
    training/networks_stylegan2.py:34-91, 277-526   modulated conv, synthesis layers/blocks/network
    training/superresolution.py:191-354   super-resolution blocks
    torch_utils/ops/conv2d_resample.py:114-136, upfirdn2d.py:169-213, bias_act.py:93-122
    training/volumetric_rendering/*       (torch twin of oracle/render_oracle.py, for speed at 128^2 x 96+)
Pinned by tests/test_model_golden.py against tests/golden/model_*.npz recorded from the reference with
name-seeded weights (tests/golden/weights.py) — no checkpoint exists offline.  Also the ``cpu_baseline``
("port") leg of bench.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fir_filter(device='cpu'):
    f = torch.tensor([1., 3., 3., 1.], device=device)
    f = torch.outer(f, f)
    return f / f.sum()


def upfirdn(x, f, up=1, down=1, pad=(0, 0, 0, 0), gain=1.0):
    """upfirdn2d.py:169-213 for a symmetric 2-D filter: zero-stuff, pad, depthwise FIR, decimate."""
    n, c, h, w = x.shape
    if up > 1:
        z = x.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(n, c, h * up, w * up)
    x = F.pad(x, [max(pad[0], 0), max(pad[1], 0), max(pad[2], 0), max(pad[3], 0)])
    x = x[:, :, max(-pad[2], 0): x.shape[2] - max(-pad[3], 0), max(-pad[0], 0): x.shape[3] - max(-pad[1], 0)]
    k = (f * gain).flip([0, 1])[None, None].expand(c, 1, -1, -1)
    x = F.conv2d(x, k, groups=c)
    return x[:, :, ::down, ::down]


def upsample2(x, f):
    """upfirdn2d.upsample2d(x, f, up=2): pad (2,1,2,1), gain 4 (upfirdn2d.py:315-350)."""
    return upfirdn(x, f, up=2, pad=(2, 1, 2, 1), gain=4.0)


def bias_act(x, b=None, act='linear', gain=None, clamp=None):
    """bias_act.py:93-122."""
    if b is not None:
        x = x + b.reshape(1, -1, *([1] * (x.ndim - 2)))
    if act == 'lrelu':
        x = F.leaky_relu(x, 0.2)
        gain = SQRT2 if gain is None else gain
    else:
        assert act == 'linear'
        gain = 1.0 if gain is None else gain
    if gain != 1:
        x = x * gain
    if clamp is not None:
        x = x.clamp(-clamp, clamp)
    return x


def fc(sd, p, x, lr_mul=1.0):
    """FullyConnectedLayer, linear (networks_stylegan2.py:114-127)."""
    w = sd[p + '.weight'] * (lr_mul / math.sqrt(sd[p + '.weight'].shape[1]))
    return F.linear(x, w, sd[p + '.bias'] * lr_mul)


def modconv(x, weight, styles, up=1, demodulate=True, f=None):
    """modulated_conv2d (networks_stylegan2.py:34-91), per-sample weights through one grouped conv; up=2 is the
    stride-2 transposed conv + 4x4 FIR route of conv2d_resample.py:114-131."""
    n, cin, h, w_ = x.shape
    cout, _, k, _ = weight.shape
    w = weight[None] * styles[:, None, :, None, None]
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    if up == 1:
        y = F.conv2d(x.reshape(1, n * cin, h, w_), w.reshape(n * cout, cin, k, k), padding=k // 2, groups=n)
        return y.reshape(n, cout, *y.shape[2:])
    # true convolution with the per-sample kernel on the zero-stuffed grid == conv_transpose2d(stride 2) with
    # [in, out] weights; then the FIR with padding px0 = py0 = 1 (after the transposed conv absorbed k-1)
    wt = w.transpose(1, 2).reshape(n * cin, cout, k, k)
    y = F.conv_transpose2d(x.reshape(1, n * cin, h, w_), wt, stride=2, groups=n)
    y = y.reshape(n, cout, *y.shape[2:])
    return upfirdn(y, f, pad=(1, 1, 1, 1), gain=4.0)


def synthesis_layer(sd, p, x, w, up=1, noise_mode='const', gain=1.0, clamp=None, f=None):
    """SynthesisLayer.forward (networks_stylegan2.py:313-332)."""
    styles = fc(sd, p + '.affine', w)
    x = modconv(x, sd[p + '.weight'], styles, up=up, f=f)
    if noise_mode == 'const':
        x = x + sd[p + '.noise_const'] * sd[p + '.noise_strength']
    else:
        assert noise_mode == 'none'
    return bias_act(x, sd[p + '.bias'], act='lrelu', gain=SQRT2 * gain, clamp=None if clamp is None else clamp * gain)


def torgb(sd, p, x, w, clamp=None):
    """ToRGBLayer.forward (networks_stylegan2.py:355-359): weight gain on the styles, no demodulation."""
    wt = sd[p + '.weight']
    styles = fc(sd, p + '.affine', w) * (1 / math.sqrt(wt.shape[1] * wt.shape[2] ** 2))
    return bias_act(modconv(x, wt, styles, demodulate=False), sd[p + '.bias'], clamp=clamp)


def synthesis_block(sd, p, x, img, ws, first=False, up=True, noise_mode='const', f=None, clamp=None):
    """SynthesisBlock.forward, 'skip' architecture (networks_stylegan2.py:419-463; superresolution.py:236-290 for up=False)."""
    i = 0
    if first:
        x = sd[p + '.const'][None].repeat(ws.shape[0], 1, 1, 1)
    else:
        x = synthesis_layer(sd, p + '.conv0', x, ws[:, i], up=2 if up else 1, noise_mode=noise_mode, clamp=clamp, f=f)
        i += 1
    x = synthesis_layer(sd, p + '.conv1', x, ws[:, i], noise_mode=noise_mode, clamp=clamp, f=f)
    i += 1
    if img is not None and up:
        img = upsample2(img, f)
    y = torgb(sd, p + '.torgb', x, ws[:, i], clamp=clamp)
    return x, (y if img is None else img + y)


def backbone(sd, p, ws, resolution=256, noise_mode='const'):
    """SynthesisNetwork.forward (networks_stylegan2.py:505-520): ws index advances by num_conv per block."""
    f = fir_filter(ws.device)
    x = img = None
    idx, res = 0, 4
    while res <= resolution:
        first = (res == 4)
        nconv = 1 if first else 2
        x, img = synthesis_block(sd, f'{p}.b{res}', x, img, ws[:, idx: idx + nconv + 1], first=first, noise_mode=noise_mode, f=f)
        idx += nconv
        res *= 2
    return img


def superresolution(sd, p, rgb, x, ws, kind='8XDC', noise_mode='none', clamp=None, input_resolution=128, antialias=True):
    """SuperresolutionHybrid8XDC / 2X forward (superresolution.py:312-323, :109-121): inputs resized to the head's input
    resolution, last w repeated three times.  ``clamp``: the reference passes conv_clamp=256 whenever
    sr_num_fp16_res > 0 (:304-310) — it applies on the fp32 CPU path too."""
    f = fir_filter(ws.device)
    w3 = ws[:, -1:].repeat(1, 3, 1)
    rgb_in, resized = rgb, x.shape[-1] != input_resolution
    if resized:
        size = (input_resolution, input_resolution)
        x = F.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=antialias)
        rgb = F.interpolate(rgb, size=size, mode='bilinear', align_corners=False, antialias=antialias)
    x, rgb0 = synthesis_block(sd, p + '.block0', x, rgb, w3, up=(kind == '8XDC'), noise_mode=noise_mode, f=f, clamp=clamp)
    x, out = synthesis_block(sd, p + '.block1', x, rgb0, w3, up=True, noise_mode=noise_mode, f=f, clamp=clamp)
    # Reference quirk kept for parity: a no-upsampling block0 adds its ToRGB output to the incoming image IN PLACE
    # (superresolution.py:281 `img.add_(y)` on a view of the feature image), so the 'image_raw' / 'semantic_raw' the
    # generator returns for the 2X/4X heads already contain that residual.  `raw` is what the caller must report.
    raw = rgb_in if (kind == '8XDC' or resized) else rgb0
    return out, raw


# ---- renderer, torch twin of render_oracle.py ------------------------------------------------------------
def ray_sampler(c2w, K, r):
    n = c2w.shape[0]
    fx, fy, cx, cy, sk = K[:, 0, 0, None], K[:, 1, 1, None], K[:, 0, 2, None], K[:, 1, 2, None], K[:, 0, 1, None]
    centre = torch.arange(r, dtype=torch.float32) * (1. / r) + (0.5 / r)
    col, row = centre.repeat(r)[None], centre.repeat_interleave(r)[None]
    x = (col - cx + cy * sk / fy - sk * row / fy) / fx
    y = (row - cy) / fy
    pts = torch.stack([x, y, torch.ones_like(x), torch.ones_like(x)], -1)
    world = torch.einsum('nij,nmj->nmi', c2w, pts)[..., :3]
    o = c2w[:, None, :3, 3]
    d = F.normalize(world - o, dim=2)
    return o.expand_as(d).contiguous(), d


def _decode(sd, planes, pts, box_warp, two_nets, sem_sigmoid, lr_mul):
    n, p, _ = pts.shape
    g = pts * (2 / box_warp)
    uv = torch.stack([g[..., [0, 1]], g[..., [0, 2]], g[..., [2, 0]]], 1).reshape(n * 3, 1, p, 2)       # plane 0 (x,y), 1 (x,z), 2 (z,x)
    feat = F.grid_sample(planes.reshape(n * 3, 32, *planes.shape[-2:]), uv, mode='bilinear', padding_mode='zeros', align_corners=False)
    x = feat.reshape(n, 3, 32, p).mean(1).transpose(1, 2).reshape(n * p, 32)
    squash = lambda t: torch.sigmoid(t) * 1.002 - 0.001
    y = fc(sd, 'decoder.net.2', F.softplus(fc(sd, 'decoder.net.0', x, lr_mul)), lr_mul)
    if not two_nets:
        return squash(y[:, 1:]).reshape(n, p, -1), y[:, 0].reshape(n, p)
    ys = fc(sd, 'decoder.net_semantic.2', F.softplus(fc(sd, 'decoder.net_semantic.0', x, lr_mul)), lr_mul)
    sem = squash(ys[:, 1:]) if sem_sigmoid else ys[:, 1:]
    return torch.cat([squash(y[:, 1:]), sem], 1).reshape(n, p, -1), ys[:, 0].reshape(n, p)


def _march(col, sig, z, white_back):
    seg = z[..., 1:] - z[..., :-1]
    dens = F.softplus((sig[..., :-1] + sig[..., 1:]) / 2 - 1)
    alpha = 1 - torch.exp(-dens * seg)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-10], -1), -1)[..., :-1]
    w = alpha * T
    rgb = (w[..., None] * (col[..., :-1, :] + col[..., 1:, :]) / 2).sum(-2)
    tot = w.sum(-1)
    depth = torch.nan_to_num((w * (z[..., :-1] + z[..., 1:]) / 2).sum(-1) / tot, float('inf')).clamp(z.min(), z.max())
    if white_back:
        rgb = rgb + 1 - tot[..., None]
    return rgb * 2 - 1, depth, w


def _importance(z, w, u):
    """Same sequential-cdf convention as render_oracle.sample_pdf, vectorised over rays."""
    r, s = z.shape
    wp = F.pad(w, (1, 1), value=float('-inf'))
    mp = torch.maximum(wp[:, :-1], wp[:, 1:])
    ap = (mp[:, :-1] + mp[:, 1:]) / 2 + 0.01
    bins = 0.5 * (z[:, :-1] + z[:, 1:])
    wk = ap[:, 1:-1] + 1e-5
    total = torch.zeros(r)
    for k in range(wk.shape[1]):
        total = total + wk[:, k]
    pdf = wk / total[:, None]
    cdf = torch.zeros(r, wk.shape[1] + 1)
    for k in range(wk.shape[1]):
        cdf[:, k + 1] = cdf[:, k] + pdf[:, k]
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo, hi = (inds - 1).clamp_min(0), inds.clamp_max(wk.shape[1])
    c0, c1, b0, b1 = cdf.gather(1, lo), cdf.gather(1, hi), bins.gather(1, lo), bins.gather(1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def render(sd, planes, o, d, opts, u_c, u_f, two_nets=True, sem_sigmoid=False, lr_mul=1.0, chunk=1 << 21):
    n, m, _ = o.shape
    sc, sf = opts['depth_resolution'], opts['depth_resolution_importance']
    step = torch.tensor((opts['ray_end'] - opts['ray_start']) / (sc - 1), dtype=torch.float32)
    z_c = torch.linspace(opts['ray_start'], opts['ray_end'], sc).reshape(1, 1, sc) + u_c.reshape(n, m, sc) * step

    def run(z):
        s = z.shape[-1]
        pts = (o[:, :, None, :] + z[..., None] * d[:, :, None, :]).reshape(n, m * s, 3)
        cols, sigs = [], []
        for a in range(0, m * s, chunk):
            c_, s_ = _decode(sd, planes, pts[:, a:a + chunk], opts['box_warp'], two_nets, sem_sigmoid, lr_mul)
            cols.append(c_); sigs.append(s_)
        return torch.cat(cols, 1).reshape(n, m, s, -1), torch.cat(sigs, 1).reshape(n, m, s)

    c_c, s_c = run(z_c)
    _, _, w_c = _march(c_c, s_c, z_c, opts.get('white_back', False))
    z_f = _importance(z_c.reshape(n * m, sc), w_c.reshape(n * m, sc - 1), u_f.reshape(n * m, sf)).reshape(n, m, sf)
    c_f, s_f = run(z_f)
    z_all, order = torch.sort(torch.cat([z_c, z_f], -1), dim=-1)
    c_all = torch.cat([c_c, c_f], 2).gather(2, order[..., None].expand(-1, -1, -1, c_c.shape[-1]))
    s_all = torch.cat([s_c, s_f], 2).gather(2, order)
    rgb, depth, w = _march(c_all, s_all, z_all, opts.get('white_back', False))
    return rgb, depth, w.sum(-1)


def synthesis(sd, cfg, ws, c, u_coarse, u_fine, nrr=128, noise_mode='const'):
    """TriPlaneSemanticEntangleGenerator.synthesis (triplane_cond.py:1020-1061) / TriPlaneGenerator.synthesis (:656-700).

    cfg: dict(rendering_kwargs=..., semantic_channels=int or None, sr_kind='8XDC'|'2X', sr_clamp=256|None, lr_mul=float)."""
    rk = cfg['rendering_kwargs']
    sem_ch = cfg.get('semantic_channels')
    o, d = ray_sampler(c[:, :16].reshape(-1, 4, 4), c[:, 16:25].reshape(-1, 3, 3), nrr)
    planes = backbone(sd, 'backbone.synthesis', ws, noise_mode=noise_mode)
    n = planes.shape[0]
    planes = planes.reshape(n, 3, 32, *planes.shape[-2:])
    feat, depth, _ = render(sd, planes, o, d, rk, u_coarse, u_fine, two_nets=sem_ch is not None, sem_sigmoid=(sem_ch == 1), lr_mul=cfg.get('lr_mul', 1.0))
    fimg = feat.permute(0, 2, 1).reshape(n, -1, nrr, nrr).contiguous()
    dimg = depth.reshape(n, 1, nrr, nrr)
    srkw = dict(kind=cfg['sr_kind'], noise_mode=rk['superresolution_noise_mode'], clamp=cfg.get('sr_clamp'),
                input_resolution=128 if cfg['sr_kind'] == '8XDC' else 64, antialias=rk.get('sr_antialias', True))
    if sem_ch is None:
        img, raw = superresolution(sd, 'superresolution', fimg[:, :3], fimg, ws, **srkw)
        return {'image': img, 'image_raw': raw, 'image_depth': dimg}
    half = fimg.shape[1] // 2
    rgbf, semf = fimg[:, :half], fimg[:, half:]
    img, raw = superresolution(sd, 'superresolution', rgbf[:, :3], rgbf, ws, **srkw)
    sem, sem_raw = superresolution(sd, 'superresolution_semantic', semf[:, :sem_ch], semf, ws, **srkw)
    return {'image': img, 'image_raw': raw, 'image_depth': dimg, 'semantic': sem, 'semantic_raw': sem_raw}
