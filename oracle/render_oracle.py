"""CPU oracle for the tri-plane renderer (ray sampler, stratified + importance sampling, plane taps, OSG
decoders, midpoint compositing).

TEST INFRASTRUCTURE ONLY — see oracle/ops_oracle.py for the rules.  Plain numpy (fp32 arrays, explicit
loops over samples where order matters) restating training/volumetric_rendering/*.py and the decoders of
training/triplane.py / training/triplane_cond.py of the reference; every function cites the lines it
follows.  Pinned by tests/test_oracle_golden.py against tests/golden/renderer_*.npz, which
tests/golden/make_golden.py records from the reference itself with its random draws captured.

One deliberate convention: reductions whose order the reference leaves to ATen (the pdf normaliser and
cdf in sample_pdf) are evaluated here as sequential fp32 sums.  That is within rounding of the reference
and makes the index work (searchsorted bins, merge order) reproducible bit for bit by the HIP kernel.
"""
import numpy as np

F32 = np.float32


def ray_sampler(cam2world, intrinsics, resolution):
    """ray_sampler.py:24-62.  cam2world [N,4,4], intrinsics [N,3,3] -> origins, directions [N, R*R, 3]."""
    c2w = np.asarray(cam2world, F32)
    K = np.asarray(intrinsics, F32)
    n, r = c2w.shape[0], int(resolution)
    fx, fy, cx, cy, sk = K[:, 0, 0, None], K[:, 1, 1, None], K[:, 0, 2, None], K[:, 1, 2, None], K[:, 0, 1, None]
    centre = (np.arange(r, dtype=F32) * F32(1. / r) + F32(0.5 / r)).astype(F32)
    col = np.tile(centre, r)[None]                 # x varies fastest (row-major pixels)
    row = np.repeat(centre, r)[None]
    x = ((col - cx + cy * sk / fy - sk * row / fy) / fx).astype(F32)
    y = ((row - cy) / fy).astype(F32)
    pts = np.stack([x, y, np.ones_like(x), np.ones_like(x)], -1)                # [N,M,4]
    world = np.einsum('nij,nmj->nmi', c2w, pts)[:, :, :3]
    o = c2w[:, None, :3, 3]
    d = world - o
    d = d / np.maximum(np.linalg.norm(d, axis=2, keepdims=True), 1e-12)
    return np.broadcast_to(o, d.shape).astype(F32).copy(), d.astype(F32)


def _linspace_f32(start, end, steps):
    """torch.linspace on fp32: symmetric evaluation from both ends."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    lo = (start + step * i.astype(F32)).astype(F32)
    hi = (end - step * (steps - 1 - i).astype(F32)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


def sample_stratified(u, ray_start, ray_end, disparity=False):
    """renderer.py:169-192.  u [N,M,S] uniforms (the reference's rand_like draw) -> depths [N,M,S].
    ray_start/ray_end: floats, or arrays [N,M] for the tensor-limits branch."""
    u = np.asarray(u, F32)
    s = u.shape[-1]
    if np.ndim(ray_start) > 0:                                          # math_utils.linspace (math_utils.py:101-118)
        a, b = np.asarray(ray_start, F32)[..., None], np.asarray(ray_end, F32)[..., None]
        t = (np.arange(s, dtype=F32) / F32(s - 1)).astype(F32)
        return (a + t * (b - a) + u * ((b - a) / F32(s - 1))).astype(F32)
    if disparity:
        t = _linspace_f32(0, 1, s) + u * F32(1 / (s - 1))
        return (F32(1.) / (F32(1.) / F32(ray_start) * (F32(1.) - t) + F32(1.) / F32(ray_end) * t)).astype(F32)
    return (_linspace_f32(ray_start, ray_end, s) + u * F32((ray_end - ray_start) / (s - 1))).astype(F32)


# inverse plane bases of generate_planes() (renderer.py:23-53): which world axes feed grid (x, y) of each plane
_PLANE_AXES = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[1, 0, 0], [0, 0, 1], [0, 1, 0]], [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], np.float64)


def project_onto_planes(coords):
    """renderer.py:39-53: coords [N,M,3] -> [N,3,M,2] using inv(plane_axes), not a hard-coded permutation."""
    inv = np.linalg.inv(_PLANE_AXES)
    return np.einsum('nmc,kcd->nkmd', np.asarray(coords, np.float64), inv)[..., :2].astype(F32)


def grid_sample_bilinear_zeros(img, gx, gy):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) for one image [C,H,W] at P points."""
    c, h, w = img.shape
    ix = ((gx + 1) * w - 1) / 2
    iy = ((gy + 1) * h - 1) / 2
    x0, y0 = np.floor(ix), np.floor(iy)
    out = np.zeros([gx.shape[0], c], F32)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
        xi, yi = x0 + dx, y0 + dy
        wgt = (1 - np.abs(ix - xi)) * (1 - np.abs(iy - yi))
        ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        xs, ys = np.clip(xi, 0, w - 1).astype(np.int64), np.clip(yi, 0, h - 1).astype(np.int64)
        out += np.where(ok, wgt, 0)[:, None].astype(F32) * img[:, ys, xs].T
    return out


def sample_from_planes(planes, coords, box_warp):
    """renderer.py:55-65: planes [N,3,C,H,W], coords [N,P,3] -> [N,3,P,C]."""
    planes = np.asarray(planes, F32)
    uv = project_onto_planes((2 / box_warp) * np.asarray(coords, F32))
    n, k = planes.shape[:2]
    return np.stack([np.stack([grid_sample_bilinear_zeros(planes[i, p], uv[i, p, :, 0], uv[i, p, :, 1]) for p in range(k)]) for i in range(n)])


def _fc(x, w, b, lr_mul):
    """FullyConnectedLayer, linear activation (networks_stylegan2.py:114-127)."""
    w = np.asarray(w, F32) * F32(lr_mul / np.sqrt(w.shape[1]))
    return x @ w.T + np.asarray(b, F32) * F32(lr_mul)


def _softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(F32)


def _squash(x):
    return (1 / (1 + np.exp(-x)) * (1 + 2 * 0.001) - 0.001).astype(F32)


def decode(features, dec):
    """features [N,3,P,32] -> colours [N,P,C], sigma [N,P].

    dec: dict with 'w1','b1','w2','b2' (+ 'w1s','b1s','w2s','b2s' and 'semantic_sigmoid' for the two-net
    decoder) and 'lr_mul'.  One net: OSGDecoder (training/triplane.py:112-135).  Two nets:
    OSGDecoder_semantic_lateSeparate (training/triplane_cond.py:926-970) — density from the label net."""
    x = np.asarray(features, F32).mean(1)
    lr = dec.get('lr_mul', 1.0)
    y = _fc(_softplus(_fc(x, dec['w1'], dec['b1'], lr)), dec['w2'], dec['b2'], lr)
    if 'w1s' not in dec:
        return _squash(y[..., 1:]), y[..., 0]
    ys = _fc(_softplus(_fc(x, dec['w1s'], dec['b1s'], lr)), dec['w2s'], dec['b2s'], lr)
    sem = _squash(ys[..., 1:]) if dec.get('semantic_sigmoid', False) else ys[..., 1:]
    return np.concatenate([_squash(y[..., 1:]), sem], -1).astype(F32), ys[..., 0]


def ray_march(colors, sigmas, depths, white_back=False, clamp_range=None):
    """ray_marcher.py:25-57 with an explicit front-to-back loop.  colors [R,S,C], sigmas [R,S], depths [R,S] ->
    rgb [R,C], depth [R], weights [R,S-1].  The depth clamp uses min/max of ``depths`` (the whole tensor) unless
    ``clamp_range`` is given."""
    colors, sigmas, depths = np.asarray(colors, F32), np.asarray(sigmas, F32), np.asarray(depths, F32)
    r, s, c = colors.shape
    T = np.ones([r], F32)
    rgb, wsum, wz = np.zeros([r, c], F32), np.zeros([r], F32), np.zeros([r], F32)
    weights = np.zeros([r, s - 1], F32)
    for i in range(s - 1):
        delta = depths[:, i + 1] - depths[:, i]
        dens = _softplus((sigmas[:, i] + sigmas[:, i + 1]) / 2 - 1)
        alpha = (1 - np.exp(-(dens * delta))).astype(F32)
        w = alpha * T
        T = (T * (1 - alpha + F32(1e-10))).astype(F32)
        weights[:, i] = w
        rgb += w[:, None] * ((colors[:, i] + colors[:, i + 1]) / 2)
        wsum += w
        wz += w * ((depths[:, i] + depths[:, i + 1]) / 2)
    with np.errstate(invalid='ignore', divide='ignore'):
        d = wz / wsum
    d = np.where(np.isnan(d), np.inf, d)
    lo, hi = (depths.min(), depths.max()) if clamp_range is None else clamp_range
    d = np.clip(d, lo, hi)
    if white_back:
        rgb = rgb + 1 - wsum[:, None]
    return (rgb * 2 - 1).astype(F32), d.astype(F32), weights


def ray_march_backward(colors, sigmas, depths, g_rgb, g_wsum=None, white_back=False):
    """Gradient of ``ray_march`` with respect to the per-sample colours and densities, for upstream gradients ``g_rgb`` [R,C] on the
    returned rgb (= rgb_raw * 2 - 1) and ``g_wsum`` [R] on sum(weights) — what autograd derives for ray_marcher.py:25-57 — written
    as the two explicit sweeps csrc/render_bwd.hip runs (fp64 here).  Depths are constants (renderer.py:198, 211).

    Forward:  w_i = a_i T_i,  T_i = prod_{j<i} (1 - a_j + 1e-10),  a_i = 1 - exp(-softplus(sm_i - 1) d_i),  sm_i = (s_i + s_{i+1}) / 2,
              rgb = 2 (sum_i w_i (c_i + c_{i+1}) / 2 [+ 1 - sum_i w_i]) - 1.
    Returns (d_colors [R,S,C], d_sigmas [R,S], colour_weight [R,S]) with d_colors = 2 g_rgb * colour_weight."""
    colors, sigmas, depths = np.asarray(colors, np.float64), np.asarray(sigmas, np.float64), np.asarray(depths, np.float64)
    g_rgb = np.asarray(g_rgb, np.float64)
    r, s, c = colors.shape
    dC = 2.0 * g_rgb                                                    # dL/d(composited colour)
    base = (0.0 if g_wsum is None else np.asarray(g_wsum, np.float64)) - (dC.sum(1) if white_back else 0.0)
    alpha, T, w, sm = (np.zeros([r, s - 1]) for _ in range(4))
    t = np.ones([r])
    for i in range(s - 1):                                              # forward sweep: what the tape records
        sm[:, i] = (sigmas[:, i] + sigmas[:, i + 1]) / 2
        x = sm[:, i] - 1
        dens = np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))
        alpha[:, i] = 1 - np.exp(-dens * (depths[:, i + 1] - depths[:, i]))
        T[:, i] = t
        w[:, i] = alpha[:, i] * t
        t = t * (1 - alpha[:, i] + 1e-10)
    dw = np.einsum('rc,rsc->rs', dC, (colors[:, :-1] + colors[:, 1:]) / 2) + np.reshape(base, [-1, 1] if np.ndim(base) else [])
    dsm = np.zeros([r, s - 1])
    suffix = np.zeros([r])
    for i in range(s - 2, -1, -1):                                      # back to front: sum_{j>i} dw_j w_j
        dalpha = dw[:, i] * T[:, i] - suffix / (1 - alpha[:, i] + 1e-10)
        suffix = suffix + dw[:, i] * w[:, i]
        x = sm[:, i] - 1
        sg = np.where(x > 20, 1.0, 1 / (1 + np.exp(-x)))
        dsm[:, i] = dalpha * (depths[:, i + 1] - depths[:, i]) * (1 - alpha[:, i]) * sg
    d_sig = np.zeros([r, s])
    d_sig[:, :-1] += dsm / 2
    d_sig[:, 1:] += dsm / 2
    cw = np.zeros([r, s])
    cw[:, :-1] += w / 2
    cw[:, 1:] += w / 2
    return dC[:, None, :] * cw[:, :, None], d_sig, cw


def importance_bins(z, weights):
    """First half of sample_importance (renderer.py:194-212): smoothed weights and bin positions.
    z [R,S], weights [R,S-1] -> bins [R,S-1], w_pdf [R,S-3]."""
    z, w = np.asarray(z, F32), np.asarray(weights, F32)
    pad = np.full([w.shape[0], 1], -np.inf, F32)
    wp = np.concatenate([pad, w, pad], 1)
    mp = np.maximum(wp[:, :-1], wp[:, 1:])                       # max_pool1d(2, 1, padding=1): S values
    ap = ((mp[:, :-1] + mp[:, 1:]) / F32(2)).astype(F32)         # avg_pool1d(2, 1): S-1 values
    ap = (ap + F32(0.01)).astype(F32)
    bins = (F32(0.5) * (z[:, :-1] + z[:, 1:])).astype(F32)
    return bins, ap[:, 1:-1]


def sample_pdf(bins, weights, u, eps=1e-5, return_index=False):
    """renderer.py:214-253 with sequential fp32 sums.  bins [R,B+1], weights [R,B], u [R,K] -> samples [R,K]."""
    bins, u = np.asarray(bins, F32), np.asarray(u, F32)
    w = (np.asarray(weights, F32) + F32(eps)).astype(F32)
    r, b = w.shape
    total = np.zeros([r], F32)
    for k in range(b):
        total = (total + w[:, k]).astype(F32)
    pdf = (w / total[:, None]).astype(F32)
    cdf = np.zeros([r, b + 1], F32)
    for k in range(b):
        cdf[:, k + 1] = (cdf[:, k] + pdf[:, k]).astype(F32)
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)            # searchsorted(cdf, u, right=True)
    below, above = np.maximum(inds - 1, 0), np.minimum(inds, b)
    rows = np.arange(r)[:, None]
    c0, c1, b0, b1 = cdf[rows, below], cdf[rows, above], bins[rows, below], bins[rows, above]
    denom = (c1 - c0).astype(F32)
    denom = np.where(denom < F32(eps), F32(1), denom)
    t = ((u - c0).astype(F32) / denom).astype(F32)
    out = (b0 + (t * (b1 - b0).astype(F32)).astype(F32)).astype(F32)
    return (out, inds) if return_index else out


def sample_importance(z, weights, u):
    """renderer.py:194-212: z [R,S], weights [R,S-1], u [R,K] -> fine depths [R,K] in draw order."""
    bins, w = importance_bins(z, weights)
    return sample_pdf(bins, w, u)


def render(planes, dec, ray_o, ray_d, opts, u_coarse, u_fine, t_start=None, t_end=None, details=False, point_fn=None):
    """ImportanceRenderer.forward (renderer.py:88-140).  planes [N,3,32,H,W]; ray_o, ray_d [N,M,3];
    u_coarse [N,M,Sc]; u_fine [N*M,Sf] -> feat [N,M,C], depth [N,M], wsum [N,M].  ``point_fn(pts [N,P,3]) -> (colours [N,P,C], sigma
    [N,P])`` replaces the plane lookup + decoder (render_semantic)."""
    planes = None if planes is None else np.asarray(planes, F32)
    o, d = np.asarray(ray_o, F32), np.asarray(ray_d, F32)
    n, m, _ = o.shape
    if t_start is not None:
        z_c = sample_stratified(np.asarray(u_coarse, F32).reshape(n, m, -1), np.asarray(t_start, F32).reshape(n, m), np.asarray(t_end, F32).reshape(n, m))
    else:
        z_c = sample_stratified(np.asarray(u_coarse, F32).reshape(n, m, -1), opts['ray_start'], opts['ray_end'], opts.get('disparity_space_sampling', False))
    sc = z_c.shape[-1]

    def run(z):
        s = z.shape[-1]
        pts = (o[:, :, None, :] + z[..., None] * d[:, :, None, :]).reshape(n, m * s, 3)
        col, sig = point_fn(pts) if point_fn is not None else decode(sample_from_planes(planes, pts, opts['box_warp']), dec)
        return col.reshape(n * m, s, -1), sig.reshape(n * m, s)

    c_c, s_c = run(z_c)
    zc = z_c.reshape(n * m, sc)
    sf = opts.get('depth_resolution_importance', 0)
    if sf > 0:
        _, _, w_c = ray_march(c_c, s_c, zc, opts.get('white_back', False))
        z_f = sample_importance(zc, w_c, np.asarray(u_fine, F32).reshape(n * m, sf))
        c_f, s_f = run(z_f.reshape(n, m, sf))
        z_all = np.concatenate([zc, z_f], 1)
        order = np.argsort(z_all, axis=1, kind='stable')          # unify_samples (renderer.py:157-167)
        rows = np.arange(n * m)[:, None]
        z_s = z_all[rows, order]
        c_s = np.concatenate([c_c, c_f], 1)[rows, order]
        s_s = np.concatenate([s_c, s_f], 1)[rows, order]
        rgb, depth, w = ray_march(c_s, s_s, z_s, opts.get('white_back', False))
    else:
        w_c, z_f, z_s = None, None, zc
        rgb, depth, w = ray_march(c_c, s_c, zc, opts.get('white_back', False))
    out = rgb.reshape(n, m, -1), depth.reshape(n, m), w.sum(1).reshape(n, m)
    if details:
        return out + (dict(z_coarse=zc, w_coarse=w_c, z_fine=z_f, z_all=z_s, colors=(c_s if sf > 0 else c_c), sigmas=(s_s if sf > 0 else s_c)),)
    return out


def run_model(planes, dec, coords, box_warp):
    """ImportanceRenderer.run_model (renderer.py:142-148): coords [N,P,3] -> rgb [N,P,C], sigma [N,P]."""
    return decode(sample_from_planes(np.asarray(planes, F32), coords, box_warp), dec)


def run_model_semantic(planes_t, planes_s, dec_t, dec_s, coords, box_warp):
    """ImportanceSemanticRenderer.run_model (renderer.py:324-333): the label decoder (OSGDecoder_semantic, triplane_cond.py:859-887) sees
    the semantic planes and gives density + labels; the colour decoder (OSGDecoder on 64 features, triplane.py:112-135) sees
    cat(texture, semantic).  dec_t / dec_s: dicts 'w1','b1','w2','b2','lr_mul'; dec_s['sigmoid'] squashes the labels.
    -> rgb [N,P,32], sigma [N,P], semantic [N,P,32]."""
    ft = sample_from_planes(planes_t, coords, box_warp)
    fs = sample_from_planes(planes_s, coords, box_warp)
    xs = fs.mean(1)
    ys = _fc(_softplus(_fc(xs, dec_s['w1'], dec_s['b1'], dec_s['lr_mul'])), dec_s['w2'], dec_s['b2'], dec_s['lr_mul'])
    sem = _squash(ys[..., 1:]) if dec_s.get('sigmoid', False) else ys[..., 1:].astype(F32)
    xt = np.concatenate([ft, fs], -1).mean(1)
    yt = _fc(_softplus(_fc(xt, dec_t['w1'], dec_t['b1'], dec_t['lr_mul'])), dec_t['w2'], dec_t['b2'], dec_t['lr_mul'])
    return _squash(yt[..., 1:]), ys[..., 0], sem


def render_semantic(planes_t, planes_s, dec_t, dec_s, ray_o, ray_d, opts, u_coarse, u_fine):
    """ImportanceSemanticRenderer.forward (renderer.py:262-322): same sampling and compositing as ``render`` over the feature vector
    cat(colour, label) -> feat [N,M,64], depth [N,M], wsum [N,M]."""
    def point_fn(pts):
        rgb, sig, sem = run_model_semantic(planes_t, planes_s, dec_t, dec_s, pts, opts['box_warp'])
        return np.concatenate([rgb, sem], -1).astype(F32), sig
    return render(None, None, ray_o, ray_d, opts, u_coarse, u_fine, point_fn=point_fn)
