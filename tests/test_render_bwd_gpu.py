"""Fused renderer backward (csrc/render_bwd.hip) against autograd through the tensor-op renderer on the same draws: gradients to the
planes and to every decoder parameter, for the four recorded renderer configurations (two-net seg2cat decoder, sigmoid labels +
white background, single-net OSG decoder, per-ray 'auto' limits).  Tolerance 2e-3 relative to the largest gradient entry: the two
paths differ by fp32 summation order over up to 10^5 atomic contributions per texel / weight."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from render_cases import CASES, load_case, make_decoder

pytestmark = pytest.mark.gpu


def _setup(name, seed=0):
    from pix2pix3d_amd.training.volumetric_rendering import renderer as R
    g, opts, _ = load_case(name)
    torch.manual_seed(seed)
    dec = make_decoder(g, 'cuda').requires_grad_(True)
    planes = torch.tensor(g['planes'], device='cuda')
    ro, rd = torch.tensor(g['ray_o'], device='cuda'), torch.tensor(g['ray_d'], device='cuda')
    return R, dec, opts, planes, ro, rd


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('with_wsum', [False, True])
def test_fused_backward_matches_tensor_op_autograd(hip_lib, name, with_wsum):
    R, dec, opts, planes, ro, rd = _setup(name)
    renderer = R.ImportanceRenderer().cuda()
    nch = 32 * len(R._decoder_nets(dec)[0])
    n, m = ro.shape[0], ro.shape[1]
    torch.manual_seed(5)
    g_feat = torch.randn(n, m, nch, device='cuda')
    g_w = torch.randn(n, m, 1, device='cuda') if with_wsum else None

    def run(fused_bwd):
        R.fused_backward = fused_bwd
        pl = planes.clone().requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        torch.manual_seed(11)                           # identical uniform draws in both runs
        feat, depth, wsum = renderer(pl, dec, ro, rd, opts)
        loss = (feat * g_feat).sum() + ((wsum * g_w).sum() if with_wsum else 0.0)
        loss.backward()
        return pl.grad.clone(), [p.grad.clone() for p in dec.parameters()], feat.detach()

    from pix2pix3d_amd import _lib
    try:
        c0 = _lib.launch_count('render')
        gp_f, gd_f, feat_f = run(True)
        assert _lib.launch_count('render') == c0 + 2, 'expected one fused forward + one fused backward call'
        gp_r, gd_r, feat_r = run(False)
        assert _lib.launch_count('render') == c0 + 3, 'the reference run must take the fused forward only'
    finally:
        R.fused_backward = True
    assert rel_err(feat_f.cpu().numpy(), feat_r.cpu().numpy()) < 1e-6          # same forward either way
    assert gp_f.shape == gp_r.shape
    assert float(gp_r.abs().max()) > 0
    assert rel_err(gp_f.cpu().numpy(), gp_r.cpu().numpy()) < 2e-3, 'plane gradients'
    names = [k for k, _ in dec.named_parameters()]
    for k, a, b in zip(names, gd_f, gd_r):
        assert a.shape == b.shape
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-3, f'decoder gradient {k}'


def test_bench_sized_backward_properties(hip_lib):
    """BASELINE training size (2 images x 128^2 rays x 48+48 samples, 256^2 planes): properties of the fused backward that need no
    oracle — (1) linearity in the upstream gradient, (2) the directional derivative along a random plane / decoder perturbation,
    measured by central differences of the fused FORWARD with the loss accumulated in fp64, (3) zero upstream gradient gives zero
    gradients."""
    from pix2pix3d_amd.training.volumetric_rendering import renderer as R
    from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    g, opts, _ = load_case('seg')
    dec = make_decoder(g, 'cuda').requires_grad_(True)
    torch.manual_seed(2)
    n, res, sc, sf = 2, 128, 48, 48
    op = dict(opts, depth_resolution=sc, depth_resolution_importance=sf)
    planes = torch.randn(n, 3, 32, 256, 256, device='cuda') * 0.5
    c2w = torch.tensor(g['c2w'][:n], device='cuda')
    K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]], device='cuda').repeat(n, 1, 1)
    o, d = RaySampler()(c2w, K, res)
    uc = torch.rand(n, res * res, sc, device='cuda')
    uf = torch.rand(n * res * res, sf, device='cuda')
    g1 = torch.randn(n, res * res, 64, device='cuda')
    g2 = torch.randn(n, res * res, 64, device='cuda')

    def bwd(gf):
        gp, gd = R.fused_render_backward(planes, dec, o, d, op, uc, uf, None, None, gf)
        return gp, [x.clone() for x in gd]

    gp1, gd1 = bwd(g1)
    gp2, gd2 = bwd(g2)
    gp12, gd12 = bwd(g1 + g2)
    scale = float(gp12.abs().max())
    assert scale > 0 and torch.isfinite(gp12).all()
    assert float((gp1 + gp2 - gp12).abs().max()) < 1e-4 * scale                          # (1) linear (atomic summation order only)
    for a, b, c in zip(gd1, gd2, gd12):
        assert float((a + b - c).abs().max()) < 1e-4 * float(c.abs().max())
    gp0, gd0 = bwd(torch.zeros_like(g1))
    assert float(gp0.abs().max()) == 0.0 and all(float(x.abs().max()) == 0.0 for x in gd0)   # (3)

    # (2) directional derivative: L(theta) = sum(feat * g1) in fp64; dL/deps at eps = 0 along v  ==  <grad, v>.
    # The colour net does not influence where the samples sit (densities come from the label net), so along its parameters the
    # central difference of the forward IS the derivative the backward computes.  (Along the planes the forward would also move its
    # importance samples, which the gradient deliberately treats as constants, renderer.py:198, 211 — see the last check instead.)
    def loss(pl):
        feat, _, _ = R.fused_render(pl, dec, o, d, op, uc, uf, exact_fp32=True)      # the forward the backward differentiates (the training forward; the
        return float((feat.double() * g1.double()).sum())                             # inference default, a bf16x3 decoder, adds 1e-6-level noise to a difference quotient)
    names = [k for k, _ in dec.named_parameters()]
    colour = [i for i, k in enumerate(names) if k.startswith('net.')]
    assert len(colour) == 4
    params = list(dec.parameters())
    with torch.no_grad():
        dirs = {i: torch.randn_like(params[i]) for i in colour}
        an_w = sum(float((gd1[i].double() * dirs[i].double()).sum()) for i in colour)
        epsw = 8e-3                                   # (2e-3 left the quotient at the mercy of the forward's fp32 rounding: ~1e-4 of noise on L / 4e-3)
        for i in colour:
            params[i].add_(epsw * dirs[i])
        lp = loss(planes)
        for i in colour:
            params[i].sub_(2 * epsw * dirs[i])
        lm = loss(planes)
        for i in colour:
            params[i].add_(epsw * dirs[i])
    fd_w = (lp - lm) / (2 * epsw)
    assert abs(fd_w - an_w) < 1e-2 * max(abs(an_w), abs(fd_w)), (fd_w, an_w)
    # plane gradients at this size: against autograd through the tensor-op renderer (8.6 GB of saved tensors) on the same draws
    renderer = R.ImportanceRenderer().cuda()
    pl = planes.clone().requires_grad_(True)
    with torch.enable_grad(), R._replay_draws(uc.reshape(n, res * res, sc, 1), uf):
        feat, _, _ = renderer._forward_tensor_ops(pl, dec, o, d, op)
        ref, = torch.autograd.grad((feat * g1).sum(), [pl])
    assert rel_err(gp1.cpu().numpy(), ref.cpu().numpy()) < 2e-3


@pytest.mark.parametrize('name', CASES)
def test_tape_matches_the_compositing_backward_oracle(hip_lib, name):
    """What the tape sweep hands each sample (depth, colour weight, dL/dsigma) against oracle.render_oracle: the oracle renders the same
    rays, keeps its sorted per-sample colours / densities, and ray_march_backward (pinned to autograd on the CPU) differentiates them."""
    from oracle import render_oracle as RO
    R, dec, opts, planes, ro, rd = _setup(name)
    g, _, dec_np = load_case(name)
    n, m = ro.shape[0], ro.shape[1]
    nch = 32 * len(R._decoder_nets(dec)[0])
    sc, sf = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
    u_c, u_f = torch.tensor(g['u_coarse'], device='cuda'), torch.tensor(g['u_fine'], device='cuda')
    torch.manual_seed(9)
    g_feat = torch.randn(n, m, nch, device='cuda')
    g_w = torch.randn(n, m, 1, device='cuda')
    t0 = t1 = None
    if opts['ray_start'] == 'auto':
        t0, t1 = R.ImportanceRenderer()._ray_limits(ro, rd, opts)
    _, _, tape = R.fused_render_backward(planes, dec, ro, rd, opts, u_c, u_f, t0, t1, g_feat, g_w, debug=True)
    tape = tape.cpu().numpy()
    _, _, _, det = RO.render(g['planes'], dec_np, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'],
                             t_start=None if t0 is None else t0.cpu().numpy(), t_end=None if t1 is None else t1.cpu().numpy(), details=True)
    _, d_sig, cw = RO.ray_march_backward(det['colors'], det['sigmas'], det['z_all'], g_feat.reshape(n * m, nch).cpu().numpy(),
                                         g_w.reshape(-1).cpu().numpy(), white_back=bool(opts.get('white_back', False)))
    assert tape.shape == (n * m, sc + sf, 4)
    assert np.abs(tape[..., 0] - det['z_all']).max() < 2e-6
    assert rel_err(tape[..., 1], cw) < 2e-4
    assert rel_err(tape[..., 2], d_sig) < 2e-3


def test_double_backward_through_the_fused_point_queries_raises(hip_lib):
    """The fused backwards call device kernels autograd cannot see through: a ``create_graph=True`` gradient (a path-length or R1-style penalty on
    G.sample_mixed) must raise on the second differentiation instead of silently contributing zeros."""
    R, dec, opts, planes, _, _ = _setup('seg')
    planes = planes.clone().requires_grad_(True)
    torch.manual_seed(3)
    xyz = (torch.rand(planes.shape[0], 256, 3, device='cuda') - 0.5) * float(opts.get('box_warp', 1))
    rgb, sigma = R._FusedPointsFn.apply(dec, opts, xyz, planes, *dec.parameters())
    g_planes, = torch.autograd.grad(sigma.sum(), [planes], create_graph=True)           # only sigma carries a gradient (g_rgb None), two nets
    assert g_planes.shape == planes.shape and float(g_planes.abs().max()) > 0
    with pytest.raises(RuntimeError, match='once_differentiable|differentiated twice|does not require grad'):
        g_planes.square().sum().backward()
