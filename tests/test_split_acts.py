"""Activations that stay split between the bf16x3 layers of an inference pass (modconv.SplitActs; csrc/conv2d.hip XS / y_split,
p3d_fir4_bias_act_nhwc_split).  The format is an internal hand-over between two modulated_conv2d calls of the reference
(training/networks_stylegan2.py:436-459); what is pinned here: the kernels that read / write it produce the SAME BITS as the plain-tensor
calls (or, on the ring pipeline, sum the same products in another order), and the generator's outputs do not move when it is switched on."""
import numpy as np
import pytest
import torch


def _split_storage(v):
    """Reference packing on the CPU: fp32 [N,C,H,W] -> fp32-typed channels-last storage of [32 x bf16 hi | 32 x bf16 lo] rows."""
    n, c, h, w = v.shape
    x = v.permute(0, 2, 3, 1).reshape(n, h, w, c // 32, 32)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    rows = torch.stack([hi, lo], dim=-2).reshape(n, h, w, c // 32, 64).view(torch.float32).reshape(n, h, w, c)
    return rows.permute(0, 3, 1, 2)


def test_dense_of_a_split_tensor_resplits_to_the_same_halves():
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(0)
    v = (torch.randn(2, 64, 5, 7) * torch.logspace(-3, 3, 64).view(1, 64, 1, 1)).contiguous(memory_format=torch.channels_last)
    s = modconv.SplitActs(_split_storage(v).contiguous(memory_format=torch.channels_last))
    d = s.dense()
    assert d.shape == v.shape and s.shape == v.shape and s.dtype == torch.float32 and s.is_contiguous(memory_format=torch.channels_last)
    assert float((d - v).abs().max() / v.abs().max()) < 2 ** -16
    # a bf16x3 consumer of dense() splits it again into a pair that stands for the same number (to fp32 rounding: the pair itself may differ at a tie)
    def halves(storage):
        n, c, h, w = storage.shape
        r = storage.permute(0, 2, 3, 1).reshape(n, h, w, c // 32, 32).view(torch.bfloat16).reshape(n, h, w, c // 32, 2, 32).double()
        return r[..., 0, :], r[..., 1, :]
    hi0, lo0 = halves(s.t)
    hi1, lo1 = halves(_split_storage(d.contiguous(memory_format=torch.channels_last)).contiguous(memory_format=torch.channels_last))
    assert float((((hi1 + lo1) - (hi0 + lo0)).abs() / hi0.abs().clamp_min(1e-30)).max()) < 2 ** -22
    assert not modconv.accepts_split_input(4, 512, 32 * 32, 1) and modconv.accepts_split_input(4, 512, 64 * 64, 1) and modconv.accepts_split_input(4, 512, 64 * 64, 2)
    assert modconv.accepts_split_input(1, 512, 32 * 32, 1) and not modconv.accepts_split_input(4, 48, 64 * 64, 1)


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.gpu
@pytest.mark.parametrize('ci,co,h,w,n', [(128, 128, 128, 64, 4), (64, 96, 70, 52, 2), (256, 64, 72, 90, 3),            # halo-slab kernel / split-K generic kernel
                                         (128, 128, 128, 128, 4), (64, 128, 100, 90, 5), (32, 256, 64, 64, 8)])      # the ring kernel: full tiles, ragged tiles, two channel blocks x one chunk pair
def test_kernels_read_and_write_the_split_layout(hip_lib, ci, co, h, w, n):
    """Kernels on split input vs the same layer on the plain tensor: bit-identical where the same pipeline runs (halo-slab, generic: same
    operands into the same MFMAs in the same order); the ring kernel (conv3x3_r2_bf16x3_kernel) sums the same products in another order:
    <= 2e-6 of the range.  A split RESULT is bit-for-bit the split of the plain result of the same kernel."""
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(ci + h)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    xs = modconv.SplitActs(_nhwc(_split_storage(x.cpu())).cuda())
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(n, ci, device='cuda') + 1
    w3 = modconv.modulate_weights(weight, styles, dtype=modconv.BF16X3)
    bias, noise, ns = torch.randn(co, device='cuda'), torch.randn(h, w, device='cuda'), torch.tensor(0.3, device='cuda')
    kw = dict(bias=bias, noise=noise, noise_strength=ns, act=1, gain=2 ** 0.5, clamp=256.0, split=True)
    ring = co % 128 == 0 and h >= 32 and w >= 32 and ((h + 15) // 16) * ((w + 15) // 16) * (co // 128) * n >= 192
    y_plain = modconv.conv2d(x, w3, **kw)
    y_in = modconv.conv2d(xs, w3, **kw)                                    # reads the split layout
    assert isinstance(y_in, torch.Tensor)
    if ring:
        e = float((y_in - y_plain).abs().max() / y_plain.abs().max())
        print((ci, co, h, w, n), 'ring kernel vs halo kernel', e)
        assert e < 2e-6
    else:
        assert torch.equal(y_in, y_plain)
    y_io = modconv.conv2d(xs, w3, out_split=True, **kw)                    # ... and writes it
    if isinstance(y_io, modconv.SplitActs):                                # granted by the halo-slab / ring kernels (launches whose own grid fills the chip)
        assert torch.equal(y_io.t.cpu().view(torch.int32), _nhwc(_split_storage(y_in.cpu())).view(torch.int32))
    else:                                                                  # split-K route: a plain tensor comes back
        assert not ring and (ci, co, h) != (128, 128, 128) and torch.equal(y_io, y_plain)
    yt_plain, yt_in = modconv.conv2d(x, w3, transposed=True, split=True), modconv.conv2d(xs, w3, transposed=True, split=True)      # generic kernel, four parity classes
    assert torch.equal(yt_in, yt_plain)
    w1 = modconv.modulate_weights(torch.randn(96, ci, 1, 1, device='cuda'), styles, demodulate=False, dtype=modconv.BF16X3)
    assert torch.equal(modconv.conv2d(xs, w1, bias=torch.zeros(96, device='cuda'), split=True), modconv.conv2d(x, w1, bias=torch.zeros(96, device='cuda'), split=True))   # 1x1 (ToRGB)
    if co % 32 == 0:                                                       # the x2 layer's FIR + epilogue writing the split layout
        f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
        nz2 = torch.randn(2 * h, 2 * w, device='cuda')
        a = modconv.fir4_bias_act(yt_plain, f, bias, nz2, ns, 'lrelu', 2 ** 0.5, 256.0)
        b = modconv.fir4_bias_act(yt_plain, f, bias, nz2, ns, 'lrelu', 2 ** 0.5, 256.0, out_split=True)
        assert isinstance(b, modconv.SplitActs) and torch.equal(b.t.cpu().view(torch.int32), _nhwc(_split_storage(a.cpu())).view(torch.int32))
    # the ring kernel against fp64 on its own (the bar of the bf16x3 formulation: 1e-5 of the range)
    if ring:
        import torch.nn.functional as F
        w32 = modconv.modulate_weights(weight, styles, dtype=torch.float32)
        wq = w32.double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3).cpu()
        ref = torch.stack([F.conv2d(x[i:i + 1].double().cpu(), wq[i], padding=1)[0] for i in range(n)])
        ref = (F.leaky_relu(ref + (noise * ns).double().cpu() + bias.double().cpu().view(1, -1, 1, 1), 0.2) * 2 ** 0.5).clamp(-256, 256)
        e64 = float((y_in.double().cpu() - ref).abs().max() / ref.abs().max())
        print('ring kernel vs fp64', e64)
        assert e64 < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('ci,co,h,w,n,skip', [(128, 96, 64, 64, 4, True), (256, 96, 34, 96, 2, True), (128, 64, 40, 32, 3, False), (256, 32, 16, 64, 1, True)])
def test_wide_torgb_with_the_skip_image_in_one_launch(hip_lib, ci, co, h, w, n, skip):
    """csrc/torgb_split.hip: ToRGB of a SplitActs + upsample2d(prev) in one pass against the two-launch form (ToRGB on the generic kernel, then
    p3d_upfirdn2d_acc) and against fp64: the products are the same (another summation order: <= 1e-6 of the range); the skip term is bit-for-bit
    the upsampling kernel's."""
    import torch.nn.functional as F
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(ci + co + h)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    xs = modconv.SplitActs(_nhwc(_split_storage(x.cpu())).cuda())
    weight = torch.randn(co, ci, 1, 1, device='cuda'); styles = (torch.randn(n, ci, device='cuda') + 1) / ci ** 0.5; bias = torch.randn(co, device='cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
    prev = _nhwc(torch.randn(n, co, h // 2, w // 2, device='cuda')) if skip else None
    with torch.no_grad():
        assert modconv.torgb_wide_skip_supported(xs, weight, prev, f)
        got = modconv.torgb_wide_skip(xs, weight, styles, bias, 256.0, prev, f)
        assert got.shape == (n, co, h, w) and got.is_contiguous(memory_format=torch.channels_last)
        y = modconv.torgb(xs, weight, styles, bias, clamp=256.0)                               # generic 1x1 kernel on the same split input
        two = upfirdn2d.upsample2d_add_(y.clone(memory_format=torch.channels_last), prev, f.detach()) if skip else y
    e2 = float((got - two).abs().max() / two.abs().max())
    ref = (torch.einsum('oc,nc,nchw->nohw', weight.reshape(co, ci).double().cpu(), styles.double().cpu(), xs.dense().double().cpu())
           + bias.double().cpu().view(1, co, 1, 1)).clamp(-256, 256)
    if skip:
        ref = ref + upfirdn2d.upsample2d(prev.double().cpu().contiguous(), f.cpu())
    e64 = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    print((ci, co, h, w, n, skip), 'vs two launches', e2, 'vs fp64', e64)
    assert e2 < 1e-6 and e64 < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('ci,rco,h,w,n,skip,noise,act,clamp', [
    (128, 96, 256, 256, 1, True, True, 1, 256.0),        # the backbone's last block at its own size (one image)
    (128, 96, 128, 128, 4, True, True, 1, 256.0),
    (64, 96, 100, 90, 5, True, True, 1, 0.7),            # ragged patches on both axes, one chunk pair, the layer's clamp active
    (256, 64, 72, 96, 8, False, False, 0, -1.0),         # no skip image, no noise, linear
    (96, 32, 64, 64, 12, True, False, 1, -1.0),
])
def test_last_conv_wide_torgb_and_skip_image_in_one_launch(hip_lib, ci, rco, h, w, n, skip, noise, act, clamp):
    """conv3x3_r2_bf16x3_kernel<TR> (p3d_conv3x3_torgb_split): the 128-channel 3x3 layer on split activations, its wide ToRGB and the skip-image sum in one launch, the
    layer's activations never stored — against the two launches it replaces (the same layer writing a split result, then torgb_wide_skip: the same (hi, lo) pieces
    into the same three products per term, another summation order: <= 2e-6 of the range) and against fp64 of the same dense input (the bf16x3 bar per layer: 1e-5)."""
    import torch.nn.functional as F
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(ci + rco + h)
    co = 128
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    xs = modconv.SplitActs(_nhwc(_split_storage(x.cpu())).cuda())
    weight = torch.randn(co, ci, 3, 3, device='cuda'); styles = torch.randn(n, ci, device='cuda') + 1
    rgb_weight = torch.randn(rco, co, 1, 1, device='cuda'); rgb_styles = (torch.randn(n, co, device='cuda') + 1) / co ** 0.5
    bias, rgb_bias = torch.randn(co, device='cuda'), torch.randn(rco, device='cuda')
    nz = torch.randn(h, w, device='cuda') if noise else None
    ns = torch.tensor(0.3, device='cuda') if noise else None
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
    prev = _nhwc(torch.randn(n, rco, h // 2, w // 2, device='cuda')) if skip else None
    gain = 2 ** 0.5 if act else 1.0
    with torch.no_grad():
        assert modconv.conv3x3_torgb_wide_supported(xs, weight, rgb_weight, prev, f, 1, 'lrelu' if act else 'linear')
        w3 = modconv.modulate_weights(weight, styles, dtype=modconv.BF16X3)
        w1 = modconv.modulate_weights(rgb_weight, rgb_styles, demodulate=False, dtype=modconv.BF16X3)
        n0 = _lib.launch_count('conv')
        got = modconv.conv3x3_torgb_wide(xs, w3, bias, nz, ns, act, gain, clamp, w1, rgb_bias, 256.0, prev, f)
        assert _lib.launch_count('conv') == n0 + 1
        assert got.shape == (n, rco, h, w) and got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
        y = modconv.conv2d(xs, w3, bias=bias, noise=nz, noise_strength=ns, act=act, gain=gain, clamp=clamp, split=True, out_split=True)
        assert isinstance(y, modconv.SplitActs)
        two = modconv.torgb_wide_skip(y, rgb_weight, rgb_styles, rgb_bias, 256.0, prev, f) if w % 32 == 0 else None
    wq = modconv.modulate_weights(weight, styles, dtype=torch.float32).double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3).cpu()
    ref = torch.stack([F.conv2d(x[i:i + 1].double().cpu(), wq[i], padding=1)[0] for i in range(n)])
    if noise:
        ref = ref + (nz * ns).double().cpu()
    ref = ref + bias.double().cpu().view(1, -1, 1, 1)
    ref = (F.leaky_relu(ref, 0.2) if act else ref) * gain
    if clamp >= 0:
        ref = ref.clamp(-clamp, clamp)
    ref = (torch.einsum('oc,nc,nchw->nohw', rgb_weight.reshape(rco, co).double().cpu(), rgb_styles.double().cpu(), ref) + rgb_bias.double().cpu().view(1, rco, 1, 1)).clamp(-256, 256)
    if skip:
        ref = ref + upfirdn2d.upsample2d(prev.double().cpu().contiguous(), f.cpu())
    e64 = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    e2 = float((got - two).abs().max() / two.abs().max()) if two is not None else None
    print((ci, rco, h, w, n, skip), 'vs two launches', e2, 'vs fp64', e64)
    assert (e2 is None or e2 < 2e-6) and e64 < 2e-5                    # (two bf16x3 layers in sequence: twice the one-layer bar; the two-launch form carries the same error)


@pytest.mark.gpu
def test_wide_torgb_narrow_output_first_then_wide_in_a_fresh_process(hip_lib):
    """The kernel's dynamic-LDS limit is reserved once per device: a process whose FIRST call is the narrow form (Ci 256, Co 32: 32 KB) must still
    be able to launch the widest one (Co 96: 96 KB) afterwards.  Needs its own process — in this one the order of the tests above decides."""
    import os, subprocess, sys
    code = (
        "import torch\n"
        "from pix2pix3d_amd.torch_utils.ops import modconv\n"
        "def run(ci, co):\n"
        "    torch.manual_seed(co)\n"
        "    x = torch.randn(1, ci, 16, 64, device='cuda').contiguous(memory_format=torch.channels_last)\n"
        "    v = x.cpu().permute(0, 2, 3, 1).reshape(1, 16, 64, ci // 32, 32)\n"
        "    hi = v.to(torch.bfloat16); lo = (v - hi.float()).to(torch.bfloat16)\n"
        "    rows = torch.stack([hi, lo], dim=-2).reshape(1, 16, 64, ci // 32, 64).view(torch.float32).reshape(1, 16, 64, ci).permute(0, 3, 1, 2)\n"
        "    xs = modconv.SplitActs(rows.contiguous(memory_format=torch.channels_last).cuda())\n"
        "    w = torch.randn(co, ci, 1, 1, device='cuda'); s = (torch.randn(1, ci, device='cuda') + 1) / ci ** 0.5; b = torch.randn(co, device='cuda')\n"
        "    with torch.no_grad():\n"
        "        assert modconv.torgb_wide_skip_supported(xs, w, None, None)\n"
        "        got = modconv.torgb_wide_skip(xs, w, s, b, 256.0, None, None)\n"
        "    ref = (torch.einsum('oc,nc,nchw->nohw', w.reshape(co, ci).double(), s.double(), xs.dense().double()) + b.double().view(1, co, 1, 1)).clamp(-256, 256)\n"
        "    e = float((got.double() - ref).abs().max() / ref.abs().max())\n"
        "    assert e < 1e-5, (ci, co, e)\n"
        "for ci, co in ((128, 32), (128, 96), (256, 32), (256, 64), (256, 96)):\n"
        "    run(ci, co)\n"
        "torch.cuda.synchronize(); print('ORDER_OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and 'ORDER_OK' in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [4, 1])
def test_generator_outputs_do_not_change_with_split_activations(hip_lib, batch):
    """G.synthesis at the benchmark's size (seg2cat, batch 4, 128^2 rays) with the activations of the >= 64^2 backbone blocks kept split vs.
    plain tensors: the same function to fp32 summation order."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    from conftest import load_golden
    from model_cases import build_generator, uniforms, replay_uniforms
    g = load_golden('model_full_seg2cat_128')
    G = build_generator('seg2cat', 'cuda', depth=tuple(int(v) for v in g['depth']))
    ws, c, nrr = torch.tensor(g['ws'], device='cuda'), torch.tensor(g['c'], device='cuda'), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)
    ws, c, u_c, u_f = ws[:batch], c[:batch], u_c[:batch], u_f[:batch * nrr * nrr]      # batch 1: no shared-weight form, EVERY block of the backbone hands over split
    prev = modconv.split_activations
    orig = modconv.SplitActs.__init__
    try:
        outs = []
        for on in (True, False):
            modconv.split_activations = on
            made = []

            def spy(self, t, _made=made):
                _made.append(tuple(t.shape)); orig(self, t)
            modconv.SplitActs.__init__ = spy
            with replay_uniforms(u_c, u_f), torch.no_grad():
                o = G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=nrr)
            outs.append((o, made))
        (a, made_on), (b, made_off) = outs
        assert len(made_on) >= (5 if batch > 1 else 7) and not made_off, (made_on, made_off)    # batch 4: conv0 and conv1 of b64, b128 and conv0 of b256 hand their results over split
                                                                                                # (b256.conv1's never leave the registers: test_last_backbone_layer_never_stores_its_activations)
        for k in ('image', 'image_raw', 'semantic', 'semantic_raw', 'image_depth'):      # (the 3x3 layers on split input take the ring kernel: same products, another summation order)
            e = float((a[k].float() - b[k].float()).abs().max() / b[k].float().abs().max())
            assert e < (2e-3 if k in ('image', 'semantic') else 2e-5), (k, e)       # fp16 SR heads amplify a last-bit difference of their input to fp16 rounding
    finally:
        modconv.split_activations = prev
        modconv.SplitActs.__init__ = orig


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [4, 1])
def test_last_backbone_layer_never_stores_its_activations(hip_lib, batch):
    """G.synthesis at the benchmark's size: the backbone's last block runs conv1 + wide ToRGB + skip-image sum as ONE launch (modconv.conv3x3_torgb_wide, x never written)
    — the outputs against the same pass with that fusion off (conv1 writing split activations, then torgb_wide_skip): the same products in another summation order.  A
    forward hook on the block's conv1 (somebody wants the activations) keeps the two-launch form."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    from conftest import load_golden
    from model_cases import build_generator, uniforms, replay_uniforms
    g = load_golden('model_full_seg2cat_128')
    G = build_generator('seg2cat', 'cuda', depth=tuple(int(v) for v in g['depth']))
    ws, c, nrr = torch.tensor(g['ws'], device='cuda'), torch.tensor(g['c'], device='cuda'), int(g['nrr'])
    u_c, u_f = uniforms(g, ws.shape[0], nrr, G.rendering_kwargs)
    ws, c, u_c, u_f = ws[:batch], c[:batch], u_c[:batch], u_f[:batch * nrr * nrr]
    prev = modconv.fuse_conv_wide_torgb
    try:
        outs = []
        for on in (True, False):
            modconv.fuse_conv_wide_torgb = on
            n0 = modconv.conv_wide_torgb_calls
            with replay_uniforms(u_c, u_f), torch.no_grad():
                o = G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=nrr)
            outs.append(o)
            assert modconv.conv_wide_torgb_calls - n0 == (1 if on else 0)
        a, b = outs
        for k in ('image', 'image_raw', 'semantic', 'semantic_raw', 'image_depth'):
            e = float((a[k].float() - b[k].float()).abs().max() / b[k].float().abs().max())
            assert e < (2e-3 if k in ('image', 'semantic') else 2e-5), (k, e)       # fp16 SR heads amplify a last-bit difference of their input to fp16 rounding
        modconv.fuse_conv_wide_torgb = True
        last = getattr(G.backbone.synthesis, f'b{G.backbone.synthesis.img_resolution}')
        seen = []
        handle = last.conv1.register_forward_hook(lambda m, a_, o_: seen.append(type(o_)))
        try:
            n0 = modconv.conv_wide_torgb_calls
            with replay_uniforms(u_c, u_f), torch.no_grad():
                G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=nrr)
            assert modconv.conv_wide_torgb_calls == n0 and len(seen) == 1
        finally:
            handle.remove()
    finally:
        modconv.fuse_conv_wide_torgb = prev
