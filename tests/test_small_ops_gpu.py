"""One-launch helpers of the low-resolution layers (csrc/small_ops.hip) against plain torch fp32 / fp64 references:
``fc`` vs the FullyConnectedLayer formula, ``im2col3x3`` vs F.unfold (bit-exact: pure data movement), ``noise_bias_act`` vs the
noise add + bias_act composition, and a whole small SynthesisLayer / block against the generic (tensor-op) route."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,in_f,out_f', [(4, 512, 512), (1, 512, 96), (16, 64, 33), (3, 516, 7)])
@pytest.mark.parametrize('act', ['linear', 'lrelu'])
def test_fc_matches_formula(hip_lib, n, in_f, out_f, act):
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(n + in_f + out_f)
    x = torch.randn(n, in_f, device='cuda')
    w = torch.randn(out_f, in_f, device='cuda')
    b = torch.randn(out_f, device='cuda')
    wg, bg, scale = 1 / np.sqrt(in_f), 0.5, 0.37
    assert modconv.fc_supported(x, w, b, act)
    y = modconv.fc(x, w, b, wg, bg, act, out_scale=scale)
    ref = x.double() @ (w.double() * wg).t() + b.double() * bg
    if act == 'lrelu':
        ref = F.leaky_relu(ref, 0.2) * np.sqrt(2)
    ref = ref * scale
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6       # fp32 accumulation order only
    y0 = modconv.fc(x, w, None, wg, bg, act)
    ref0 = x.double() @ (w.double() * wg).t()
    if act == 'lrelu':
        ref0 = F.leaky_relu(ref0, 0.2) * np.sqrt(2)
    assert rel_err(y0.cpu().numpy(), ref0.cpu().numpy()) < 2e-6


@pytest.mark.parametrize('h,w', [(9, 9), (17, 12)])
def test_im2col_strided_is_unfold(hip_lib, h, w):
    from pix2pix3d_amd.torch_utils.ops import modconv
    x = torch.randn(2, 5, h, w, device='cuda').to(memory_format=torch.channels_last)
    assert torch.equal(modconv.im2col3x3(x, pad=0, stride=2), F.unfold(x.contiguous(), kernel_size=3, padding=0, stride=2))


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('n,c,h,w', [(4, 512, 4, 4), (2, 37, 9, 5), (1, 8, 32, 32)])
def test_im2col_is_unfold(hip_lib, layout, n, c, h, w):
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(c + h)
    x = torch.randn(n, c, h, w, device='cuda')
    if layout == 'nhwc':
        x = x.to(memory_format=torch.channels_last)
    cols = modconv.im2col3x3(x)
    ref = F.unfold(x.contiguous(), kernel_size=3, padding=1)
    assert cols.shape == ref.shape
    assert torch.equal(cols, ref)                                    # data movement: bit-exact


@pytest.mark.parametrize('act', ['linear', 'lrelu'])
@pytest.mark.parametrize('with_noise', [False, True])
@pytest.mark.parametrize('clamp', [None, 0.7])
def test_noise_bias_act(hip_lib, act, with_noise, clamp):
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(5)
    n, c, h, w = 3, 21, 6, 10
    y = torch.randn(n, c, h, w, device='cuda')
    bias = torch.randn(c, device='cuda')
    noise = torch.randn(h, w, device='cuda') if with_noise else None
    ns = torch.tensor(0.31, device='cuda') if with_noise else None
    ref = y.double()
    if with_noise:
        ref = ref + noise.double() * ns.double()
    ref = ref + bias.double().reshape(1, -1, 1, 1)
    if act == 'lrelu':
        ref = F.leaky_relu(ref, 0.2)
    ref = ref * 1.4
    if clamp is not None:
        ref = ref.clamp(-clamp, clamp)
    out = modconv.noise_bias_act(y.clone(), bias, noise, ns, act, 1.4, clamp)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-6


@pytest.mark.parametrize('res,up', [(8, 1), (16, 2), (32, 1), (32, 2)])
def test_small_synthesis_layer_matches_generic_route(hip_lib, res, up):
    """A low-resolution SynthesisLayer through the launch-diet route vs the same module with the native path disabled."""
    from pix2pix3d_amd.training import networks_stylegan2 as ns2
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(res + up)
    layer = ns2.SynthesisLayer(48, 40, w_dim=64, resolution=res, up=up).cuda().eval()
    with torch.no_grad():
        layer.noise_strength.fill_(0.2)
        layer.bias.normal_()
    x = torch.randn(2, 48, res // up, res // up, device='cuda')
    w = torch.randn(2, 64, device='cuda')
    with torch.no_grad():
        y = layer(x, w, noise_mode='const')
        modconv.enabled = False
        try:
            ref = layer(x, w, noise_mode='const')
        finally:
            modconv.enabled = True
    assert y.shape == ref.shape
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('res', [8, 64, 128])
def test_native_ray_sampler_matches_tensor_ops(hip_lib, res):
    from pix2pix3d_amd.training.volumetric_rendering import ray_sampler
    torch.manual_seed(res)
    n = 3
    c2w = torch.eye(4, device='cuda')[None].repeat(n, 1, 1)
    q, _ = torch.linalg.qr(torch.randn(n, 3, 3, device='cuda'))
    c2w[:, :3, :3] = q
    c2w[:, :3, 3] = torch.randn(n, 3, device='cuda') * 2.7
    k = torch.tensor([[4.26, 0.03, 0.5], [0, 4.1, 0.48], [0, 0, 1]], device='cuda')[None].repeat(n, 1, 1)
    rs = ray_sampler.RaySampler()
    o, d = rs(c2w, k, res)
    ray_sampler.native = False
    try:
        o_ref, d_ref = rs(c2w, k, res)
    finally:
        ray_sampler.native = True
    assert torch.equal(o, o_ref)
    assert (d - d_ref).abs().max().item() < 2e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('separable', [True, False])
@pytest.mark.parametrize('size', [17, 40, 97])
def test_fused_fir_any_filter(hip_lib, dtype, separable, size):
    """4x4 FIR + noise + bias + lrelu in one pass (csrc/upfirdn2d.hip): rank-1 filters take the two-pass branch, anything else the
    16-tap branch; both against upfirdn2d + the explicit epilogue, on sizes that leave partial tiles and partial tile groups."""
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(size)
    c = 64
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda') if separable else torch.rand(4, 4, device='cuda') / 8
    x = torch.randn(2, c, size + 1, size + 1, device='cuda').to(dtype).to(memory_format=torch.channels_last)
    bias = torch.randn(c, device='cuda')
    noise = torch.randn(size, size, device='cuda')
    ns = torch.tensor(0.3, device='cuda')
    y = modconv.fir4_bias_act(x, f, bias, noise, ns, 'lrelu', 1.3, 2.5)
    ref = upfirdn2d.upfirdn2d(x.float().contiguous(), f, padding=[1, 1, 1, 1], gain=4).double()
    ref = ref + (noise * ns).double() + bias.double().reshape(1, -1, 1, 1)
    ref = (F.leaky_relu(ref, 0.2) * 1.3).clamp(-2.5, 2.5)
    assert y.shape == ref.shape and y.dtype == dtype
    assert rel_err(y.float().cpu().numpy(), ref.cpu().numpy()) < (2e-3 if dtype == torch.float16 else 1e-5)


@pytest.mark.parametrize('cin,cout,k,down,res', [(64, 64, 3, 1, 64), (64, 128, 3, 2, 64), (64, 128, 1, 2, 64), (6, 64, 1, 1, 64), (96, 40, 3, 2, 50),
                                                 (48, 40, 3, 1, 16), (48, 40, 3, 2, 16), (48, 40, 1, 2, 8), (40, 24, 3, 2, 4)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_plain_conv2d_layer_native_route(hip_lib, cin, cout, k, down, res, dtype):
    """Conv2dLayer (the Encoder / discriminator building block) on the MFMA kernels vs the same module on the generic route:
    "same" 3x3, FIR + valid stride-2 3x3, FIR-decimate + 1x1 (skip branch), 1x1 fromrgb with 6 input channels."""
    from pix2pix3d_amd.training import networks_stylegan2 as ns2
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(cin + cout + k + down)
    layer = ns2.Conv2dLayer(cin, cout, kernel_size=k, activation='lrelu', down=down, conv_clamp=256 if dtype == torch.float16 else None).cuda().eval()
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(2, cin, res, res, device='cuda').to(dtype)
    with torch.no_grad():
        if not modconv.plain_layer_supported(x, layer.weight, 1, down, 'lrelu'):
            pytest.skip('low-resolution GEMM route is fp32 only')
        y = layer(x, gain=0.7)
        modconv.enabled = False
        try:
            ref = layer(x.float(), gain=0.7)            # fp32 reference of the same (fp16-rounded) input
        finally:
            modconv.enabled = True
    assert y.shape == ref.shape and y.dtype == dtype
    assert rel_err(y.float().cpu().numpy(), ref.cpu().numpy()) < (3e-3 if dtype == torch.float16 else 1e-5)


@pytest.mark.parametrize('num_fp16_res', [0, 2])
def test_discriminator_and_encoder_forward_on_native_convs(hip_lib, num_fp16_res):
    """DualDiscriminator.forward (and with it every DiscriminatorBlock / Conv2dLayer variant: fromrgb, 3x3, down-2 3x3, down-2 1x1
    skip, low-resolution GEMM route, 4x4 epilogue) under no_grad on the device = the native conv route, against the same module with
    the native route switched off (ATen convolutions)."""
    from pix2pix3d_amd.training.dual_discriminator import DualDiscriminator
    from pix2pix3d_amd.torch_utils.ops import modconv
    from pix2pix3d_amd import _lib
    torch.manual_seed(3)
    D = DualDiscriminator(c_dim=25, img_resolution=128, img_channels=3, channel_base=4096, channel_max=128, num_fp16_res=num_fp16_res,
                          conv_clamp=256 if num_fp16_res else None, block_kwargs=dict(freeze_layers=0), mapping_kwargs=dict(),
                          epilogue_kwargs=dict(mbstd_group_size=2)).cuda().eval().requires_grad_(False)
    img = {'image': torch.randn(2, 3, 128, 128, device='cuda'), 'image_raw': torch.randn(2, 3, 32, 32, device='cuda')}
    c = torch.randn(2, 25, device='cuda')
    with torch.no_grad():
        n0 = _lib.launch_count('conv')
        y = D(img, c)
        assert _lib.launch_count('conv') > n0, 'the native conv kernels did not run'
        modconv.enabled = False
        try:
            ref = D(img, c)
        finally:
            modconv.enabled = True
    assert y.shape == (2, 1)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < (2e-2 if num_fp16_res else 1e-4)


def test_plain_layer_weight_cache_survives_recycled_tensors(hip_lib):
    """The derived-weight cache of the native Conv2dLayer route must not serve a freed layer's weights to a new layer that happens to get
    the same id() / data_ptr() (regression: a key made of those alone did)."""
    import gc
    from pix2pix3d_amd.training import networks_stylegan2 as ns2
    from pix2pix3d_amd.torch_utils.ops import modconv
    x = torch.randn(1, 64, 64, 64, device='cuda')
    for i in range(12):
        torch.manual_seed(100 + i)
        layer = ns2.Conv2dLayer(64, 64, kernel_size=3, activation='lrelu').cuda().eval()
        with torch.no_grad():
            y = layer(x)
            modconv.enabled = False
            try:
                ref = layer(x)
            finally:
                modconv.enabled = True
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5, i
        del layer
        gc.collect()


@pytest.mark.gpu
def test_demod_coefficients_of_several_layers_in_one_launch(hip_lib):
    """p3d_demod_coefs_multi (modconv.demod_coefs_many: every shared-weight layer of a network ahead of the convolutions) == one p3d_demod_coefs each, bit for bit,
    and == the reference's rsqrt(sum (w s)^2 + 1e-8) (networks_stylegan2.py:57-63)."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(3)
    pairs = [(torch.randn(co, ci, 3, 3, device='cuda'), torch.randn(4, ci, device='cuda') + 1) for co, ci in ((512, 512), (512, 512), (256, 512), (96, 64), (130, 36), (512, 512), (64, 128))]
    many = modconv.demod_coefs_many(pairs)
    for (w, s), d in zip(pairs, many):
        one = modconv.demod_coefs(w, s)
        assert d.shape == one.shape == (4, w.shape[0]) and torch.equal(d, one)
        ref = ((w.double()[None] * s.double()[:, None, :, None, None]).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        assert float(((d.double() - ref).abs() / ref).max()) < 1e-5
    assert len(modconv.demod_coefs_many(pairs[:1])) == 1


@pytest.mark.gpu
def test_rays_straight_from_the_camera_labels(hip_lib):
    """RaySampler.forward on the two views the generators pass (c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3): p3d_ray_sample_labels reads the label rows in
    place) == the same call on contiguous copies (p3d_ray_sample), bit for bit, == the oracle's ray_sampler (ray_sampler.py:24-62); a view that is NOT such a pair
    takes the copying route."""
    from oracle import render_oracle as R
    from pix2pix3d_amd import _lib, configs
    from pix2pix3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    rs = RaySampler()
    c = torch.tensor(np.stack([configs.orbit_camera(7 * k + 3, radius=2.7, pivot=[0, 0, -0.06]) for k in range(5)]), device='cuda')
    for lab in (c, torch.cat([c, torch.zeros(5, 7, device='cuda')], 1)):           # 25-float rows, and rows padded to 32 floats
        c2w, k = lab[:, :16].view(-1, 4, 4), lab[:, 16:25].view(-1, 3, 3)
        assert not c2w.is_contiguous()
        n0 = _lib.launch_count('aux')
        o1, d1 = rs(c2w, k, 48)
        assert _lib.launch_count('aux') == n0 + 1
        o0, d0 = rs(c2w.contiguous(), k.contiguous(), 48)
        assert torch.equal(o1, o0) and torch.equal(d1, d0)
        oo, do = R.ray_sampler(c2w.cpu().numpy(), k.cpu().numpy(), 48)
        assert np.abs(o1.cpu().numpy() - oo).max() < 1e-6 and np.abs(d1.cpu().numpy() - do).max() < 2e-6
    k_other = torch.eye(3, device='cuda').repeat(5, 1, 1) * 4.2647               # intrinsics from another tensor: not a label pair
    o2, d2 = rs(c[:, :16].view(-1, 4, 4), k_other, 16)
    o3, d3 = rs(c[:, :16].view(-1, 4, 4).contiguous(), k_other, 16)
    assert torch.equal(o2, o3) and torch.equal(d2, d3)


@pytest.mark.gpu
def test_last_sr_block_stores_its_activations_when_somebody_looks(hip_lib):
    """The super-resolution heads let their last block skip the store of x (nobody reads it) — unless a forward hook on that block could: then x is stored and the
    image is the same."""
    from pix2pix3d_amd.training.superresolution import SuperresolutionHybrid8XDC
    torch.manual_seed(0)
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True, channel_base=32768, channel_max=512,
                                   fused_modconv_default='inference_only').cuda().eval().requires_grad_(False)
    x = torch.randn(2, 32, 128, 128, device='cuda')
    ws = torch.randn(2, 14, 512, device='cuda')
    with torch.no_grad():
        y0 = sr(x[:, :3].contiguous(), x, ws, noise_mode='none')
    seen = []
    h = sr.block1.register_forward_hook(lambda m, a, out: seen.append(out[0]))
    try:
        with torch.no_grad():
            y1 = sr(x[:, :3].contiguous(), x, ws, noise_mode='none')
    finally:
        h.remove()
    assert len(seen) == 1 and seen[0] is not None and tuple(seen[0].shape) == (2, 128, 512, 512) and bool(torch.isfinite(seen[0].float()).all())
    assert y0.shape == y1.shape == (2, 3, 512, 512)
    assert float((y0 - y1).abs().max()) <= 1e-5 * float(y1.abs().max())           # two epilogues of the same contraction (another fp32 summation order)
    # a hook one level down — a feature extractor on the block's last LAYER — must be handed the activations too
    seen2 = []
    h = sr.block1.conv1.register_forward_hook(lambda m, a, out: seen2.append(out))
    try:
        with torch.no_grad():
            y2 = sr(x[:, :3].contiguous(), x, ws, noise_mode='none')
    finally:
        h.remove()
    got = seen2[0][0] if isinstance(seen2[0], tuple) else seen2[0]
    assert len(seen2) == 1 and torch.is_tensor(got) and tuple(got.shape) == (2, 128, 512, 512) and bool(torch.isfinite(got.float()).all())
    assert torch.equal(y2, y1)
