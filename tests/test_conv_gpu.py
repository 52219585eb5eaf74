"""The MFMA implicit-GEMM modulated convolution (csrc/conv2d.hip) against fp32 torch convolutions of the same
fp16-rounded operands.  fp16 storage, fp32 accumulation: tolerance 2e-3 relative-to-max on the conv output (one fp16
rounding of the result), 1e-3 on the fp32 ToRGB output."""
import numpy as np
import os
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.to(memory_format=torch.channels_last)


def _ref_modulated(weight, styles, demod=True):
    w = weight[None].float() * styles[:, None, :, None, None].float()
    if demod:
        w = w * (w.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    return w


@pytest.mark.parametrize('ci,co,h,w', [(64, 96, 16, 16), (128, 128, 9, 20), (32, 130, 12, 12)])
@pytest.mark.parametrize('k', [3, 1])
def test_fp32_conv_is_exact_fp32(hip_lib, ci, co, h, w, k):
    """fp32 variant (v_mfma_f32_32x32x2_f32): plain, transposed and 1x1 against torch fp64 references; error is fp32 rounding."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(ci + co + k)
    n = 2
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    weight = torch.randn(co, ci, k, k, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles, demodulate=(k == 3), dtype=torch.float32)
    wq = wmod.double().reshape(n, co, k, k, ci).permute(0, 1, 4, 2, 3)
    wref = _ref_modulated(weight.double(), styles.double(), demod=(k == 3))
    assert rel_err(wq.cpu().numpy(), wref.cpu().numpy()) < 1e-6
    bias = torch.randn(co, device='cuda')
    y = modconv.conv2d(x, wmod, bias=bias, act=1, gain=1.3, clamp=2.0)
    yr = torch.stack([F.conv2d(x[i:i + 1].double(), wq[i], padding=k // 2)[0] for i in range(n)])
    yr = (F.leaky_relu(yr + bias.double().view(1, -1, 1, 1), 0.2) * 1.3).clamp(-2, 2)
    assert y.dtype == torch.float32 and rel_err(y.cpu().numpy(), yr.cpu().numpy()) < 1e-5          # fp32 accumulation over K = 9*Ci terms
    if k == 3:
        yt = modconv.conv2d(x, wmod, transposed=True)
        ytr = torch.stack([F.conv_transpose2d(x[i:i + 1].double(), wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
        assert rel_err(yt.cpu().numpy(), ytr.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('ci,co,h,w', [(64, 128, 16, 16), (128, 128, 33, 20), (256, 96, 8, 40), (32, 256, 12, 12), (64, 200, 5, 7)])
def test_modulate_and_conv3x3(hip_lib, ci, co, h, w):
    from pix2pix3d_amd.torch_utils.ops import modconv
    from pix2pix3d_amd import _lib
    torch.manual_seed(ci + co)
    n = 3
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda').half())
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles)
    wref = _ref_modulated(weight, styles)                                   # [n, co, ci, 3, 3]
    assert rel_err(wmod.float().cpu().numpy(), wref.permute(0, 1, 3, 4, 2).reshape(n, co, 9, ci).cpu().numpy()) < 1e-3
    wq = wmod.float().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3)       # the fp16-rounded weights the kernel sees
    n0 = _lib.launch_count('conv')
    y = modconv.conv3x3(x, wmod)
    assert _lib.launch_count('conv') > n0 and y.is_contiguous(memory_format=torch.channels_last)
    yr = torch.stack([F.conv2d(x[i:i + 1].float(), wq[i], padding=1)[0] for i in range(n)])
    assert rel_err(y.float().cpu().numpy(), yr.cpu().numpy()) < 2e-3
    # fused epilogue: noise, bias, lrelu, gain, clamp
    bias = torch.randn(co, device='cuda')
    noise = torch.randn(h, w, device='cuda')
    strength = torch.tensor(0.3, device='cuda')
    y2 = modconv.conv3x3(x, wmod, bias=bias, noise=noise, noise_strength=strength, act=1, gain=2 ** 0.5, clamp=1.5)
    yr2 = (F.leaky_relu(yr + noise * strength + bias.view(1, -1, 1, 1), 0.2) * 2 ** 0.5).clamp(-1.5, 1.5)
    assert rel_err(y2.float().cpu().numpy(), yr2.cpu().numpy()) < 2e-3
    # shared (non-modulated) weights: stride 0 between images
    y3 = modconv.conv3x3(x, wmod[:1])
    yr3 = F.conv2d(x.float(), wq[0], padding=1)
    assert rel_err(y3.float().cpu().numpy(), yr3.cpu().numpy()) < 2e-3


@pytest.mark.parametrize('ci,co,h,w', [(64, 128, 8, 8), (128, 64, 17, 9), (32, 256, 16, 16),
                                      (64, 128, 32, 32), (32, 256, 40, 33), (256, 128, 48, 70), (96, 128, 64, 32)])   # >= 32 x 32 with Co % 128 == 0: convT_h2_f16_kernel
def test_transposed_stride2_conv(hip_lib, ci, co, h, w):
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(ci * co)
    n = 2
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda').half())
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles)
    wq = wmod.float().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3)       # [n, co, ci, 3, 3]
    y = modconv.conv3x3(x, wmod, transposed=True)
    assert tuple(y.shape) == (n, co, 2 * h + 1, 2 * w + 1)
    yr = torch.stack([F.conv_transpose2d(x[i:i + 1].float(), wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    assert rel_err(y.float().cpu().numpy(), yr.cpu().numpy()) < 2e-3


@pytest.mark.parametrize('ci,co', [(128, 3), (256, 6), (128, 1), (64, 4)])
def test_torgb(hip_lib, ci, co):
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(co)
    n, h, w = 2, 24, 20
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda').half())
    weight = torch.randn(co, ci, 1, 1, device='cuda')
    styles = torch.randn(n, ci, device='cuda') / ci ** 0.5
    bias = torch.randn(co, device='cuda')
    y = modconv.torgb(x, weight, styles, bias, clamp=1.0)
    yr = (torch.einsum('nihw,oi,ni->nohw', x.float(), weight[:, :, 0, 0], styles) + bias.view(1, -1, 1, 1)).clamp(-1, 1)
    assert y.dtype == torch.float32 and rel_err(y.cpu().numpy(), yr.cpu().numpy()) < 1e-3
    acc = torch.ones_like(y)
    y2 = modconv.torgb(x, weight, styles, bias, clamp=1.0, out=acc)
    assert y2 is acc and rel_err(acc.cpu().numpy(), (yr + 1).cpu().numpy()) < 1e-3


def test_synthesis_layer_native_vs_generic(hip_lib):
    """A whole fp16 channels-last SynthesisLayer / ToRGB through the native path equals the generic operator route."""
    from pix2pix3d_amd.training.networks_stylegan2 import SynthesisLayer, ToRGBLayer
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(0)
    for up in (1, 2):
        layer = SynthesisLayer(128, 64, w_dim=32, resolution=32 * up, up=up, conv_clamp=256, channels_last=True).cuda().eval().requires_grad_(False)
        layer.noise_strength.fill_(0.2); layer.bias.normal_()
        x = _nhwc(torch.randn(2, 128, 32, 32, device='cuda').half())
        wl = torch.randn(2, 32, device='cuda')
        for mode in ('const', 'none'):
            with torch.no_grad():
                modconv.enabled = True
                y1 = layer(x, wl, noise_mode=mode, fused_modconv=True)
                modconv.enabled = False
                y0 = layer(x, wl, noise_mode=mode, fused_modconv=True)
                modconv.enabled = True
            assert y1.shape == y0.shape and rel_err(y1.float().cpu().numpy(), y0.float().cpu().numpy()) < 6e-3, (up, mode)
    # fp32 layers (the tri-plane backbone) through the fp32 MFMA kernel, incl. a wide ToRGB (1x1 through the same kernel)
    for up in (1, 2):
        layer = SynthesisLayer(64, 64, w_dim=32, resolution=32 * up, up=up).cuda().eval().requires_grad_(False)
        layer.noise_strength.fill_(0.2); layer.bias.normal_()
        x = _nhwc(torch.randn(2, 64, 32, 32, device='cuda'))
        wl = torch.randn(2, 32, device='cuda')
        with torch.no_grad():
            y1 = layer(x, wl, noise_mode='const', fused_modconv=True)
            modconv.enabled = False
            y0 = layer(x, wl, noise_mode='const', fused_modconv=True)
            modconv.enabled = True
        assert y1.dtype == torch.float32 and rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 2e-5, up
    wide = ToRGBLayer(64, 96, w_dim=32).cuda().eval().requires_grad_(False)
    with torch.no_grad():
        y1 = wide(x, wl)
        modconv.enabled = False
        y0 = wide(x, wl)
        modconv.enabled = True
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 2e-5
    rgb = ToRGBLayer(128, 3, w_dim=32, conv_clamp=256, channels_last=True).cuda().eval().requires_grad_(False)
    rgb.bias.normal_()
    x = _nhwc(torch.randn(2, 128, 32, 32, device='cuda').half())
    with torch.no_grad():
        y1 = rgb(x, wl)
        modconv.enabled = False
        y0 = rgb(x, wl).float()
        modconv.enabled = True
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 6e-3


@pytest.mark.parametrize('dtype,ci,co,h,w', [
    (torch.float16, 128, 128, 64, 64),      # conv3x3_h2_f16_kernel: the fp16 SR layers (16 x 16 patches, Ci multiple of 64, Co of 128)
    (torch.float16, 256, 256, 80, 72),      # ... with partial patches on both axes and four channel chunks
    (torch.float16, 128, 256, 64, 200),
    (torch.float16, 64, 128, 40, 24),       # conv3x3_halo_kernel<half>: 8 x 16 patches (Ci not a multiple of 128)
    (torch.float32, 64, 128, 24, 40),       # conv3x3_halo_kernel<float>: the fp32 backbone layers
    (torch.float32, 96, 72, 19, 33),
])
@pytest.mark.parametrize('with_noise', [False, True])
def test_halo_kernels_at_their_own_sizes(hip_lib, dtype, ci, co, h, w, with_noise):
    """The halo-reuse 3x3 kernels only engage from 8 x 16 (fp32 / narrow fp16) and 32 x 32 (fp16, 16 x 16 patches) pixels on: checked here directly (bias, noise, lrelu, gain,
    clamp in the epilogue) against torch's fp32 convolution of the same (fp16-rounded) operands, with per-sample weights."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(ci + co + h)
    n = 2
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda').to(dtype))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    bias = torch.randn(co, device='cuda')
    noise = torch.randn(h, w, device='cuda') if with_noise else None
    ns = torch.tensor(0.25, device='cuda') if with_noise else None
    wmod = modconv.modulate_weights(weight, styles, dtype=dtype)
    y = modconv.conv2d(x, wmod, bias=bias, noise=noise, noise_strength=ns, act=1, gain=1.3, clamp=3.0)
    wq = wmod.float().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3)
    ref = torch.stack([F.conv2d(x[i:i + 1].float(), wq[i], padding=1)[0] for i in range(n)])
    if with_noise:
        ref = ref + noise * ns
    ref = (F.leaky_relu(ref + bias.reshape(1, -1, 1, 1), 0.2) * 1.3).clamp(-3.0, 3.0)
    assert y.shape == ref.shape and y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(y.float().cpu().numpy(), ref.cpu().numpy()) < (2e-3 if dtype == torch.float16 else 1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('res', [4, 8, 16, 32])
def test_lowres_512_channel_layers_take_the_split_k_schedule(hip_lib, dtype, res):
    """The backbone's 512-channel layers at 4^2 .. 32^2: a handful of output tiles with a 144-step K loop.  The generic kernel deals
    the K steps to many work-groups (fp32 partial tiles + a finishing launch that also applies noise / bias / lrelu / gain / clamp);
    per-sample weights, shared weights (batch folded into the GEMM rows) and the x2 transposed form, against fp64 torch."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(res)
    n, ci, co = 4, 512, 512
    x = _nhwc(torch.randn(n, ci, res, res, device='cuda').to(dtype))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    wmod = modconv.modulate_weights(weight, styles, dtype=dtype)
    mode = 0
    assert int(_lib.lib().p3d_conv2d_nhwc_workspace(_lib.DTYPE_CODE[dtype], n, res, res, ci, co, co * 9 * ci, 3, mode)) > 0 or (dtype == torch.float16 and res == 32)
    wq = wmod.double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3)
    bias, noise, strength = torch.randn(co, device='cuda'), torch.randn(res, res, device='cuda'), torch.tensor(0.3, device='cuda')
    tol = 2e-5 if dtype == torch.float32 else 3e-3
    y = modconv.conv3x3(x, wmod, bias=bias, noise=noise, noise_strength=strength, act=1, gain=2 ** 0.5, clamp=1.5)
    yr = torch.stack([F.conv2d(x[i:i + 1].double(), wq[i], padding=1)[0] for i in range(n)])
    yr = (F.leaky_relu(yr + (noise * strength).double() + bias.double().view(1, -1, 1, 1), 0.2) * 2 ** 0.5).clamp(-1.5, 1.5)
    assert rel_err(y.double().cpu().numpy(), yr.cpu().numpy()) < tol
    y3 = modconv.conv3x3(x, wmod[:1])                                       # shared weights
    assert rel_err(y3.double().cpu().numpy(), F.conv2d(x.double(), wq[0], padding=1).cpu().numpy()) < tol
    yt = modconv.conv2d(x, wmod, transposed=True)
    ytr = torch.stack([F.conv_transpose2d(x[i:i + 1].double(), wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    assert rel_err(yt.double().cpu().numpy(), ytr.cpu().numpy()) < tol


@pytest.mark.parametrize('ci,co,h,w,n', [(256, 128, 64, 64, 4), (64, 96, 70, 52, 4), (32, 256, 33, 40, 3), (128, 128, 128, 128, 1)])
@pytest.mark.parametrize('split', [True, False], ids=['bf16x3', 'exact_fp32'])
def test_transposed_conv_at_layer_sizes(hip_lib, ci, co, h, w, n, split):
    """The fp32 stride-2 transposed convolution (four parity classes in one launch) at the sizes of the backbone's x2 layers, against
    conv_transpose2d in fp64: ragged class grids (odd sizes: the classes differ by a row / column), a channel count that is not a
    multiple of the tile, exact fp32 and bf16x3.  (Written for a halo-slab build of this form, which measured no faster than the generic
    kernel and was not kept — profiles/round3_ac_*; the cases stay.)"""
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(ci + co + h)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    w32 = modconv.modulate_weights(weight, styles, dtype=torch.float32)
    wm = modconv.modulate_weights(weight, styles, dtype=modconv.BF16X3) if split else w32
    y = modconv.conv2d(x, wm, transposed=True, split=split)
    assert y.shape == (n, co, 2 * h + 1, 2 * w + 1)
    wq = w32.double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3).cpu()
    xd = x.double().cpu()
    ref = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    e = rel_err(y.double().cpu().numpy(), ref.numpy())
    print((ci, co, h, w, n, split), e)
    assert e < (1e-5 if split else 3e-6), e


@pytest.mark.parametrize('ci,co,res,transposed', [(256, 256, 64, False), (128, 128, 128, False), (256, 128, 64, True), (64, 96, 70, False)])
def test_bf16x3_formulation_of_the_fp32_convolution(hip_lib, ci, co, res, transposed):
    """P3D_F32_BF16X3: every fp32 product as three bf16 MFMAs of (hi, lo) splits with fp32 accumulation.  Bar: <= 1e-5
    of the output's maximum against an fp64 convolution of the same fp32 operands, per layer (measured 4-5e-6; the exact fp32 MFMA
    kernel: ~2e-6), with the fused epilogue on top."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(ci + res)
    n = 2
    x = _nhwc(torch.randn(n, ci, res, res, device='cuda'))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    w32 = modconv.modulate_weights(weight, styles, dtype=torch.float32)
    w3 = modconv.modulate_weights(weight, styles, dtype=modconv.BF16X3)
    assert w3.dtype == torch.float32 and w3.shape == w32.shape
    wq = w32.double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3).cpu()
    xd = x.double().cpu()
    if transposed:
        y = modconv.conv2d(x, w3, transposed=True, split=True)
        ref = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    else:
        bias, noise, strength = torch.randn(co, device='cuda'), torch.randn(res, res, device='cuda'), torch.tensor(0.3, device='cuda')
        y = modconv.conv2d(x, w3, bias=bias, noise=noise, noise_strength=strength, act=1, gain=2 ** 0.5, split=True)
        ref = torch.stack([F.conv2d(xd[i:i + 1], wq[i], padding=1)[0] for i in range(n)])
        ref = F.leaky_relu(ref + (noise * strength).double().cpu() + bias.double().cpu().view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    e = rel_err(y.double().cpu().numpy(), ref.numpy())
    print((ci, co, res, transposed), e)
    assert e < 1e-5, e


@pytest.mark.parametrize('ci,co,h,w,n,noise,act,clamp', [
    (32, 256, 128, 128, 2, True, 'lrelu', 256.0),      # SR block0.conv0 (one 32-channel chunk)
    (256, 128, 64, 64, 2, True, 'lrelu', 256.0),       # SR block1.conv0's channels (eight chunks), several tiles
    (64, 32, 14, 14, 1, False, 'linear', None),        # exactly one 28 x 28 tile
    (64, 64, 15, 29, 3, True, 'lrelu', 0.5),           # ragged tiles on both axes, odd chunk count below, clamp active
    (96, 96, 33, 20, 2, True, 'lrelu', None),          # three chunks (odd): both slab buffers end a loop
    (32, 32, 3, 5, 2, True, 'lrelu', None),            # image smaller than a tile
])
@pytest.mark.parametrize('taps', [[1, 3, 3, 1], [0.1, 0.45, 0.35, 0.1]], ids=['fir_on_mfma', 'fir_on_valu'])
def test_x2_layer_in_one_kernel(hip_lib, ci, co, h, w, n, noise, act, clamp, taps):
    """csrc/up2_fir.hip: conv_transpose2d(stride 2) + 4x4 FIR (pad 1, gain 4) + noise + bias + act + clamp against the same chain in
    fp64 torch on the fp16-rounded operands (with the fp16 rounding of the transposed conv's output the reference has), and
    against the two-kernel form.  Tolerance: one fp16 rounding of the result + one of the intermediate (2e-3 of the range).
    Two filters: setup_filter([1, 3, 3, 1]) — tap products {1, 3, 9} / 16 are fp16 numbers, the FIR runs as banded-matrix MFMAs — and an
    asymmetric one whose products are not, which takes the kernel's vector-ALU epilogue (fp32 taps)."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(ci + co + h)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda').half())
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    bias = torch.randn(co, device='cuda')
    nz = torch.randn(2 * h, 2 * w, device='cuda') if noise else None
    ns = torch.tensor(0.3, device='cuda') if noise else None
    f = upfirdn2d.setup_filter(taps, device=torch.device('cuda'))
    gain = float(np.sqrt(2)) if act == 'lrelu' else 1.0
    prev = modconv.fuse_up2
    try:
        modconv.fuse_up2 = True
        n0, u0 = _lib.launch_count('conv'), _lib.launch_count('upfirdn2d')
        y1 = modconv.synthesis_layer(x, weight, styles, bias, 2, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=clamp)
        assert _lib.launch_count("conv") == n0 + 2 and _lib.launch_count("upfirdn2d") == u0      # weight modulation + ONE layer launch, no separate FIR
        modconv.fuse_up2 = False
        y0 = modconv.synthesis_layer(x, weight, styles, bias, 2, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=clamp)
    finally:
        modconv.fuse_up2 = prev
    assert y1.shape == (n, co, 2 * h, 2 * w) and y1.dtype == torch.float16 and y1.is_contiguous(memory_format=torch.channels_last)
    wmod = modconv.modulate_weights(weight, styles)
    wq = wmod.double().reshape(n, co, 3, 3, ci).permute(0, 1, 4, 2, 3).cpu()
    xd = x.double().cpu()
    ct = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)]).half().double()
    fd = (f.double().cpu() * 4).flip([0, 1])
    yr = F.conv2d(F.pad(ct, [1, 1, 1, 1]).reshape(n * co, 1, 2 * h + 3, 2 * w + 3), fd[None, None]).reshape(n, co, 2 * h, 2 * w)
    if noise:
        yr = yr + (nz.double().cpu() * 0.3)
    yr = yr + bias.double().cpu().view(1, -1, 1, 1)
    if act == 'lrelu':
        yr = F.leaky_relu(yr, 0.2)
    yr = yr * gain
    if clamp is not None:
        yr = yr.clamp(-clamp, clamp)
    e1, e0 = rel_err(y1.double().cpu().numpy(), yr.numpy()), rel_err(y0.double().cpu().numpy(), yr.numpy())
    print(ci, co, h, w, 'fused', e1, 'two-kernel', e0)
    assert e1 < 2e-3 and e0 < 2e-3


def test_style_affines_in_one_launch(hip_lib):
    """p3d_fc_multi (every style affine of a synthesis network in one launch) equals the per-layer kernel and the torch formulation."""
    from pix2pix3d_amd.training.networks_stylegan2 import FullyConnectedLayer
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(1)
    layers = [FullyConnectedLayer(512, o, bias_init=1).cuda().requires_grad_(False) for o in (512, 512, 256, 128, 96, 3, 70)]
    layers.append(FullyConnectedLayer(64, 32, activation='lrelu', lr_multiplier=0.01).cuda().requires_grad_(False))
    ws = torch.randn(4, len(layers), 512, device='cuda')
    xs = [ws[:, k] for k in range(len(layers) - 1)] + [torch.randn(4, 64, device='cuda')]
    scales = [1, 1, 0.5, 1, 2.0, 1 / 16, 1, 1]
    outs = modconv.fc_multi([(x, l, sc) for x, l, sc in zip(xs, layers, scales)])
    for x, l, sc, y in zip(xs, layers, scales, outs):
        y1 = modconv.fc(x, l.weight, l.bias, l.weight_gain, l.bias_gain, l.activation, sc)
        assert torch.equal(y, y1)
        yr = torch.nn.functional.linear(x.double(), l.weight.double() * l.weight_gain, l.bias.double() * l.bias_gain)
        if l.activation == 'lrelu':
            yr = torch.nn.functional.leaky_relu(yr, 0.2) * np.sqrt(2)
        assert rel_err(y.double().cpu().numpy(), (yr * sc).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('n,ci,co,res,up,act', [(4, 512, 512, 4, 1, 'lrelu'), (4, 512, 512, 4, 2, 'lrelu'), (4, 512, 512, 16, 1, 'lrelu'), (3, 256, 128, 32, 2, 'lrelu'),
                                                (2, 64, 96, 8, 1, 'linear'), (4, 128, 64, 32, 1, 'lrelu'), (5, 96, 160, 12, 2, 'linear'),
                                                (8, 512, 256, 4, 1, 'lrelu'), (4, 512, 512, 32, 2, 'lrelu'), (6, 160, 128, 6, 1, 'lrelu')])   # (8 x 512 scales per tile: the table overflows -> own pass)
def test_shared_weight_form_of_the_low_resolution_layers(hip_lib, n, ci, co, res, up, act):
    """x * styles -> convolution with the UNMODULATED weights (batch folded into the GEMM rows) -> * demodulation coefficients
    (modconv.use_shared_weights) against the per-image-weights route and against fp64 torch: the same function, <= 1e-5 of the range."""
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(n * ci + res)
    x = _nhwc(torch.randn(n, ci, res, res, device='cuda'))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    bias = torch.randn(co, device='cuda')
    nz = torch.randn(res * up, res * up, device='cuda')
    ns = torch.tensor(0.2, device='cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
    gain = float(np.sqrt(2)) if act == 'lrelu' else 1.0
    prev = modconv.shared_weight_max_pixels
    try:
        modconv.shared_weight_max_pixels = 1024
        assert modconv.use_shared_weights(x, weight, styles)
        y1 = modconv.synthesis_layer(x, weight, styles, bias, up, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=None)
        # x * styles inside the convolution kernel (p3d_conv2d_nhwc_scaled_in, the default) against the pass of its own it replaces: the same rounded
        # fp32 product enters the same split and the same MFMAs -> bit-identical
        prev_f, modconv.fuse_input_scale = modconv.fuse_input_scale, False
        try:
            y1_two = modconv.synthesis_layer(x, weight, styles, bias, up, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=None)
        finally:
            modconv.fuse_input_scale = prev_f
        assert prev_f and torch.equal(y1, y1_two)
        modconv.shared_weight_max_pixels = 0
        y0 = modconv.synthesis_layer(x, weight, styles, bias, up, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=None)
    finally:
        modconv.shared_weight_max_pixels = prev
    wq = weight.double().cpu()[None] * styles.double().cpu()[:, None, :, None, None]
    wq = wq * (wq.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    xd = x.double().cpu()
    if up == 1:
        yr = torch.stack([F.conv2d(xd[i:i + 1], wq[i], padding=1)[0] for i in range(n)])
    else:
        ct = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
        fd = (f.double().cpu() * 4).flip([0, 1])
        yr = F.conv2d(F.pad(ct, [1, 1, 1, 1]).reshape(n * co, 1, 2 * res + 3, 2 * res + 3), fd[None, None]).reshape(n, co, 2 * res, 2 * res)
    yr = yr + nz.double().cpu() * 0.2 + bias.double().cpu().view(1, -1, 1, 1)
    if act == 'lrelu':
        yr = F.leaky_relu(yr, 0.2)
    yr = yr * gain
    e1, e0, e01 = rel_err(y1.double().cpu().numpy(), yr.numpy()), rel_err(y0.double().cpu().numpy(), yr.numpy()), rel_err(y1.cpu().numpy(), y0.cpu().numpy())
    print(n, ci, co, res, up, 'shared', e1, 'per-image', e0, 'between', e01)
    assert y1.shape == y0.shape and e1 < 1e-5 and e0 < 1e-5


@pytest.mark.parametrize('img_channels,in_ch,res,out_ch', [(3, 256, 256, 128), (6, 128, 176, 128), (1, 64, 160, 128),
                                                           (3, 128, 256, 256), (6, 256, 176, 256)])      # 256: both channel blocks of a patch in one work-group (SR block0)
def test_last_conv_and_torgb_in_one_launch(hip_lib, img_channels, in_ch, res, out_ch):
    """SynthesisBlock.conv1 + ToRGB + skip-image sum from one launch (p3d_conv3x3_torgb_f16) against the three-launch form: the same x
    (bit-identical: the convolution is untouched) and the same skip image up to the ToRGB weights' precision."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.training.networks_stylegan2 import SynthesisBlock
    from pix2pix3d_amd.torch_utils.ops import modconv
    torch.manual_seed(img_channels)
    blk = SynthesisBlock(in_ch, out_ch, w_dim=64, resolution=res, img_channels=img_channels, is_last=out_ch == 128, use_fp16=True, conv_clamp=256,
                         fp16_channels_last=True).cuda().eval().requires_grad_(False)
    blk.torgb.bias.normal_(); blk.conv1.bias.normal_()
    n = 2
    x = torch.randn(n, in_ch, res // 2, res // 2, device='cuda')
    img = torch.randn(n, img_channels, res // 2, res // 2, device='cuda')
    ws = torch.randn(n, 3, 64, device='cuda')
    outs = {}
    prev = modconv.fuse_torgb
    try:
        for flag in (True, False):
            modconv.fuse_torgb = flag
            c0 = modconv.fused_torgb_calls
            with torch.no_grad():
                outs[flag] = blk(x, img.clone(), ws, noise_mode='none')
            outs[flag] += (modconv.fused_torgb_calls - c0,)
    finally:
        modconv.fuse_torgb = prev
    (x1, i1, l1), (x0, i0, l0) = outs[True], outs[False]
    assert (l1, l0) == (1, 0)                                                  # the fused kernel ran / did not run
    assert torch.equal(x1, x0)                                                 # same kernel, same K order
    # both contract fp16 activations with the modulated weights rounded to fp16 (as the reference's fp16 layer does); fp32 summation order differs
    assert rel_err(i1.cpu().numpy(), i0.cpu().numpy()) < 1e-5
    # the form the super-resolution heads use for their last block (SynthesisBlock.forward(_x_dead=True)): x is returned to nobody, so the launch stores
    # no activations at all — swapped MFMA operands, the ToRGB contracted straight from the accumulator registers (conv3x3_h2_f16_kernel<true>) —
    # and the skip image must come out the same: the same fp16-rounded activations times the same fp16-rounded weights, another fp32 summation order
    c0 = modconv.fused_torgb_calls
    with torch.no_grad():
        xd, idd = blk(x, img.clone(), ws, noise_mode='none', _x_dead=True)
    assert xd is None and modconv.fused_torgb_calls == c0 + 1
    assert rel_err(idd.cpu().numpy(), i1.cpu().numpy()) < 1e-5 and rel_err(idd.cpu().numpy(), i0.cpu().numpy()) < 1e-5
    assert torch.isfinite(idd).all() and idd.shape == i1.shape


@pytest.mark.parametrize('ci,co,h,w,n,noise,act,clamp', [
    (256, 128, 128, 128, 2, True, 'lrelu', None),      # backbone b256.conv0
    (512, 256, 64, 64, 2, True, 'lrelu', None),        # backbone b128.conv0 (sixteen chunks)
    (64, 32, 14, 14, 1, False, 'linear', None),        # exactly one tile
    (96, 96, 33, 20, 3, True, 'lrelu', 0.5),           # ragged tiles, three chunks, clamp active
    (32, 64, 6, 4, 2, True, 'lrelu', None),            # image smaller than a tile
])
def test_x2_layer_in_one_kernel_fp32_as_bf16x3(hip_lib, ci, co, h, w, n, noise, act, clamp):
    """The bf16x3 build of csrc/up2_fir.hip (fp32 tensors) against the same chain in fp64 torch and against the two-kernel form:
    the accuracy class of every other bf16x3 layer (1e-5 of the range)."""
    from pix2pix3d_amd import _lib
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    torch.manual_seed(ci + co + h)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    bias = torch.randn(co, device='cuda')
    nz = torch.randn(2 * h, 2 * w, device='cuda') if noise else None
    ns = torch.tensor(0.3, device='cuda') if noise else None
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=torch.device('cuda'))
    gain = float(np.sqrt(2)) if act == 'lrelu' else 1.0
    prev = (modconv.fuse_up2_f32_min_res, modconv.shared_weight_max_pixels)
    try:
        modconv.shared_weight_max_pixels = 0
        modconv.fuse_up2_f32_min_res = 1
        u0 = _lib.launch_count('upfirdn2d')
        y1 = modconv.synthesis_layer(x, weight, styles, bias, 2, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=clamp)
        assert _lib.launch_count('upfirdn2d') == u0                              # no separate FIR
        modconv.fuse_up2_f32_min_res = 1 << 30
        y0 = modconv.synthesis_layer(x, weight, styles, bias, 2, f, noise_const=nz, noise_strength=ns, act=act, act_gain=gain, clamp=clamp)
    finally:
        modconv.fuse_up2_f32_min_res, modconv.shared_weight_max_pixels = prev
    assert y1.shape == (n, co, 2 * h, 2 * w) and y1.dtype == torch.float32 and y1.is_contiguous(memory_format=torch.channels_last)
    wq = weight.double().cpu()[None] * styles.double().cpu()[:, None, :, None, None]
    wq = wq * (wq.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    xd = x.double().cpu()
    ct = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    fd = (f.double().cpu() * 4).flip([0, 1])
    yr = F.conv2d(F.pad(ct, [1, 1, 1, 1]).reshape(n * co, 1, 2 * h + 3, 2 * w + 3), fd[None, None]).reshape(n, co, 2 * h, 2 * w)
    if noise:
        yr = yr + (nz.double().cpu() * 0.3)
    yr = yr + bias.double().cpu().view(1, -1, 1, 1)
    if act == 'lrelu':
        yr = F.leaky_relu(yr, 0.2)
    yr = yr * gain
    if clamp is not None:
        yr = yr.clamp(-clamp, clamp)
    e1, e0 = rel_err(y1.double().cpu().numpy(), yr.numpy()), rel_err(y0.double().cpu().numpy(), yr.numpy())
    print(ci, co, h, w, 'fused', e1, 'two-kernel', e0)
    tol = 1e-5 if clamp is None else 1e-4                                        # (a clamp at 0.5 shrinks the range the error is measured against)
    assert e1 < tol and e0 < tol


@pytest.fixture(params=['auto', 'presplit'])
def f32_x6(request):
    """bf16x6 on; 'presplit' forces conv3x3_halo_x6p_kernel on every 3x3 'same' layer the halo-slab route takes (P3D_X6_PRESPLIT=2: the library reads the switch at every
    call) — by itself it takes the launches of at least 256 work-groups, which few test geometries are."""
    from pix2pix3d_amd.torch_utils.ops import modconv
    old, prev = modconv.f32_x6, os.environ.get('P3D_X6_PRESPLIT')
    modconv.f32_x6 = True
    if request.param == 'presplit':
        os.environ['P3D_X6_PRESPLIT'] = '2'
    yield modconv
    modconv.f32_x6 = old
    if prev is None:
        os.environ.pop('P3D_X6_PRESPLIT', None)
    else:
        os.environ['P3D_X6_PRESPLIT'] = prev


@pytest.mark.parametrize('ci,co,h,w,n,k,mode', [
    (256, 256, 64, 64, 2, 3, 'same'),          # halo-slab kernel
    (64, 96, 70, 52, 2, 3, 'same'),            # ragged tiles, channel count that is no multiple of the tile
    (32, 64, 40, 40, 3, 3, 'same'),            # the 64-channel tile form
    (96, 160, 26, 34, 2, 3, 'same'),           # three 32-channel K rows (the pre-split kernel walks them as six 16-channel chunks), two column blocks, ragged patches
    (128, 48, 8, 16, 1, 3, 'same'),            # ONE 8 x 16 patch: every slab pixel outside it is padding
    (512, 512, 16, 16, 4, 3, 'same'),          # the split-K schedule
    (256, 128, 64, 64, 2, 3, 'transposed'),    # four parity classes, class-major order
    (32, 256, 33, 40, 3, 3, 'transposed'),
    (128, 130, 24, 24, 2, 1, 'same'),          # 1x1 on the matrix pipe
    (128, 128, 33, 33, 2, 3, 'down'),          # valid, stride 2
])
def test_bf16x6_formulation_of_the_fp32_convolution(hip_lib, f32_x6, ci, co, h, w, n, k, mode):
    """P3D_F32_BF16X6: plain fp32 activations and weights, every product as six bf16 MFMAs of three-piece splits (made in registers, or — 3x3 'same' layers on
    conv3x3_halo_x6p_kernel — once per work-group on the way into LDS).  Bar: the error
    class of the exact fp32 MFMA kernel (3e-6 of the output's maximum against an fp64 convolution of the same fp32 operands) and within 2x of
    what that kernel measures on the same inputs — fp32-accurate, unlike bf16x3 (1e-5)."""
    modconv = f32_x6
    torch.manual_seed(ci + co + h + k)
    x = _nhwc(torch.randn(n, ci, h, w, device='cuda'))
    weight = torch.randn(co, ci, k, k, device='cuda')
    styles = torch.randn(n, ci, device='cuda') + 1
    w32 = modconv.modulate_weights(weight, styles, demodulate=(k == 3), dtype=torch.float32)
    wq = w32.double().reshape(n, co, k, k, ci).permute(0, 1, 4, 2, 3).cpu()
    xd = x.double().cpu()
    kw = dict(transposed=True) if mode == 'transposed' else (dict(down=2) if mode == 'down' else dict(bias=torch.randn(co, device='cuda'), act=1, gain=2 ** 0.5))

    y6 = modconv.conv2d(x, w32, **kw)
    modconv.f32_x6 = False
    y1 = modconv.conv2d(x, w32, **kw)
    modconv.f32_x6 = True
    if mode == 'transposed':
        ref = torch.stack([F.conv_transpose2d(xd[i:i + 1], wq[i].transpose(0, 1), stride=2)[0] for i in range(n)])
    elif mode == 'down':
        ref = torch.stack([F.conv2d(xd[i:i + 1], wq[i], stride=2)[0] for i in range(n)])
    else:
        ref = torch.stack([F.conv2d(xd[i:i + 1], wq[i], padding=k // 2)[0] for i in range(n)])
        ref = F.leaky_relu(ref + kw['bias'].double().cpu().view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    assert y6.shape == ref.shape
    e6 = rel_err(y6.double().cpu().numpy(), ref.numpy())
    e1 = rel_err(y1.double().cpu().numpy(), ref.numpy())
    print((ci, co, h, w, n, k, mode), 'bf16x6', e6, 'exact', e1)
    assert e6 < 3e-6 and e6 < 2 * e1 + 1e-7, (e6, e1)
    assert not torch.equal(y6, y1)                       # (the other arithmetic did run)


def test_bf16x6_non_finite_operands(hip_lib, f32_x6):
    """What csrc/bf16_split.h states about operands outside bf16's range: an +-inf (or a finite value >= 3.39e38, which rounds to inf as bf16) makes every output it reaches
    non-finite under both arithmetics — NaN under bf16x6 (inf - inf in the residual), +-inf / NaN under the f32-input MFMA — and touches nothing else: the other images of the
    batch and the pixels outside the 3 x 3 footprint stay in the exact kernel's error class."""
    modconv = f32_x6
    torch.manual_seed(5)
    n, ci, co, h, w = 3, 64, 64, 24, 24
    x = torch.randn(n, ci, h, w, device='cuda')
    x[0, 7, 10, 11] = float('inf')
    x[1, 3, 5, 5] = 3.4e38                                      # finite in fp32 (max 3.4028e38), inf as bf16 (anything above 3.3961e38 rounds up)
    x = _nhwc(x)
    weight = torch.randn(co, ci, 3, 3, device='cuda')
    w32 = modconv.modulate_weights(weight, torch.ones(n, ci, device='cuda'), demodulate=True, dtype=torch.float32)
    y6 = modconv.conv2d(x, w32)
    modconv.f32_x6 = False
    y1 = modconv.conv2d(x, w32)
    modconv.f32_x6 = True
    foot = torch.zeros(n, 1, h, w, dtype=torch.bool, device='cuda')
    foot[0, 0, 9:12, 10:13] = True
    foot[1, 0, 4:7, 4:7] = True
    foot = foot.expand(n, co, h, w)
    assert not bool(torch.isfinite(y6[foot]).any()) and bool(torch.isnan(y6[foot]).all())
    assert not bool(torch.isfinite(y1[0:1][foot[0:1]]).any())                  # the exact kernel: +-inf (or NaN where +inf and -inf meet)
    clean6, clean1 = y6[~foot], y1[~foot]
    assert bool(torch.isfinite(clean6).all()) and bool(torch.isfinite(clean1).all())
    assert float((clean6 - clean1).abs().max()) <= 3e-6 * float(clean1.abs().max())
