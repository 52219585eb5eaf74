"""Host-side decisions of the native routes that need no GPU: which tensors the per-channel kernels accept, the separable form of the
x2 layers' FIR that p3d_up2_fir_f16 takes, and that CPU tensors stay on the tensor-op formulation with the reference's values."""
import numpy as np
import torch

from conftest import rel_err


def test_separable_fir_of_the_x2_layers():
    from pix2pix3d_amd.torch_utils.ops import modconv, upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    taps = modconv._separable_fir(f)
    fy, fx = np.array(taps[:4]), np.array(taps[4:8])
    # out[oy][ox] = sum fy[a] fx[b] ct[oy - 1 + a][ox - 1 + b] must be upfirdn2d's flipped filter times the gain 4 (conv2d_resample.py:128)
    want = (f.double().numpy() * 4)[::-1, ::-1]
    assert np.allclose(np.outer(fy, fx), want, rtol=1e-6)
    assert modconv._separable_fir(f) is taps                                   # cached per filter tensor
    assert modconv._separable_fir(torch.eye(4)) is None                        # not an outer product: the two-kernel form takes it
    assert modconv._separable_fir(upfirdn2d.setup_filter([1, 2, 1])) is None   # not 4 x 4


def test_per_channel_kernels_take_only_dense_device_tensors():
    from pix2pix3d_amd.torch_utils.ops import bcast
    x = torch.randn(2, 8, 4, 4)
    assert bcast.layout(x) is None                                             # CPU
    assert not bcast.scale_channels_supported(x, torch.randn(2, 8))
    assert not bcast.fma_supported(x, torch.randn(2, 8, 1, 1), torch.randn(2, 1, 4, 4))
    assert not bcast.bias_sum_supported(x, 1)


def test_fma_and_bias_act_on_cpu_keep_the_reference_formulation():
    from pix2pix3d_amd.torch_utils.ops import fma, bias_act
    torch.manual_seed(0)
    a, b, c = (torch.randn(2, 8, 4, 4, dtype=torch.float64, requires_grad=True), torch.randn(2, 8, 1, 1, dtype=torch.float64, requires_grad=True),
               torch.randn(1, 1, 4, 4, dtype=torch.float64, requires_grad=True))
    y = fma.fma(a, b, c)
    g = torch.randn_like(y)
    ga, gb, gc = torch.autograd.grad(y, [a, b, c], g)
    assert torch.allclose(y, a * b + c) and torch.allclose(ga, g * b) and torch.allclose(gb, (g * a).sum([2, 3], keepdim=True))
    assert torch.allclose(gc, g.sum([0, 1], keepdim=True))
    x = torch.randn(3, 8, 5, 5, requires_grad=True)
    bb = torch.randn(8, requires_grad=True)
    yb = bias_act.bias_act(x, bb, act='lrelu')
    gx, gbias = torch.autograd.grad(yb, [x, bb], torch.ones_like(yb))
    ref = torch.nn.functional.leaky_relu(x + bb.reshape(1, -1, 1, 1), 0.2) * np.sqrt(2)
    assert rel_err(yb.detach().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(gbias.numpy(), torch.autograd.grad(ref, bb, torch.ones_like(ref))[0].numpy()) < 1e-6
