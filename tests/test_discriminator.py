"""DualDiscriminator / SingleDiscriminator in the 'Dboth' training phase on real images (loss.py:327-367) against the reference:
logits, the R1 gradients w.r.t. both input images (double-backward through conv2d_gradfix under no_weight_gradients) and the
parameter gradients of softplus(-logits) + 5 * r1.  Goldens: tests/golden/make_golden.py group ``discriminator``."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from model_cases import weights

CASES = dict(dual=dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=64, img_channels=3, channel_base=1024,
                       channel_max=32, num_fp16_res=0, conv_clamp=None, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={},
                       epilogue_kwargs=dict(mbstd_group_size=2)),
             dual_clamp=dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=32, img_channels=1, channel_base=512,
                             channel_max=16, num_fp16_res=0, conv_clamp=0.5, architecture='resnet', epilogue_kwargs=dict(mbstd_group_size=4, mbstd_num_channels=2)),
             single=dict(class_name='training.dual_discriminator.SingleDiscriminator', c_dim=0, img_resolution=32, img_channels=3, channel_base=512,
                         channel_max=16, num_fp16_res=0, conv_clamp=None))


def _field_close(a, ref, tol, robust):
    """max-norm parity of a gradient field; ``robust``: relative L2 error <= 2 tol and at most 1 % of the elements off by more than tol of the
    field's maximum (see test_discriminator_dboth_phase_on_the_native_convolutions for why the bf16x3 leg needs that form)."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    if not robust:
        return rel_err(a, ref) < tol, rel_err(a, ref)
    l2 = float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30))
    frac = float((np.abs(a - ref) > tol * np.abs(ref).max()).mean())
    frac_max = 1e-2 if a.size >= 20000 else 5e-2            # one flipped unit's neighbourhood is ~150 elements: 5 % of the 3 072-element raw-image field
    return (l2 < 2 * tol and frac <= frac_max and rel_err(a, ref) < 10 * tol), (l2, frac, rel_err(a, ref))


def _dboth(name, device, tol, grad_tol=None, robust=False):
    gt = tol if grad_tol is None else grad_tol
    from pix2pix3d_amd import dnnlib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    g = {k.split('.', 1)[1]: v for k, v in load_golden('discriminator').items() if k.startswith(name + '.')}
    torch.manual_seed(0)
    D = dnnlib.util.construct_class_by_name(**CASES[name]).train().requires_grad_(True)
    weights.seed_module(D, seed=9)
    D = D.to(device)
    img = {'image': torch.tensor(g['image']).to(device).requires_grad_(True)}
    if name != 'single':
        img['image_raw'] = torch.tensor(g['image_raw']).to(device).requires_grad_(True)
    c = torch.tensor(g['c']).to(device) if name != 'single' else None
    logits = D(img, c)
    assert rel_err(logits.detach().cpu().numpy(), g['logits']) < tol
    with conv2d_gradfix.no_weight_gradients():
        grads = torch.autograd.grad(outputs=[logits.sum()], inputs=list(img.values()), create_graph=True, only_inputs=True)
    r1 = sum(gr.square().sum([1, 2, 3]) for gr in grads)
    (torch.nn.functional.softplus(-logits) + r1 * 5).mean().backward()
    ok, stat = _field_close(grads[0].detach().cpu().numpy(), g['g_img'], gt, robust)
    assert ok, ('g_img', stat)
    if name != 'single':
        ok, stat = _field_close(grads[1].detach().cpu().numpy(), g['g_raw'], gt, robust)
        assert ok, ('g_raw', stat)
    assert rel_err(r1.detach().cpu().numpy(), g['r1']) < gt
    params = dict(D.named_parameters())
    names = [n for n, p in params.items() if p.grad is not None]
    assert names == list(g['grad_names'])
    norms = np.array([float(params[n].grad.double().norm()) for n in names])
    assert np.abs(norms - g['grad_norms']).max() / g['grad_norms'].max() < gt
    assert np.all(np.abs(norms - g['grad_norms']) <= gt * 10 * np.maximum(g['grad_norms'], 1e-3 * g['grad_norms'].max()))
    assert rel_err(params[names[0]].grad.reshape(-1)[:64].cpu().numpy(), g['grad_head']) < gt


@pytest.mark.parametrize('name', list(CASES))
def test_discriminator_dboth_phase_matches_reference_cpu(name):
    _dboth(name, 'cpu', 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_discriminator_dboth_phase_matches_reference_device(name):
    _dboth(name, 'cuda', 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_discriminator_dboth_phase_on_the_native_convolutions(hip_lib, name):
    """The same phase as the training loop runs it (conv2d_gradfix.enabled = True, training_loop.py:281): every convolution, its
    data gradient, the R1 double-backward and the weight gradients go through libp3d_hip.so — none through torch's operators.

    Two legs.  The default (exact fp32 products): everything within 2e-3 of the reference records, max-norm.  The bf16x3 opt-in
    (P3D_TRAIN_BF16X3=1): every convolution is within 1e-5 of fp32 (tests/gpu_probe_split_d.py compares them call by call), logits within
    2e-3, r1 / every parameter-gradient norm / the gradient head within 1e-2 (round 2 allowed 0.1), and the two R1 gradient FIELDS in the
    robust form of ``_field_close``: relative L2 <= 2e-2, max-norm <= 0.1 and at most 1 % of the elements (5 % for the 3 072-element raw-image
    field) off by more than 1e-2 of the field's maximum (measured on 'dual': image field L2 1.1e-2, 0.6 % of the elements, max-norm 3e-2; raw
    field L2 1.2e-2, 1.3 %, 4.4e-2; 'dual_clamp' and 'single' pass the same statistic at 1e-2).
    Why a field is not held in max-norm on that leg: a leaky-ReLU layer of these networks has up to 262 144 pre-activations of range ~1.5,
    the closest to zero sits at 4e-8 .. 5e-7 of the range for EVERY input seed (tests/golden/seed_search_discriminator.py lists seeds
    31..45: there is no seed without such units), so a 5e-6 perturbation carries ~10 of the ~800 000 units of the high-resolution layers
    across zero per pass; the slope jumps 0.2 -> 1 there and the input-gradient of the neighbourhood behind each such unit (50-150 of the
    field's 49 152 elements) moves by a few per cent of the field's maximum — a property of the function at those points, not of the
    kernels, and the reason training defaults to exact fp32."""
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    prev_split = conv2d_gradfix.split_bf16
    c0 = dict(conv2d_gradfix.native_calls)
    try:
        conv2d_gradfix.split_bf16 = False
        _dboth(name, 'cuda', 2e-3)
        conv2d_gradfix.split_bf16 = True
        _dboth(name, 'cuda', 2e-3, grad_tol=1e-2, robust=True)
    finally:
        conv2d_gradfix.enabled, conv2d_gradfix.split_bf16 = prev, prev_split
    assert conv2d_gradfix.native_calls['aten'] == c0['aten'], conv2d_gradfix.native_calls
    assert conv2d_gradfix.native_calls['forward'] > c0['forward'] + 10 and conv2d_gradfix.native_calls['weight_grad'] > c0['weight_grad'] + 5


# ------------------------------------------------------------------------------------------------------------------------------------------------
# Config 3's discriminators at their REAL size (512^2, channel_base 32768, conv_clamp 256, batch 2): D on 3 image channels and D_semantic on 3 + 6
# (training_loop.py:308), the 'Dboth' phase on real input with gamma 5 — goldens from the reference on the CPU (make_golden.py discriminator_full).
def _full_cases():
    import importlib.util, os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('p3d_disc_full_cases', os.path.join(ROOT, 'tests', 'golden', 'disc_full_cases.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def _dboth_full(tag, ch, device, tol, fp16=False, grad_tol=None):
    gt = tol if grad_tol is None else grad_tol
    from pix2pix3d_amd import dnnlib
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    cases = _full_cases()
    g = {k.split('.', 1)[1]: v for k, v in load_golden('discriminator_full').items() if k.startswith(tag + '.')}
    torch.manual_seed(0)
    D = dnnlib.util.construct_class_by_name(**cases.full_discriminator_kwargs(ch)).train().requires_grad_(True)
    weights.seed_discriminator(D, seed=9 + ch)
    D = D.to(device)
    img, raw, c = (t.to(device) for t in cases.full_discriminator_inputs(ch))
    x = {'image': img.clone().requires_grad_(True), 'image_raw': raw.clone().requires_grad_(True)}
    logits = D(x, c) if fp16 else D(x, c, force_fp32=True)
    with conv2d_gradfix.no_weight_gradients():
        g_img, g_raw = torch.autograd.grad(outputs=[logits.sum()], inputs=[x['image'], x['image_raw']], create_graph=True, only_inputs=True)
    r1 = g_img.square().sum([1, 2, 3]) + g_raw.square().sum([1, 2, 3])
    (torch.nn.functional.softplus(-logits) + r1 * (5 / 2)).mean().backward()
    e = {'logits': float(np.abs(logits.detach().float().cpu().numpy() - g['logits']).max() / max(float(np.abs(g['logits']).max()), 0.1)),
         'r1': rel_err(r1.detach().cpu().numpy(), g['r1'])}
    for name, field, tile in (('g_img', g_img, 32), ('g_raw', g_raw, 8)):
        s, m = cases.tile_stats(field.detach().float().cpu(), tile)             # every element of the field is in one tile's sum and abs-max
        e[name + '.norm'] = abs(float(field.detach().double().norm()) - float(g[name + '_norm'])) / float(g[name + '_norm'])
        e[name + '.tile_max'] = float(np.abs(m.numpy() - g[name + '_tile_max']).max() / g[name + '_tile_max'].max())
        e[name + '.tile_sum'] = float(np.abs(s.numpy() - g[name + '_tile_sum']).max() / max(np.abs(g[name + '_tile_sum']).max(), tile * g[name + '_tile_max'].max()))
    e['g_img.crop'] = rel_err(g_img.detach()[:, :, 240:272, 240:272].float().cpu().numpy(), g['g_img_crop'])
    params = dict(D.named_parameters())
    names = [n for n, p in params.items() if p.grad is not None]
    assert names == list(g['grad_names'])
    norms = np.array([float(params[n].grad.double().norm()) for n in names])
    e['grad_norms'] = float(np.abs(norms - g['grad_norms']).max() / g['grad_norms'].max())
    e['grad_norms.each'] = float((np.abs(norms - g['grad_norms']) / np.maximum(g['grad_norms'], 1e-3 * g['grad_norms'].max())).max())
    e['heads'] = max(rel_err(params[nm].grad.reshape(-1)[:64].float().cpu().numpy(), g[f'h{j}']) for j, nm in enumerate(g['head_names'].tolist()))
    print(tag, device, 'fp16-top-4' if fp16 else 'fp32', {k: float(f'{v:.2e}') for k, v in e.items()})
    # Scalars (logits, penalty, field and gradient NORMS) to the leg's tolerance.  Local statistics of the R1 gradient FIELDS (tile maxima and sums, the
    # crop) to 10x that (measured on an MI355X, fp32 leg: norms 1e-6 .. 7e-6, tile maxima <= 5e-3, tile sums <= 2.3e-3; fp16 leg: penalty 5e-3, norms 6e-3,
    # parameter-gradient norms 4e-3, tile statistics 0.07 .. 0.11, crop 0.15): the field is piecewise constant in the 16.8 M leaky-ReLU pre-activations of the 512^2 layers, and a rounding-level difference
    # in a summation order carries a few units across zero (slope 0.2 <-> 1), moving the field behind each by a few per cent of its local value —
    # the same property of the function test_discriminator_dboth_phase_on_the_native_convolutions documents for the small instances.
    assert e['logits'] < tol and e['r1'] < gt and e['g_img.norm'] < gt and e['g_raw.norm'] < gt, e
    assert all(e[k] < 10 * gt for k in ('g_img.tile_max', 'g_raw.tile_max', 'g_img.tile_sum', 'g_raw.tile_sum', 'g_img.crop')), e
    assert e['grad_norms'] < gt and e['grad_norms.each'] < 10 * gt and e['heads'] < 5 * gt, e
    return e


@pytest.mark.parametrize('tag,ch', [('d', 3), ('dsem', 9)])
def test_full_size_discriminators_match_reference_cpu(tag, ch):
    _dboth_full(tag, ch, 'cpu', 5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,ch', [('d', 3), ('dsem', 9)])
def test_full_size_discriminators_on_the_native_convolutions(hip_lib, tag, ch):
    """Same phase on the device through libp3d_hip.so (conv2d_gradfix.enabled): the all-fp32 leg (force_fp32: the function the CPU reference computes)
    within 2e-3 everywhere; then the configuration training uses — fp16 top-4 blocks with fp32 accumulation, conv_clamp 256 — at the fp16 class
    (logits 2e-2 of their scale; R1 penalties, field norms, tile sums and parameter-gradient norms 3e-2; pointwise field statistics 0.3: fp16 storage of
    512^2 x 64..512-channel activations through four blocks and a double backward)."""
    from pix2pix3d_amd.torch_utils.ops import conv2d_gradfix
    prev, conv2d_gradfix.enabled = conv2d_gradfix.enabled, True
    c0 = dict(conv2d_gradfix.native_calls)
    try:
        _dboth_full(tag, ch, 'cuda', 2e-3)
        _dboth_full(tag, ch, 'cuda', 2e-2, fp16=True, grad_tol=3e-2)
    finally:
        conv2d_gradfix.enabled = prev
    assert conv2d_gradfix.native_calls['aten'] == c0['aten'], conv2d_gradfix.native_calls
    assert conv2d_gradfix.native_calls['forward'] > c0['forward'] + 20 and conv2d_gradfix.native_calls['weight_grad'] > c0['weight_grad'] + 20
