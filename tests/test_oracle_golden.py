"""The numpy oracle (oracle/ops_oracle.py) against vectors recorded from the reference implementation
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import ops_oracle as O

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']


def _opt(v):
    v = float(v)
    return None if v < 0 else v


@pytest.mark.parametrize('act', ACTS)
def test_bias_act_oracle_matches_reference(act):
    g = load_golden('ops_bias_act')
    kw = dict(dim=1, act=act, alpha=_opt(g[f'{act}.alpha']), gain=_opt(g[f'{act}.gain']), clamp=_opt(g[f'{act}.clamp']))
    x, b, dy, ddx = g[f'{act}.x'], g[f'{act}.b'], g[f'{act}.dy'], g[f'{act}.ddx']
    assert rel_err(O.bias_act(x, b, **kw), g[f'{act}.y']) < 1e-12
    dx, db = O.bias_act_grads(x, b, dy, **kw)
    assert rel_err(dx, g[f'{act}.dx']) < 1e-12
    assert rel_err(db, g[f'{act}.db']) < 1e-12
    d2 = O.bias_act_second(x, b, dy, ddx, **kw)
    assert np.abs(d2 - g[f'{act}.d2']).max() < 1e-10


def test_bias_act_oracle_fc_and_nobias():
    g = load_golden('ops_bias_act')
    assert rel_err(O.bias_act(g['fc.x'], g['fc.b'], dim=1, act='lrelu'), g['fc.y']) < 1e-12
    assert rel_err(O.bias_act(g['fc.x'], None, act='swish', gain=0.5, clamp=0.4), g['nob.y']) < 1e-12


def test_upfirdn2d_oracle_matches_reference():
    g = load_golden('ops_upfirdn2d')
    for i in range(int(g['num_cases'])):
        f = g[f'{i}.f']
        f = None if f.size == 0 else f
        y = O.upfirdn2d(g[f'{i}.x'], f, up=g[f'{i}.up'].tolist(), down=g[f'{i}.down'].tolist(), padding=g[f'{i}.pad'].tolist(),
                        flip_filter=bool(g[f'{i}.flip']), gain=float(g[f'{i}.gain']))
        assert y.shape == g[f'{i}.y'].shape, i
        assert rel_err(y, g[f'{i}.y']) < 2e-6, i
    assert rel_err(O.setup_filter([1, 4, 6, 4, 1, 2, 3, 5], gain=2.0, flip_filter=True), g['h.f_sep']) < 1e-6
    assert rel_err(O.setup_filter([1, 3, 3, 1]), g['h.f']) < 1e-7


def test_conv_oracles_match_reference():
    g = load_golden('ops_conv')
    f = g['f']
    for i in range(int(g['num_resample'])):
        k, up, down, flipw = g[f'r{i}.cfg'].tolist()
        y = O.conv2d_resample(g[f'r{i}.x'], g[f'r{i}.w'], f=f, up=up, down=down, padding=g[f'r{i}.pad'].tolist(), flip_weight=bool(flipw))
        assert y.shape == g[f'r{i}.y'].shape, i
        assert rel_err(y, g[f'r{i}.y']) < 5e-6, i
    for i in range(int(g['num_mod'])):
        k, up, demod, fused = g[f'm{i}.cfg'].tolist()
        noise = g[f'm{i}.noise']
        y = O.modulated_conv2d(g[f'm{i}.x'], g[f'm{i}.w'], g[f'm{i}.s'], noise=None if noise.size == 0 else noise, up=up, padding=k // 2,
                               resample_filter=f, demodulate=bool(demod), flip_weight=(up == 1))
        assert rel_err(y, g[f'm{i}.y']) < 5e-6, i


def test_oracle_filtered_lrelu_and_its_sign_tensor_backward():
    """The oracle's filtered_lrelu against the reference's records — forward, and the backward pass driven ONLY by the packed sign
    tensor (the reference's plugin formulation, filtered_lrelu.py:240-270) against the gradient the reference's autograd gives:
    this is what pins the sign codes, their packing, the sign offsets and the adjoint's padding."""
    g = load_golden('ops_filtered_lrelu')
    for i in range(int(g['num'])):
        up, down, flip = g[f'{i}.cfg'].tolist()
        fu, fd = g[f'{i}.fu'], g[f'{i}.fd']
        fu, fd = (None if fu.size == 0 else fu), (None if fd.size == 0 else fd)
        clamp = float(g[f'{i}.clamp'])
        clamp = None if clamp < 0 else clamp
        pad = g[f'{i}.pad'].tolist()
        y, signs = O.filtered_lrelu(g[f'{i}.x'], fu, fd, g[f'{i}.b'], up, down, pad, 1.3, 0.15, clamp, bool(flip), write_signs=True)
        sz = O.filtered_lrelu_sizes(g[f'{i}.x'].shape, fu, fd, up, down, pad)
        assert signs.dtype == np.uint8 and signs.shape[-2:] == (sz['sh'], sz['sw_bytes']) and sz['sw_bytes'] % 4 == 0
        assert rel_err(y, g[f'{i}.y']) < 1e-6
        gx = O.filtered_lrelu_backward(g[f'{i}.gy'], fu, fd, g[f'{i}.x'].shape, signs, up, down, pad, 1.3, 0.15, bool(flip))
        assert rel_err(gx, g[f'{i}.gx']) < 1e-6
        codes = O.unpack_signs(signs, *np.meshgrid(np.arange(sz['sw_active']), np.arange(sz['sh']), indexing='xy'))
        assert set(np.unique(codes)) <= {0, 1, 2} and (clamp is not None or 2 not in codes)
