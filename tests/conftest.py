import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    """max|a-b| / max|b| — the parity statistic of SURVEY §8(c)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


_measured = {}


def record_error(name, value):
    """Keep what a parity test measured (gpurun_out/parity_errors.json, rewritten on every call): the tolerances of the fp16 legs are set from
    this record (3 x the worst value), not from a guess."""
    import json
    _measured[name] = value
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_errors.json'), 'w') as f:
            json.dump(_measured, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope='session')
def hip_lib():
    """The loaded kernel library; GPU tests fail (not skip) if it is missing."""
    import torch
    from pix2pix3d_amd import _lib
    assert torch.cuda.is_available(), 'gpu-marked test running without a GPU'
    return _lib.lib()
