"""The import-path drop-in: after install() the reference's absolute names resolve to this package's mirrors."""
import subprocess
import sys

from conftest import ROOT


def test_install_dropin_aliases_reference_import_paths():
    code = r'''
import sys
sys.path.insert(0, %r)
import pix2pix3d_amd
names = pix2pix3d_amd.install_dropin()
assert 'torch_utils.ops.bias_act' in names and 'training.triplane_cond' in names
from torch_utils.ops import bias_act, upfirdn2d, conv2d_gradfix, conv2d_resample, fma, grid_sample_gradfix
from torch_utils import custom_ops, misc, persistence
from training.volumetric_rendering.renderer import ImportanceRenderer
from training.volumetric_rendering.ray_sampler import RaySampler
from training.networks_stylegan2 import SynthesisBlock, DiscriminatorBlock, FullyConnectedLayer
from training.triplane_cond import TriPlaneSemanticEntangleGenerator, TriPlaneGenerator
import dnnlib
assert bias_act.__name__.startswith('pix2pix3d_amd.')
assert dnnlib.EasyDict(a=1).a == 1
custom_ops.verbosity = 'none'                       # train.py:54 writes this
conv2d_gradfix.enabled = True                       # training_loop.py:281
grid_sample_gradfix.enabled = False                 # training_loop.py:282
cls = dnnlib.util.get_obj_by_name('training.superresolution.SuperresolutionHybrid8XDC')
assert cls.__module__.startswith('pix2pix3d_amd.')
import legacy                                       # applications/generate_samples.py:16
assert legacy.load_network_pkl.__module__ == 'pix2pix3d_amd.legacy'
print('ok')
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]
