"""The C-ABI shared library: loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = re.sub(r'/\*.*?\*/', '', open(h).read(), flags=re.S)
        names |= set(re.findall(r'\b(p3d_[a-z0-9_]+)\s*\(', src))
    return sorted(names)


def test_library_loads_and_exports_every_declared_symbol():
    from pix2pix3d_amd import _lib
    handle = _lib.lib()                      # raises if the .so is missing or unresolved
    declared = _declared_symbols()
    assert len(declared) >= 6
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/ but not exported'
    assert handle.p3d_abi_version() >= 1
    assert isinstance(_lib.launch_count(), int)


def test_python_binding_covers_every_declared_symbol():
    from pix2pix3d_amd import _lib
    _lib.lib()
    import importlib
    for m in ('torch_utils.ops.bias_act', 'torch_utils.ops.upfirdn2d', 'torch_utils.ops.modconv', 'torch_utils.ops.filtered_lrelu', 'torch_utils.ops.conv2d_gradfix', 'torch_utils.ops.bcast', 'torch_utils.ops.conv_layer',
              'training.volumetric_rendering.renderer', 'training.volumetric_rendering.ray_sampler', 'diagnostics'):
        try:
            importlib.import_module('pix2pix3d_amd.' + m)          # op modules register their entry points on import
        except ImportError:
            pass
    missing = [n for n in _declared_symbols() if n not in _lib._SIGNATURES]
    assert not missing, f'no ctypes signature for {missing}'


def test_argument_errors_are_reported_not_thrown():
    from pix2pix3d_amd import _lib
    h = _lib.lib()
    # null x: must come back as an error code + message, with no GPU work attempted
    code = h.p3d_bias_act(None, None, None, None, None, None, 0, 0, 1, 0.0, 1.0, -1.0, 16, 0, 1, None)
    assert code == -2
    assert b'non-null' in h.p3d_last_error()


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/p3d_hip.h is the boundary a maintainer binds from C / cgo / JNI: it must compile as C99 (and C++) on its own, and a C
    translation unit that only includes it must link against libp3d_hip.so and be able to call the entry points that need no GPU."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    hdr = os.path.join(ROOT, 'include', 'p3d_hip.h')
    for lang, std in (('c', 'c99'), ('c++', 'c++11')):
        r = subprocess.run([gcc, f'-std={std}', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-x', lang, hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    src = tmp_path / 'client.c'
    src.write_text('#include "p3d_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %d\\n", p3d_abi_version(), p3d_render_decoder_floats()); return p3d_abi_version() > 0 ? 0 : 1; }\n')
    exe = tmp_path / 'client'
    libdir = os.path.join(ROOT, 'pix2pix3d_amd')
    r = subprocess.run([gcc, '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe), '-L', libdir, '-lp3d_hip',
                        f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH=libdir + ':/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', '')))
    from pix2pix3d_amd import _lib
    assert r.returncode == 0 and int(r.stdout.split()[0]) == _lib.lib().p3d_abi_version() >= 3, (r.stdout, r.stderr)
