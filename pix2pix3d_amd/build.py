"""Build libp3d_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

No torch headers, no pybind: every ``csrc/*.hip`` translation unit is compiled to an object
(in parallel, cached by mtime) and linked into ``pix2pix3d_amd/libp3d_hip.so`` so the binary
travels with the source tree to the GPU box.  hipcc cross-compiles without a GPU present.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB_PATH = os.path.join(PKG_DIR, 'libp3d_hip.so')
PROBES_DIR = os.path.join(CSRC, 'probes')
PROBES_LIB_PATH = os.path.join(PKG_DIR, 'libp3d_probes.so')      # hardware probes: a library of its own, never part of the product ABI
ARCH = 'gfx950'
CXXFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
            '-fno-gpu-rdc', '-DNDEBUG']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the gfx950 kernel library cannot be built')


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def probe_sources():
    return sorted(os.path.join(PROBES_DIR, f) for f in os.listdir(PROBES_DIR) if f.endswith('.hip'))


def headers():
    inc = os.path.join(os.path.dirname(PKG_DIR), 'include')
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs += [os.path.join(PROBES_DIR, f) for f in os.listdir(PROBES_DIR) if f.endswith('.h')]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')]
    return sorted(hs)


def _compile(src, verbose, extra=()):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o')
    if _newer(obj, [src] + headers()):
        return obj, False
    cmd = [_hipcc()] + CXXFLAGS + list(extra) + ['-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}')
    if verbose and r.stdout.strip():
        print(r.stdout)
    return obj, True


def build_library(force=False, verbose=False):
    """Compile + link; returns the path of the shared library."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, verbose), srcs))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not _newer(LIB_PATH, objs):
        cmd = [_hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}')
    build_probes(verbose=verbose)
    return LIB_PATH


def build_probes(verbose=False):
    """libp3d_probes.so (csrc/probes/*.hip, -DP3D_BUILD_PROBES): the hardware probes the tests and the kernel studies use, with their own
    copy of the library services (p3d_common.hip) so that nothing of them is exported through libp3d_hip.so."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = probe_sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, len(srcs)))) as ex:
        results = list(ex.map(lambda s: _compile(s, verbose, ('-DP3D_BUILD_PROBES',)), srcs))
    objs = [o for o, _ in results] + [os.path.join(OBJ_DIR, 'p3d_common.o')]
    if any(changed for _, changed in results) or not _newer(PROBES_LIB_PATH, objs):
        cmd = [_hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', PROBES_LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}')
    return PROBES_LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
