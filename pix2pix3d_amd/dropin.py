"""Make this package answer to the reference's import paths.

The reference (and every released checkpoint, whose pickles re-import module source by name — SURVEY §5 N3) uses
absolute imports such as ``from torch_utils.ops import bias_act`` and ``training.triplane_cond.TriPlane...``.
``install()`` registers this package's mirrors under those names in ``sys.modules`` so that
``training_loop.py`` / ``applications/*.py`` of the reference run unchanged on top of the HIP kernels.
Modules this package does not mirror (loss, dataset, camera_utils, metrics, ...) keep resolving to the
reference checkout if it is on ``sys.path``: for that, ``training`` / ``torch_utils`` get the reference's
directories appended to their ``__path__``.
"""
import importlib
import importlib.util
import os
import sys

_reference_root = None          # set by install(reference_root=...)
_reference_modules = {}

_MIRRORED = [
    'dnnlib', 'dnnlib.util', 'legacy',
    'torch_utils', 'torch_utils.misc', 'torch_utils.persistence', 'torch_utils.custom_ops',
    'torch_utils.ops', 'torch_utils.ops.bias_act', 'torch_utils.ops.upfirdn2d', 'torch_utils.ops.fma',
    'torch_utils.ops.conv2d_gradfix', 'torch_utils.ops.conv2d_resample', 'torch_utils.ops.grid_sample_gradfix',
    'torch_utils.ops.filtered_lrelu',
    'training', 'training.networks_stylegan2', 'training.superresolution', 'training.triplane', 'training.triplane_cond',
    'training.dual_discriminator',
    'training.volumetric_rendering', 'training.volumetric_rendering.renderer', 'training.volumetric_rendering.ray_marcher',
    'training.volumetric_rendering.ray_sampler', 'training.volumetric_rendering.math_utils',
]
# host-side consumers this package restates only so that the training phases can run WITHOUT a checkout (bench.py --train-step, the GPU tests):
# aliased when no checkout is registered; with one, the checkout's own file serves the name (the reference's loss.py runs unchanged on the mirrors)
_FALLBACK = ['training.loss']


def install(reference_root=None, strict=False):
    """Alias the mirrors; returns the list of names installed.  ``reference_root`` (optional) is appended to the
    package search paths so un-mirrored reference modules (e.g. ``training.loss``) remain importable."""
    done = []
    for name in _MIRRORED:
        try:
            mod = importlib.import_module('pix2pix3d_amd.' + name)
        except ImportError:
            if strict:
                raise
            continue
        sys.modules[name] = mod
        done.append(name)
    if reference_root:
        global _reference_root
        _reference_root = reference_root
        for pkg in ('training', 'torch_utils', 'dnnlib'):
            extra = os.path.join(reference_root, pkg)
            if pkg in sys.modules and os.path.isdir(extra) and extra not in sys.modules[pkg].__path__:
                sys.modules[pkg].__path__.insert(0, extra)        # first: a name that is NOT aliased above resolves to the checkout's file
        if reference_root not in sys.path:
            sys.path.append(reference_root)
    for name in _FALLBACK:
        served_by_checkout = reference_root and os.path.isfile(os.path.join(reference_root, *name.split('.')) + '.py')
        if served_by_checkout:
            if getattr(sys.modules.get(name), '__name__', '').startswith('pix2pix3d_amd.'):
                del sys.modules[name]                             # an earlier install() without a checkout aliased the restatement
        elif name not in sys.modules:
            try:
                sys.modules[name] = importlib.import_module('pix2pix3d_amd.' + name)
                done.append(name)
            except ImportError:
                if strict:
                    raise
    return done


def reference_attr(module, name):
    """``module.name`` served by the reference checkout's own file — for the host-side helpers the mirrors deliberately do not restate
    (``dnnlib.util.open_url`` / ``Logger`` / ``format_time``, ``misc.print_module_summary`` / ``ddp_sync``, ...): the mirrored modules
    forward unknown attributes here (PEP 562), so ``training_loop.py`` and ``applications/*.py`` find them where they expect them.  The
    file is executed under a private module name; its own ``import dnnlib`` / ``from torch_utils import misc`` resolve to the mirrors."""
    if name.startswith('__'):
        raise AttributeError(name)
    if _reference_root is None:
        raise AttributeError(f'{module}.{name} is a host-side helper pix2pix3d_amd does not mirror; call '
                             f'pix2pix3d_amd.dropin.install(reference_root=<pix2pix3D checkout>) and the reference\'s own {module} will serve it')
    mod = _reference_modules.get(module)
    if mod is None:
        base = os.path.join(_reference_root, *module.split('.'))
        path = base + '.py' if os.path.isfile(base + '.py') else os.path.join(base, '__init__.py')
        spec = importlib.util.spec_from_file_location('_p3d_reference.' + module, path)
        mod = importlib.util.module_from_spec(spec)
        _reference_modules[module] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            del _reference_modules[module]
            raise
    try:
        return getattr(mod, name)
    except AttributeError:
        raise AttributeError(f'neither pix2pix3d_amd nor the reference checkout defines {module}.{name}') from None
