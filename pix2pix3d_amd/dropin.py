"""Make this package answer to the reference's import paths.

The reference (and every released checkpoint, whose pickles re-import module source by name — SURVEY §5 N3) uses
absolute imports such as ``from torch_utils.ops import bias_act`` and ``training.triplane_cond.TriPlane...``.
``install()`` registers this package's mirrors under those names in ``sys.modules`` so that
``training_loop.py`` / ``applications/*.py`` of the reference run unchanged on top of the HIP kernels.

Leaf modules are aliased as the SAME objects (``sys.modules['training.triplane_cond'] is pix2pix3d_amd.training.triplane_cond``:
one set of classes).  The PACKAGES (``training``, ``torch_utils``, ``torch_utils.ops``, ``dnnlib``, ``training.volumetric_rendering``) are
aliased by package objects of their own: their ``__path__`` holds only the reference checkout's directory (when one is registered), so a
name this package does not mirror (``training.loss_utils``, ``torch_utils.training_stats``, ``training.dataset`` ...) — or one it restates
only as a fallback (``training.loss``) — resolves to the checkout's file under the REFERENCE's name, and nothing the checkout serves is
ever registered under, or attached to, ``pix2pix3d_amd.*``: ``import pix2pix3d_amd.training.loss`` is this package's file in every state.
Attributes of the real package that are not submodules (``dnnlib.EasyDict``, ``dnnlib.make_cache_dir_path`` ...) are forwarded.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

_reference_root = None          # set by install(reference_root=...)
_reference_modules = {}

_PACKAGES = ['dnnlib', 'torch_utils', 'torch_utils.ops', 'training', 'training.volumetric_rendering']
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


class _AliasPackage(types.ModuleType):
    """``training`` / ``torch_utils`` / ... as the reference names them: submodules come through the import system (the aliases in
    ``sys.modules``, then the checkout's directory on ``__path__``); every other attribute is the real package's."""

    def __getattr__(self, attr):
        real = self.__dict__['_p3d_real']
        v = getattr(real, attr)                                   # AttributeError passes through (the real package may forward to reference_attr)
        if isinstance(v, types.ModuleType) and getattr(v, '__name__', '').startswith(real.__name__ + '.'):
            raise AttributeError(f'{self.__name__}.{attr}: submodules are resolved by the import system, not through the package attribute')
        return v


def _alias_package(name, real):
    m = sys.modules.get(name)
    if not isinstance(m, _AliasPackage):
        m = _AliasPackage(name, getattr(real, '__doc__', None))
        m.__dict__['_p3d_real'] = real
        m.__package__ = name
        m.__path__ = []
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        m.__spec__.submodule_search_locations = m.__path__
        if getattr(real, '__file__', None):
            m.__file__ = real.__file__
        sys.modules[name] = m
    return m


def _attach(name, mod):
    parent, _, leaf = name.rpartition('.')
    if parent and isinstance(sys.modules.get(parent), _AliasPackage):
        setattr(sys.modules[parent], leaf, mod)


def install(reference_root=None, strict=False):
    """Alias the mirrors; returns the list of names installed.  ``reference_root`` (optional): the reference checkout whose files serve
    every name this package does not mirror (``training.loss_utils``, ``camera_utils`` ...) and the fallback names (``training.loss``)."""
    global _reference_root
    done = []
    for name in _FALLBACK:                                        # registered under this package's own name first, whatever the aliases end up serving
        try:
            importlib.import_module('pix2pix3d_amd.' + name)
        except ImportError:
            if strict:
                raise
    for name in _MIRRORED:
        try:
            mod = importlib.import_module('pix2pix3d_amd.' + name)
        except ImportError:
            if strict:
                raise
            continue
        if name in _PACKAGES:
            mod = _alias_package(name, mod)
        else:
            sys.modules[name] = mod
        _attach(name, mod)
        done.append(name)
    if reference_root:
        _reference_root = reference_root
        for pkg in _PACKAGES:
            extra = os.path.join(reference_root, *pkg.split('.'))
            alias = sys.modules.get(pkg)
            if isinstance(alias, _AliasPackage) and os.path.isdir(extra) and extra not in alias.__path__:
                alias.__path__.append(extra)                      # (the mirrors never need it: they are in sys.modules under the reference's names already)
        if reference_root not in sys.path:
            sys.path.append(reference_root)
    for name in _FALLBACK:
        served_by_checkout = reference_root and os.path.isfile(os.path.join(reference_root, *name.split('.')) + '.py')
        parent, _, leaf = name.rpartition('.')
        if served_by_checkout:
            if getattr(sys.modules.get(name), '__name__', '').startswith('pix2pix3d_amd.'):
                del sys.modules[name]                             # an earlier install() without a checkout aliased the restatement
                if isinstance(sys.modules.get(parent), _AliasPackage):
                    sys.modules[parent].__dict__.pop(leaf, None)
        elif name not in sys.modules:
            try:
                sys.modules[name] = importlib.import_module('pix2pix3d_amd.' + name)
                _attach(name, sys.modules[name])
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
