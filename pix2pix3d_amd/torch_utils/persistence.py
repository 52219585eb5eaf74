"""Checkpoint wire format of ``@persistent_class`` objects (reference: torch_utils/persistence.py:37-229).

On the wire a persistent object is ``REDUCE(torch_utils.persistence._reconstruct_persistent_obj, (meta,))`` with
``meta = dict(type='class', version=6, module_src=<source text of the defining module>, class_name=<str>, state=<__dict__>)``
(persistence.py:120-130).  The reference rebuilds the class by ``exec``-ing ``module_src`` (persistence.py:207-218), which is how a
released ``pix2pix3d_*.pkl`` carries its own (CUDA-plugin) model code.  This module speaks the same format but never executes
pickled source: ``class_name`` is resolved to this package's mirror of the class (the module is identified by matching the
``class`` statements of ``module_src`` against each mirrored module), the pickled ``state`` is installed with
``Module.__setstate__`` exactly as the reference does (persistence.py:199-203), and classes this package does not mirror
(``AugmentPipe``, the StyleGAN3 networks) come back as inert ``torch.nn.Module`` holders whose parameters, buffers and children
are intact.  ``legacy.load_network_pkl`` then re-instantiates the top-level networks from ``init_args`` / ``init_kwargs`` so that
the returned modules are this package's, running on the HIP kernels.

Writing: ``pickle.dump`` of a decorated object emits the same record (plus ``writer='pix2pix3d_amd'``), with ``module_src`` the
source of the mirror module, so snapshots written by ``training_loop.py`` over this package load back through the same reader.
"""
import copy
import functools
import importlib
import inspect
import re
import sys

import torch

_version = 6                    # persistence.py:28 — asserted on load (persistence.py:191)
_import_hooks = []              # persistence.py:162-185
_classes = {}                   # (module name, class name) -> decorated class
_module_src = {}                # module name -> source text

# modules whose classes may be named by a pickle; imported on first use so that every @persistent_class is registered
_MIRROR_MODULES = ['training.networks_stylegan2', 'training.superresolution', 'training.triplane', 'training.triplane_cond',
                   'training.dual_discriminator']


def _src_of(module_name):
    src = _module_src.get(module_name)
    if src is None:
        try:
            src = inspect.getsource(sys.modules[module_name])
        except (OSError, TypeError, KeyError):
            src = ''
        _module_src[module_name] = src
    return src


def persistent_class(orig_class):
    """Class decorator (persistence.py:37-134): records the constructor arguments as ``init_args`` / ``init_kwargs`` and makes
    instances pickle as a self-describing record.  The class is decorated in place (no wrapper subclass), so ``type(obj)`` is the
    class a reader of the source sees."""
    orig_init = orig_class.__init__

    @functools.wraps(orig_init)
    def __init__(self, *args, **kwargs):
        if not hasattr(self, '_init_args'):            # outermost constructor call wins
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)
        orig_init(self, *args, **kwargs)

    def __reduce__(self):
        getstate = getattr(self, '__getstate__', None)
        state = getstate() if callable(getstate) else None
        if state is None:
            state = self.__dict__
        cls = type(self)
        meta = dict(type='class', version=_version, module_src=_src_of(cls.__module__), class_name=cls.__name__, state=state,
                    writer='pix2pix3d_amd', module_name=cls.__module__)
        return _reconstruct_persistent_obj, (meta,), None

    orig_class.__init__ = __init__
    orig_class.__reduce__ = __reduce__
    orig_class.init_args = property(lambda self: copy.deepcopy(self._init_args))
    orig_class.init_kwargs = property(lambda self: copy.deepcopy(self._init_kwargs))
    orig_class._p3d_persistent = True
    _classes[(orig_class.__module__, orig_class.__name__)] = orig_class
    return orig_class


def is_persistent(obj):
    """persistence.py:138-147."""
    return bool(getattr(obj, '_p3d_persistent', False))


def import_hook(hook):
    """persistence.py:151-185: ``hook(meta) -> meta`` runs on every record before its class is resolved."""
    assert callable(hook)
    _import_hooks.append(hook)
    return hook


# ---------------------------------------------------------------------------------------------------------------------
class PickledModule(torch.nn.Module):
    """Stand-in for a persistent class this package has no mirror of: parameters / buffers / children as pickled, no forward."""
    pickled_class_name = None

    def forward(self, *args, **kwargs):
        raise NotImplementedError(f'{self.pickled_class_name} was restored from a checkpoint as a parameter holder: '
                                  'pix2pix3d_amd has no implementation of this class')

    init_args = property(lambda self: copy.deepcopy(self.__dict__.get('_init_args', ())))
    init_kwargs = property(lambda self: copy.deepcopy(self.__dict__.get('_init_kwargs', {})))


_holders = {}


def _holder_class(class_name):
    cls = _holders.get(class_name)
    if cls is None:
        cls = _holders[class_name] = type(class_name, (PickledModule,), dict(pickled_class_name=class_name))
    return cls


def _load_mirrors():
    pkg = __name__.rsplit('.torch_utils.', 1)[0]
    for name in _MIRROR_MODULES:
        importlib.import_module(pkg + '.' + name)


_persistent_stmt = re.compile(r'^@persistence\.persistent_class\s*\nclass\s+(\w+)', re.M)


def resolve_class(class_name, module_src='', module_name=None):
    """This package's class for a pickled (class_name, module_src), or None.  Several reference modules define the same class name
    (``TriPlaneGenerator`` in triplane.py and triplane_cond.py; ``SynthesisLayer`` / ``Generator`` in the StyleGAN2 and StyleGAN3
    files), so the module is identified first: the mirror whose set of persistent classes is closest (Jaccard) to the set of
    ``@persistence.persistent_class`` classes in the pickled source wins, and below 0.4 nothing matches — that source is some
    other network family (StyleGAN3 against the StyleGAN2 mirror scores 5/14) and its objects come back as holders."""
    _load_mirrors()
    if module_name is not None and (module_name, class_name) in _classes:
        return _classes[(module_name, class_name)]
    cands = [(m, c) for (m, n), c in _classes.items() if n == class_name]
    if not cands:
        return None
    wanted = set(_persistent_stmt.findall(module_src or ''))
    if not wanted:
        return cands[0][1] if len(cands) == 1 else None
    best, best_score = None, 0.0
    for m, c in cands:
        mine = {n for (mm, n) in _classes if mm == m}
        score = len(wanted & mine) / len(wanted | mine)
        if score > best_score:
            best, best_score = c, score
    return best if best_score >= 0.4 else None


def _reconstruct_persistent_obj(meta):
    """Reader of one record (persistence.py:189-203), without the ``exec`` of ``module_src``."""
    from .. import dnnlib
    meta = dnnlib.EasyDict(meta)
    meta.state = dnnlib.EasyDict(meta.state)
    for hook in _import_hooks:
        meta = hook(meta)
        assert meta is not None
    assert meta.version == _version, f'persistence version {meta.version} (this reader: {_version})'
    assert meta.type == 'class'
    cls = resolve_class(meta.class_name, meta.get('module_src', ''), meta.get('module_name'))
    foreign = cls is None
    if foreign:
        cls = _holder_class(meta.class_name)
    obj = cls.__new__(cls)
    state = dict(meta.state)
    setstate = getattr(obj, '__setstate__', None)
    if callable(setstate):
        setstate(state)
    else:
        obj.__dict__.update(state)
    # records not written by this package lack the attributes this package's constructors add: mark for re-instantiation
    obj.__dict__['_p3d_needs_rebuild'] = (not foreign) and meta.get('writer') != 'pix2pix3d_amd'
    return obj


_PLAIN = (bool, int, float, str, type(None), dict, list, tuple)


def rebuild(obj):
    """A persistent module restored from a reference-written record -> a freshly constructed instance of this package's class
    (``type(obj)(*init_args, **init_kwargs)``) carrying the record's parameters, buffers, train/eval flags, ``requires_grad``
    and plain top-level attributes (``rendering_kwargs``, ``neural_rendering_resolution`` — training_loop.py / loss.py assign
    those after construction).  Other objects are returned unchanged."""
    from . import misc
    if not (isinstance(obj, torch.nn.Module) and obj.__dict__.get('_p3d_needs_rebuild')):
        return obj
    new = type(obj)(*obj.init_args, **obj.init_kwargs)
    misc.copy_params_and_buffers(obj, new, require_all=True)
    old_mods = dict(obj.named_modules())
    for name, m in new.named_modules():
        if name in old_mods:
            m.training = old_mods[name].training
    old_params = dict(obj.named_parameters())
    for name, p in new.named_parameters():
        if name in old_params:
            p.requires_grad_(old_params[name].requires_grad)
    for k, v in obj.__dict__.items():
        if not k.startswith('_') and k != 'training' and isinstance(v, _PLAIN) and k in new.__dict__:
            new.__dict__[k] = copy.deepcopy(v)
    return new
