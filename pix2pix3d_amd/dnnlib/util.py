"""Config plumbing used by the hot path (reference: dnnlib/util.py:42 EasyDict, :303 construct_class_by_name)."""
import importlib
from typing import Any


class EasyDict(dict):
    """dict whose items are also attributes (reference: dnnlib/util.py:42-55)."""
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)


def get_module_from_obj_name(obj_name: str):
    """Split 'pkg.mod.Obj.attr' into (imported module, 'Obj.attr'), trying the longest module prefix first."""
    # the reference's dotted names ('training.superresolution.X') resolve to this package's mirrors first, so
    # configs written for the reference work whether or not pix2pix3d_amd.dropin.install() has aliased sys.modules
    if obj_name.split('.')[0] in ('training', 'torch_utils', 'dnnlib') and not obj_name.startswith('pix2pix3d_amd.'):
        try:
            return get_module_from_obj_name('pix2pix3d_amd.' + obj_name)
        except ImportError:
            pass
    parts = obj_name.split('.')
    last_err = None
    for cut in range(len(parts) - 1, 0, -1):
        mod_name, local = '.'.join(parts[:cut]), '.'.join(parts[cut:])
        try:
            module = importlib.import_module(mod_name)
        except ModuleNotFoundError as e:
            if e.name is not None and not mod_name.startswith(e.name) and e.name != mod_name:
                raise                      # a real missing dependency inside the module, not a wrong split
            last_err = e
            continue
        obj = module
        try:
            for p in local.split('.'):
                obj = getattr(obj, p)
            return module, local
        except AttributeError as e:
            last_err = e
    raise ImportError(f'cannot resolve {obj_name!r}: {last_err}')


def get_obj_by_name(name: str) -> Any:
    module, local = get_module_from_obj_name(name)
    obj = module
    for p in local.split('.'):
        obj = getattr(obj, p)
    return obj


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn)
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    return call_func_by_name(*args, func_name=class_name, **kwargs)


def __getattr__(name):
    """Host-side helpers that are not restated here resolve to the reference checkout's own ``dnnlib.util`` (see dropin.reference_attr)."""
    from .. import dropin
    return dropin.reference_attr('dnnlib.util', name)
