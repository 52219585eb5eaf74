"""Minimal stand-in for the reference's ``dnnlib`` (dnnlib/util.py:42, 58, 303): only the pieces the
generator/renderer path touches — attribute dictionaries and construction of objects by dotted name."""
from .util import EasyDict, construct_class_by_name, get_obj_by_name, call_func_by_name, get_module_from_obj_name  # noqa: F401
from . import util  # noqa: F401


def __getattr__(name):          # dnnlib.make_cache_dir_path and friends: the reference's own dnnlib/util.py (see dropin.reference_attr)
    return getattr(util, name)
