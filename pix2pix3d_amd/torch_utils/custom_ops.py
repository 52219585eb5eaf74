"""Plugin seam of the reference (torch_utils/custom_ops.py:61-157 ``get_plugin``): there it JIT-builds a
pybind module from .cpp/.cu sources; here the "plugins" are thin objects over the prebuilt C-ABI
library, so nothing is ever compiled (or hipified) at import or call time."""
from .. import _lib
from . import ops  # noqa: F401

verbosity = 'brief'      # written by train.py:54; kept for API parity
_cached_plugins = dict()


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):
    """Return an object exposing the reference plugin's functions (bias_act.cpp:98, upfirdn2d.cpp:106,
    filtered_lrelu.cpp:298) implemented over libp3d_hip.so.  ``sources``/``headers``/``build_kwargs`` are ignored."""
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    _lib.lib()                                        # fail loudly when the kernel library is absent
    if module_name == 'bias_act_plugin':
        from .ops import bias_act as m
        plugin = m._Plugin()
    elif module_name == 'upfirdn2d_plugin':
        from .ops import upfirdn2d as m
        plugin = m._Plugin()
    elif module_name == 'filtered_lrelu_plugin':
        from .ops import filtered_lrelu as m
        plugin = m._Plugin()
    else:
        raise RuntimeError(f'unknown plugin {module_name!r}')
    _cached_plugins[module_name] = plugin
    return plugin
