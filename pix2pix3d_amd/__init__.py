"""pix2pix3d_amd — MI355X-native (gfx950) generator/renderer hot path of pix2pix3D.

Layout: ``csrc/`` HIP kernels + C ABI (``include/p3d_hip.h``), and a host-side mirror of the
reference's operator / renderer / generator interface under ``torch_utils`` and ``training``
(same module names, signatures and error behaviour as the reference so its ``training_loop.py``
and ``applications/`` run unchanged once ``install_dropin()`` has aliased the import paths).

Derived-weight caches: the layers keep re-laid / split / modulated copies of their weights keyed on the parameter's autograd version counter
(``torch_utils/ops/modconv.py``, ``conv_layer.py``).  Every in-place tensor op — optimizer steps, ``copy_``, ``load_state_dict`` — bumps that
counter; a write THROUGH ``.data`` (or any other path that bypasses it: a custom EMA on ``p.data``, ``dist.broadcast(p.data)``) does not, and
must be followed by ``pix2pix3d_amd.invalidate_weight_caches()`` (``dp.broadcast_module`` and ``legacy`` loading already do).
"""
from . import _lib  # noqa: F401

__all__ = ['install_dropin', 'kernel_library_available', 'invalidate_weight_caches']


def kernel_library_available():
    return _lib.available()


def install_dropin():
    """Alias this package's mirrors as the reference's top-level import paths."""
    from .dropin import install
    return install()


def invalidate_weight_caches():
    """Drop every cached derived weight (see the module docstring): call after writing parameters through ``.data``."""
    from .torch_utils.ops import modconv
    modconv.invalidate_caches()
