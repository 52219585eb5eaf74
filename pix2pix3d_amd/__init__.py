"""pix2pix3d_amd — MI355X-native (gfx950) generator/renderer hot path of pix2pix3D.

Layout: ``csrc/`` HIP kernels + C ABI (``include/p3d_hip.h``), and a host-side mirror of the
reference's operator / renderer / generator interface under ``torch_utils`` and ``training``
(same module names, signatures and error behaviour as the reference so its ``training_loop.py``
and ``applications/`` run unchanged once ``install_dropin()`` has aliased the import paths).
"""
from . import _lib  # noqa: F401

__all__ = ['install_dropin', 'kernel_library_available']


def kernel_library_available():
    return _lib.available()


def install_dropin():
    """Alias this package's mirrors as the reference's top-level import paths."""
    from .dropin import install
    return install()
