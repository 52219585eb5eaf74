"""ctypes binding of libp3d_hip.so (C ABI declared in include/p3d_hip.h).

The library is mandatory for CUDA/HIP tensors: ``lib()`` raises if it cannot be loaded, and no op
in this package has a GPU fallback.  torch is imported first so the HIP runtime the library binds
to (SONAME libamdhip64.so.7) is the one torch already mapped — streams and device pointers are
then shared between torch and the kernels.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P3D_LIB_PATH') or os.path.join(_PKG_DIR, 'libp3d_hip.so')      # (P3D_LIB_PATH: A/B runs of two builds on one box)

P3D_OK = 0
P3D_ERR_UNSUPPORTED = -1
DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.float64: 2}
FAMILY = {'bias_act': 0, 'upfirdn2d': 1, 'filtered_lrelu': 2, 'render': 3, 'conv': 4, 'aux': 5}

_c_void_p, _c_int, _c_i32, _c_i64, _c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_i32x4, _i64x4 = ctypes.c_int32 * 4, ctypes.c_int64 * 4
_i32x2, _i64x2 = ctypes.c_int32 * 2, ctypes.c_int64 * 2

_SIGNATURES = {
    'p3d_last_error': (ctypes.c_char_p, []),
    'p3d_abi_version': (_c_int, []),
    'p3d_launch_count': (ctypes.c_uint64, []),
    'p3d_launch_count_of': (ctypes.c_uint64, [_c_int]),
    'p3d_bias_act': (_c_int, [_c_void_p] * 6 + [_c_int, _c_int, _c_int, _c_float, _c_float, _c_float,
                              _c_i64, _c_i32, _c_i64, _c_void_p]),
    'p3d_upfirdn2d': (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int,
                               ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                               ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                               ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                               _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_float, _c_void_p]),
    'p3d_upfirdn2d_acc': (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int,
                                   ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                                   ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                                   ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64),
                                   _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_float, _c_void_p]),
}

_lib = None
_load_error = None
kernel_events = {}      # name -> list of (start, end) torch.cuda.Event pairs; filled only while a key exists (bench.py)


def _load():
    global _lib, _load_error
    if _lib is not None or _load_error is not None:
        return
    try:
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so is stale
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    except (OSError, AttributeError) as e:      # missing file, unresolved HIP runtime, stale build
        _load_error = e


def available():
    _load()
    return _lib is not None


_tls = threading.local()       # .device: ordinal of the tensor the pending call's stream was taken from (set by stream_of)


class _Guarded:
    """The library handle with a device guard around every entry point: a kernel must be enqueued with ITS tensors' device
    current (the reference plugins wrap each op in ``OptionalCUDAGuard(device_of(x))``, bias_act.cpp:57), not whatever device the
    caller last selected.  ``stream_of(t)`` — evaluated while the call's arguments are built — notes t's device; the call then
    switches to it for its duration when it is not the current one."""

    def __init__(self, handle):
        self._handle, self._fns = handle, {}

    def __getattr__(self, name):
        fn = self._fns.get(name)
        if fn is None:
            raw = getattr(self._handle, name)

            def fn(*args, _raw=raw):
                dev, _tls.device = getattr(_tls, 'device', None), None
                if dev is None or dev == torch.cuda.current_device():
                    return _raw(*args)
                with torch.cuda.device(dev):
                    return _raw(*args)
            self._fns[name] = fn
        return fn


_guarded = None


def lib():
    """Return the loaded library (device-guarded) or raise: there is no fallback for device tensors."""
    global _guarded
    _load()
    if _lib is None:
        raise RuntimeError(
            f'pix2pix3d_amd: the gfx950 kernel library {LIB_PATH} could not be loaded ({_load_error}). '
            'Build it with `python -m pix2pix3d_amd.build` (or __graft_entry__.build()); '
            'CUDA/HIP tensors have no fallback path in this package.')
    if _guarded is None:
        _guarded = _Guarded(_lib)
    return _guarded


def register(name, restype, argtypes):
    """Declare the signature of one more exported symbol (used by the op modules)."""
    _SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def check(code, what):
    if code != P3D_OK:
        msg = lib().p3d_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what}: libp3d_hip error {code}: {msg}')


def stream_of(t):
    """Current stream of t's device (what the reference plugins launch on) — and tell the device guard which device that is."""
    _tls.device = t.device.index
    return _c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    return None if t is None else _c_void_p(t.data_ptr())


def launch_count(family=None):
    if not available():
        return 0
    return int(_lib.p3d_launch_count() if family is None else _lib.p3d_launch_count_of(FAMILY[family]))


def i32x4(*v): return _i32x4(*v)
def i64x4(*v): return _i64x4(*v)
def i32x2(*v): return _i32x2(*v)
def i64x2(*v): return _i64x2(*v)


class kernel_timer:
    """``with kernel_timer('render_forward', tensor):`` brackets the enclosed launches with events on the tensor's
    current stream, but only while ``kernel_events['render_forward']`` exists — otherwise it costs one dict lookup."""

    def __init__(self, name, t):
        self.log = kernel_events.get(name)
        self.dev = t.device

    def __enter__(self):
        if self.log is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.log is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.dev))
            self.log.append((self.e0, e1))
