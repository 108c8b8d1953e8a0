"""Python view of the native C++ tensor library (``lib/libsinga_amd_cpp.so``)
through its flat C interface (csrc/cpp/include/singa/api/c_api.h): devices and
float32 / int32 tensors with element-wise functions, products, sums and
softmax.

The training stack of this package runs on torch-backed tensors and HIP
kernels; this module is the way to reach the devices that only the C++ library
has, which today means the OpenCL device (``device.create_opencl_device()``,
role of the reference's python/singa/device.py:44-73), and it doubles as an
executable specification of the interface the Java binding uses.
"""
import ctypes
import os

import numpy as np

_LIB = None
_LANGS = {0: "host", 1: "hip", 2: "opencl"}


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                        "libsinga_amd_cpp.so")
    if not os.path.exists(path):
        from . import build
        build.build_cpp(verbose=False)
    lib = ctypes.CDLL(path)
    vp, i64, i32, f32 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                         ctypes.c_float)
    for name, (res, args) in {
        "singa_last_error": (ctypes.c_char_p, []),
        "singa_num_gpus": (i32, []),
        "singa_num_opencl_devices": (i32, []),
        "singa_device_host": (vp, []),
        "singa_device_gpu": (vp, [i32]),
        "singa_device_opencl": (vp, [i32]),
        "singa_device_lang": (i32, [vp]),
        "singa_device_description": (i32, [vp, ctypes.c_char_p, i32]),
        "singa_device_sync": (None, [vp]),
        "singa_device_free": (None, [vp]),
        "singa_tensor_new": (vp, [ctypes.POINTER(i64), i32, i32, vp]),
        "singa_tensor_free": (None, [vp]),
        "singa_tensor_clone": (vp, [vp]),
        "singa_tensor_ndim": (i32, [vp]),
        "singa_tensor_dim": (i64, [vp, i32]),
        "singa_tensor_size": (i64, [vp]),
        "singa_tensor_dtype": (i32, [vp]),
        "singa_tensor_set_value": (i32, [vp, f32]),
        "singa_tensor_copy_from_float": (i32, [vp, ctypes.POINTER(f32), i64]),
        "singa_tensor_copy_from_int": (i32, [vp, ctypes.POINTER(i32), i64]),
        "singa_tensor_get_float": (i32, [vp, ctypes.POINTER(f32), i64]),
        "singa_tensor_get_int": (i32, [vp, ctypes.POINTER(i32), i64]),
        "singa_tensor_to_device": (i32, [vp, vp]),
        "singa_tensor_reshape": (vp, [vp, ctypes.POINTER(i64), i32]),
        "singa_tensor_transpose": (vp, [vp]),
        "singa_unary": (vp, [ctypes.c_char_p, vp]),
        "singa_binary": (vp, [ctypes.c_char_p, vp, vp]),
        "singa_matmul": (vp, [vp, vp]),
        "singa_sum": (vp, [vp, i32]),
        "singa_softmax": (vp, [vp]),
        "singa_axpy": (i32, [f32, vp, vp]),
    }.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


def _checked(value):
    """Handles are non-null, status codes non-negative; else raise with the
    library's message."""
    if value is None or (isinstance(value, int) and value < 0):
        raise RuntimeError(_lib().singa_last_error().decode() or
                           "native call failed")
    return value


def num_gpus():
    return _lib().singa_num_gpus()


def num_opencl_devices():
    return _lib().singa_num_opencl_devices()


class Device(object):
    """A device of the C++ library: ``Device.host()``, ``Device.gpu(i)``,
    ``Device.opencl(i)``."""

    def __init__(self, handle, owned=True):
        self._h, self._owned = _checked(handle), owned

    @classmethod
    def host(cls):
        return cls(_lib().singa_device_host())

    @classmethod
    def gpu(cls, index=0):
        return cls(_lib().singa_device_gpu(index))

    @classmethod
    def opencl(cls, index=0):
        return cls(_lib().singa_device_opencl(index))

    def lang(self):
        return _LANGS[_checked(_lib().singa_device_lang(self._h))]

    def description(self):
        buf = ctypes.create_string_buffer(512)
        _checked(_lib().singa_device_description(self._h, buf, 512))
        return buf.value.decode()

    def sync(self):
        _lib().singa_device_sync(self._h)

    Sync = sync

    def tensor(self, array):
        return Tensor.from_numpy(array, self)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None) and \
                _LIB is not None:
            _LIB.singa_device_free(self._h)
            self._h = None


class Tensor(object):
    """float32 / int32 tensor on a native device."""

    def __init__(self, handle, device):
        self._h, self.device = _checked(handle), device

    @classmethod
    def from_numpy(cls, array, device):
        a = np.ascontiguousarray(array)
        is_int = np.issubdtype(a.dtype, np.integer)
        a = a.astype(np.int32 if is_int else np.float32)
        shape = (ctypes.c_int64 * a.ndim)(*a.shape)
        t = cls(_lib().singa_tensor_new(shape, a.ndim, 2 if is_int else 0,
                                        device._h), device)
        if is_int:
            _checked(_lib().singa_tensor_copy_from_int(
                t._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), a.size))
        else:
            _checked(_lib().singa_tensor_copy_from_float(
                t._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                a.size))
        return t

    @property
    def shape(self):
        lib = _lib()
        return tuple(int(lib.singa_tensor_dim(self._h, i))
                     for i in range(lib.singa_tensor_ndim(self._h)))

    def numpy(self):
        lib = _lib()
        n = int(lib.singa_tensor_size(self._h))
        if lib.singa_tensor_dtype(self._h) == 2:
            out = np.empty(n, np.int32)
            _checked(lib.singa_tensor_get_int(
                self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n))
        else:
            out = np.empty(n, np.float32)
            _checked(lib.singa_tensor_get_float(
                self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                n))
        return out.reshape(self.shape)

    def _new(self, handle):
        return Tensor(handle, self.device)

    def unary(self, op):
        return self._new(_lib().singa_unary(op.encode(), self._h))

    def _binary(self, op, other):
        return self._new(_lib().singa_binary(op.encode(), self._h, other._h))

    def __add__(self, other):
        return self._binary("add", other)

    def __sub__(self, other):
        return self._binary("sub", other)

    def __mul__(self, other):
        return self._binary("mul", other)

    def __truediv__(self, other):
        return self._binary("div", other)

    def __matmul__(self, other):
        return self._new(_lib().singa_matmul(self._h, other._h))

    @property
    def T(self):
        return self._new(_lib().singa_tensor_transpose(self._h))

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(
            shape[0], int) else shape
        arr = (ctypes.c_int64 * len(shape))(*shape)
        return self._new(_lib().singa_tensor_reshape(self._h, arr, len(shape)))

    def sum(self, axis):
        return self._new(_lib().singa_sum(self._h, axis))

    def softmax(self):
        return self._new(_lib().singa_softmax(self._h))

    def axpy(self, alpha, x):
        """self += alpha * x"""
        _checked(_lib().singa_axpy(alpha, x._h, self._h))
        return self

    def to(self, device):
        t = self._new(_lib().singa_tensor_clone(self._h))
        _checked(_lib().singa_tensor_to_device(t._h, device._h))
        t.device = device
        return t

    def __del__(self):
        # a failed call raises before the handle is stored
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.singa_tensor_free(self._h)
            self._h = None
