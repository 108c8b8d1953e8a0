"""The flat C interface of the native tensor library (csrc/cpp/include/singa/
api/c_api.h), driven through ctypes: the calls the Java binding (java/) makes,
including the reference's Java tensor test (java/src/test/java/org/apache/
singa/TestTensor.java: SetFloatValue, Square, CopyFloatDataFromHostPtr,
GetFloatValue) and the error path."""
import ctypes
import os

import numpy as np
import pytest

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "singa_amd", "lib", "libsinga_amd_cpp.so")


@pytest.fixture(scope="module")
def api():
    if not os.path.exists(LIB):
        pytest.skip("C++ library not built")
    lib = ctypes.CDLL(LIB)
    vp, i64, i32, f32 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                         ctypes.c_float)
    sig = {
        "singa_last_error": (ctypes.c_char_p, []),
        "singa_version": (ctypes.c_char_p, []),
        "singa_num_gpus": (i32, []),
        "singa_device_host": (vp, []),
        "singa_device_gpu": (vp, [i32]),
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
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def _new(api, shape, dev=None, dtype=0):
    arr = (ctypes.c_int64 * len(shape))(*shape)
    t = api.singa_tensor_new(arr, len(shape), dtype, dev)
    assert t, api.singa_last_error()
    return t


def _put(api, t, a):
    a = np.ascontiguousarray(a, np.float32)
    assert api.singa_tensor_copy_from_float(
        t, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size) == 0


def _get(api, t):
    n = api.singa_tensor_size(t)
    shape = [api.singa_tensor_dim(t, i)
             for i in range(api.singa_tensor_ndim(t))]
    out = np.empty(n, np.float32)
    assert api.singa_tensor_get_float(
        t, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n) == 0
    return out.reshape(shape)


def _devices(api):
    devs = [("host", None)]
    if api.singa_num_gpus() > 0:
        devs.append(("gpu", api.singa_device_gpu(0)))
    return devs


def test_reference_java_tensor_test(api):
    assert api.singa_version() == b"4.2.0"
    for _, dev in _devices(api):
        t1 = _new(api, [2, 3], dev)
        assert api.singa_tensor_set_value(t1, 0.1) == 0
        t2 = api.singa_unary(b"square", t1)
        np.testing.assert_allclose(_get(api, t2), 0.01, atol=1e-4)
        data = np.arange(6, dtype=np.float32)
        t3 = _new(api, [2, 3], dev)
        _put(api, t3, data)
        np.testing.assert_allclose(_get(api, t3).reshape(-1), data)
        for t in (t1, t2, t3):
            api.singa_tensor_free(t)


def test_arithmetic_shapes_and_errors(api):
    rs = np.random.RandomState(0)
    a, b = rs.randn(4, 5).astype(np.float32), rs.randn(4, 5).astype(np.float32)
    for _, dev in _devices(api):
        ta, tb = _new(api, [4, 5], dev), _new(api, [4, 5], dev)
        _put(api, ta, a)
        _put(api, tb, b)
        for op, fn in ((b"add", np.add), (b"sub", np.subtract),
                       (b"mul", np.multiply), (b"div", np.divide)):
            r = api.singa_binary(op, ta, tb)
            np.testing.assert_allclose(_get(api, r), fn(a, b), rtol=1e-5)
            api.singa_tensor_free(r)
        for op, fn in ((b"relu", lambda v: np.maximum(v, 0)), (b"exp", np.exp),
                       (b"tanh", np.tanh), (b"abs", np.abs)):
            r = api.singa_unary(op, ta)
            np.testing.assert_allclose(_get(api, r), fn(a), rtol=1e-5,
                                       atol=1e-6)
            api.singa_tensor_free(r)
        tt = api.singa_tensor_transpose(tb)
        assert [api.singa_tensor_dim(tt, i) for i in range(2)] == [5, 4]
        p = api.singa_matmul(ta, tt)
        np.testing.assert_allclose(_get(api, p), a @ b.T, rtol=1e-4, atol=1e-5)
        s0 = api.singa_sum(ta, 0)
        np.testing.assert_allclose(_get(api, s0), a.sum(0), rtol=1e-5,
                                   atol=1e-5)
        sm = api.singa_softmax(ta)
        e = np.exp(a - a.max(1, keepdims=True))
        np.testing.assert_allclose(_get(api, sm), e / e.sum(1, keepdims=True),
                                   rtol=1e-5)
        assert api.singa_axpy(2.0, ta, tb) == 0
        np.testing.assert_allclose(_get(api, tb), b + 2 * a, rtol=1e-5)
        shp = (ctypes.c_int64 * 3)(2, 2, 5)
        r3 = api.singa_tensor_reshape(ta, shp, 3)
        assert api.singa_tensor_ndim(r3) == 3 and \
            api.singa_tensor_size(r3) == 20
        c = api.singa_tensor_clone(ta)
        np.testing.assert_allclose(_get(api, c), a)
        # failures come back as NULL / negative with a message, never as a
        # C++ exception
        assert not api.singa_unary(b"cube", ta)
        assert b"cube" in api.singa_last_error()
        bad = _new(api, [3, 3], dev)
        assert not api.singa_matmul(ta, bad)
        assert api.singa_last_error()
        big = np.zeros(64, np.float32)
        assert api.singa_tensor_copy_from_float(
            bad, big.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 64) < 0
        assert api.singa_tensor_dim(ta, 7) == -1
        for t in (ta, tb, tt, p, s0, sm, r3, c, bad):
            api.singa_tensor_free(t)


def test_int_tensors_and_device_moves(api):
    ids = np.array([3, 1, 4, 1, 5], np.int32)
    t = _new(api, [5], None, 2)
    assert api.singa_tensor_dtype(t) == 2
    assert api.singa_tensor_copy_from_int(
        t, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 5) == 0
    out = np.zeros(5, np.int32)
    assert api.singa_tensor_get_int(
        t, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 5) == 0
    np.testing.assert_array_equal(out, ids)
    host = api.singa_device_host()
    assert host
    api.singa_device_sync(host)
    if api.singa_num_gpus() > 0:
        g = api.singa_device_gpu(0)
        f = _new(api, [2, 2])
        _put(api, f, np.eye(2))
        assert api.singa_tensor_to_device(f, g) == 0
        sq = api.singa_unary(b"exp", f)
        np.testing.assert_allclose(_get(api, sq), np.exp(np.eye(2)), rtol=1e-5)
        api.singa_tensor_free(sq)
        api.singa_tensor_free(f)
        api.singa_device_free(g)
    api.singa_device_free(host)
    api.singa_tensor_free(t)


@pytest.mark.gpu
def test_c_interface_on_the_gpu(api):
    """The same calls with the tensors on the MI355X (the loops above include
    the GPU whenever one is present)."""
    assert api.singa_num_gpus() > 0
    test_reference_java_tensor_test(api)
    test_arithmetic_shapes_and_errors(api)
    test_int_tensors_and_device_moves(api)
