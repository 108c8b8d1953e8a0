"""Python tensor API tests, run on the host and (``-m gpu``) on the MI355X.

Cases follow the reference's test/python/test_tensor.py; expected values come
from numpy.
"""
import math
import random

import numpy as np
import pytest

from singa_amd import autograd, tensor


def T(a, dev):
    return tensor.Tensor(device=dev, data=np.asarray(a))


def test_tensor_fields(dev):
    t = tensor.Tensor((2, 3), dev)
    assert t.shape == (2, 3)
    assert tensor.product(t.shape) == 6
    assert t.ndim() == 2
    assert t.size() == 6
    assert t.memsize() == 6 * tensor.sizeof(tensor.float32)
    assert not t.is_transpose()


def test_unary_operators(dev):
    t = tensor.Tensor((2, 3), dev)
    t.set_value(0)
    assert tensor.to_numpy(t)[0, 0] == 0.0
    t += 1.23
    assert tensor.to_numpy(t)[0, 0] == pytest.approx(1.23)
    t -= 0.23
    assert tensor.to_numpy(t)[0, 0] == pytest.approx(1.0)
    t *= 2.5
    assert tensor.to_numpy(t)[0, 0] == pytest.approx(2.5)
    t /= 2
    assert tensor.to_numpy(t)[0, 0] == pytest.approx(1.25)


def test_binary_operators(dev):
    t = tensor.Tensor((2, 3), dev)
    s = tensor.Tensor((2, 3), dev)
    t.set_value(3.2)
    s.set_value(2.1)
    assert tensor.to_numpy(t + s)[0, 0] == pytest.approx(5.3, abs=1e-5)
    assert tensor.to_numpy(t - s)[0, 0] == pytest.approx(1.1, abs=1e-5)
    assert tensor.to_numpy(t * s)[0, 0] == pytest.approx(6.72, abs=1e-5)
    assert tensor.to_numpy(t / s)[0, 0] == pytest.approx(3.2 / 2.1, abs=1e-5)


def test_comparison_operators(dev):
    t = tensor.Tensor((2, 3), dev)
    t.set_value(3.45)
    assert tensor.to_numpy(t < 3.45)[0, 0] == 0
    assert tensor.to_numpy(t <= 3.45)[0, 0] == 1
    assert tensor.to_numpy(t > 3.45)[0, 0] == 0
    assert tensor.to_numpy(t >= 3.45)[0, 0] == 1
    assert tensor.to_numpy(t == 3.45)[0, 0] == 1
    assert tensor.to_numpy(tensor.lt(t, 3.45))[0, 0] == 0
    assert tensor.to_numpy(tensor.le(t, 3.45))[0, 0] == 1
    assert tensor.to_numpy(tensor.gt(t, 3.45))[0, 0] == 0
    assert tensor.to_numpy(tensor.ge(t, 3.45))[0, 0] == 1
    assert tensor.to_numpy(tensor.eq(t, 3.45))[0, 0] == 1
    a = T(np.array([1.0, 2.0, 3.0], np.float32), dev)
    b = T(np.array([2.0, 2.0, 2.0], np.float32), dev)
    np.testing.assert_array_equal(tensor.to_numpy(a < b), [1, 0, 0])
    np.testing.assert_array_equal(tensor.to_numpy(a >= b), [0, 1, 1])


def test_tensor_copy(dev):
    t = tensor.Tensor((2, 3), dev)
    t.set_value(0)
    t += 1.23
    tc = t.copy()
    tdc = t.deepcopy()
    assert tensor.to_numpy(tc)[0, 0] == pytest.approx(1.23)
    assert tensor.to_numpy(tdc)[0, 0] == pytest.approx(1.23)
    t += 1.23
    assert tensor.to_numpy(t)[0, 0] == pytest.approx(2.46)
    assert tensor.to_numpy(tc)[0, 0] == pytest.approx(2.46)
    assert tensor.to_numpy(tdc)[0, 0] == pytest.approx(1.23)


def test_copy_data(dev):
    t = tensor.Tensor((2, 3), dev)
    s = tensor.Tensor((2, 3), dev)
    t.set_value(1.23)
    s.set_value(5.43)
    tensor.copy_data_to_from(t, s, 2)
    out = tensor.to_numpy(t)
    assert out[0, 0] == pytest.approx(5.43)
    assert out[0, 1] == pytest.approx(5.43)
    assert out[0, 2] == pytest.approx(1.23)


def test_global_methods(dev):
    x = np.array([[0.5, 1.5, 12.34], [2.0, 0.1, 3.0]], np.float32)
    t = T(x, dev)
    np.testing.assert_allclose(tensor.to_numpy(tensor.log(t)), np.log(x),
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.exp(t)), np.exp(x),
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sqrt(t)), np.sqrt(x),
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.square(t)), x * x,
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.tanh(t)), np.tanh(x),
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sigmoid(t)),
                               1 / (1 + np.exp(-x)), rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.abs(T(x - 1, dev))),
                               np.abs(x - 1), rtol=1e-6)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sign(T(x - 1, dev))),
                               np.sign(x - 1))
    np.testing.assert_allclose(tensor.to_numpy(tensor.relu(T(x - 1, dev))),
                               np.maximum(x - 1, 0))
    np.testing.assert_allclose(tensor.to_numpy(tensor.pow(t, 2.0)), x ** 2,
                               rtol=1e-5)
    e = np.exp(x - x.max(1, keepdims=True))
    np.testing.assert_allclose(tensor.to_numpy(tensor.softmax(t)),
                               e / e.sum(1, keepdims=True), rtol=1e-5)


def test_random(dev):
    x = tensor.Tensor((4000,), dev)
    x.gaussian(1, 0.01)
    assert tensor.average(x) == pytest.approx(1, abs=1e-3)
    x.uniform(2, 4)
    v = tensor.to_numpy(x)
    assert v.min() >= 2 and v.max() <= 4
    assert v.mean() == pytest.approx(3, abs=0.1)
    x.bernoulli(0.3)
    v = tensor.to_numpy(x)
    assert set(np.unique(v)) <= {0.0, 1.0}
    assert v.mean() == pytest.approx(0.3, abs=0.05)


def test_reflected_operators(dev):
    x = tensor.Tensor((3,), dev)
    x.set_value(1)
    assert tensor.average(1 + x) == 2.0
    assert tensor.average(1 - x) == 0.0
    assert tensor.average(2 * x) == 2.0
    assert tensor.average(2 / x) == 2.0
    assert tensor.average(-x) == -1.0


def test_inplace_api_returns_self(dev):
    x = tensor.Tensor((3,), dev)
    assert x.set_value(1) is x
    assert x.uniform(1, 2) is x
    assert x.bernoulli(1) is x
    assert x.gaussian(1, 2) is x


def test_numpy_convert(dev):
    a = np.asarray([[1, 0, 0], [0, 1, 0]], dtype=np.int32)
    assert np.sum(a - tensor.to_numpy(T(a, dev))) == 0
    a = a.astype(np.float32)
    assert np.sum(a - tensor.to_numpy(T(a, dev))) == 0.0
    h = np.asarray([[1.5, -2.25]], dtype=np.float16)
    th = T(h, dev)
    assert th.dtype == tensor.float16
    np.testing.assert_array_equal(tensor.to_numpy(th), h)


def test_transpose(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2) * 1.1
    ta = T(a, dev)
    np.testing.assert_array_almost_equal(
        tensor.to_numpy(tensor.transpose(ta)), np.transpose(a))
    np.testing.assert_array_almost_equal(
        tensor.to_numpy(tensor.transpose(ta, [0, 2, 1])),
        np.transpose(a, [0, 2, 1]))
    assert tensor.transpose(ta, [0, 2, 1]).is_transpose()


def test_6d_transpose(dev):
    s1, axes1, s2 = (2, 7, 6, 5, 4, 3), [5, 4, 3, 2, 1, 0], (2, 4, 3, 5, 7, 6)
    a = np.random.random(s1).astype(np.float32)
    ta = T(a, dev)
    ta = tensor.reshape(ta, s1)
    ta = tensor.transpose(ta, axes1)
    ta = tensor.reshape(ta, s2)
    np.testing.assert_array_almost_equal(
        tensor.to_numpy(ta), np.reshape(np.transpose(a, axes1), s2))


def test_einsum(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2) * 0.3 + 1
    ta = T(a, dev)
    np.testing.assert_allclose(
        tensor.to_numpy(tensor.einsum('kij,kij->kij', ta, ta)),
        np.einsum('kij,kij->kij', a, a), rtol=1e-5)
    np.testing.assert_allclose(
        tensor.to_numpy(tensor.einsum('kij,kih->kjh', ta, ta)),
        np.einsum('kij,kih->kjh', a, a), rtol=1e-5)


def test_repeat(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2)
    ta = T(a, dev)
    np.testing.assert_array_equal(
        tensor.to_numpy(tensor.repeat(ta, 2, axis=None)),
        np.repeat(a, 2, axis=None))
    np.testing.assert_array_equal(
        tensor.to_numpy(tensor.repeat(ta, 4, axis=1)), np.repeat(a, 4, axis=1))
    np.testing.assert_array_equal(
        tensor.to_numpy(ta.repeat([1, 2, 3], axis=1)),
        np.repeat(a, [1, 2, 3], axis=1))


def test_sum(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2) * 0.7
    ta = T(a, dev)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sum(ta)), np.sum(a),
                               rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sum(ta, axis=1)),
                               np.sum(a, axis=1), rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sum(ta, axis=2)),
                               np.sum(a, axis=2), rtol=1e-5)
    m = T(a.reshape(3, 4), dev)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sum_columns(m)),
                               a.reshape(3, 4).sum(1, keepdims=True), rtol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.sum_rows(m)),
                               a.reshape(3, 4).sum(0, keepdims=True), rtol=1e-5)
    assert tensor.average(ta) == pytest.approx(a.mean(), rel=1e-5)


def test_tensordot(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2) * 0.5 + 0.1
    ta = T(a, dev)
    np.testing.assert_allclose(
        tensor.to_numpy(tensor.tensordot(ta, ta, axes=1)),
        np.tensordot(a, a, axes=1), rtol=1e-5)
    np.testing.assert_allclose(
        tensor.to_numpy(tensor.tensordot(ta, ta, axes=([0, 1], [2, 1]))),
        np.tensordot(a, a, axes=([0, 1], [2, 1])), rtol=1e-5)


def test_reshape(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 2, 3)
    tb = tensor.reshape(T(a, dev), [2, 6])
    assert tb.shape == (2, 6)
    np.testing.assert_array_equal(tensor.to_numpy(tb), a.reshape(2, 6))


def test_transpose_then_reshape(dev):
    a = np.arange(12, dtype=np.float32).reshape(2, 3, 2)
    ta = T(a, dev).transpose((2, 0, 1)).reshape((2, 6))
    np.testing.assert_array_equal(tensor.to_numpy(ta),
                                  np.reshape(a.transpose((2, 0, 1)), (2, 6)))


def test_concatenate(dev):
    np1 = np.random.random([5, 6, 7, 8]).astype(np.float32)
    np2 = np.random.random([5, 6, 7, 1]).astype(np.float32)
    t3 = tensor.concatenate((T(np1, dev), T(np2, dev)), 3)
    np.testing.assert_array_equal(tensor.to_numpy(t3),
                                  np.concatenate((np1, np2), axis=3))
    t0 = tensor.concatenate((T(np1, dev), T(np1, dev)), 0)
    np.testing.assert_array_equal(tensor.to_numpy(t0),
                                  np.concatenate((np1, np1), axis=0))


def test_subscription(dev):
    np1 = np.random.random((5, 5, 5, 5)).astype(np.float32)
    t = T(np1, dev)
    np.testing.assert_array_equal(tensor.to_numpy(t[1:3, :, 1:, :-1]),
                                  np1[1:3, :, 1:, :-1])
    np.testing.assert_array_equal(tensor.to_numpy(t[2]), np1[2:3])


def test_ceil(dev):
    np1 = np.random.random([5, 6, 7, 8]).astype(np.float32) * 10
    np.testing.assert_array_equal(tensor.to_numpy(tensor.ceil(T(np1, dev))),
                                  np.ceil(np1))


def test_astype(dev):
    np_flt = np.random.random([2, 3]).astype(np.float32) * 10 - 5
    np_int = np_flt.astype(np.int32)
    t2 = T(np_flt, dev).as_type('int')
    np.testing.assert_array_equal(tensor.to_numpy(t2), np_int)
    t1 = t2.reshape([3, 2])
    np.testing.assert_array_equal(tensor.to_numpy(t1), np_int.reshape(3, 2))
    t1 = t1.as_type('float')
    np.testing.assert_array_equal(tensor.to_numpy(t1),
                                  np_int.astype(np.float32).reshape(3, 2))
    th = T(np_flt, dev).as_type(tensor.float16)
    np.testing.assert_allclose(tensor.to_numpy(th).astype(np.float32), np_flt,
                               rtol=2e-3)


@pytest.mark.parametrize("s1,s2", [
    ((2, 3, 4), (2, 4, 3)), ((2, 3, 4), (2, 4, 5)),
    ((2, 12, 64, 32), (2, 12, 32, 64)), ((1, 12, 7, 64), (1, 12, 64, 7)),
    ((1, 7, 96), (96, 96)),
])
def test_matmul_high_dim(dev, s1, s2):
    a = np.random.randn(*s1).astype(np.float32)
    b = np.random.randn(*s2).astype(np.float32)
    y = autograd.matmul(T(a, dev), T(b, dev))
    np.testing.assert_allclose(tensor.to_numpy(y), np.matmul(a, b), rtol=1e-4,
                               atol=1e-4)


def test_matmul_transpose(dev):
    X = np.random.random((1, 32, 12, 16)).astype(np.float32)
    W = np.random.random((1, 32, 12, 16)).astype(np.float32)
    x = autograd.transpose(T(X, dev), (0, 2, 1, 3))
    w = autograd.transpose(T(W, dev), (0, 2, 1, 3))
    w = autograd.transpose(w, (0, 1, 3, 2))
    y = autograd.matmul(x, w)
    Xt = np.transpose(X, (0, 2, 1, 3))
    Wt = np.transpose(np.transpose(W, (0, 2, 1, 3)), (0, 1, 3, 2))
    np.testing.assert_allclose(tensor.to_numpy(x), Xt)
    np.testing.assert_allclose(tensor.to_numpy(w), Wt)
    np.testing.assert_allclose(tensor.to_numpy(y), np.matmul(Xt, Wt),
                               rtol=1e-4)


def test_mult_gemm_gemv(dev):
    A = np.random.randn(5, 7).astype(np.float32)
    B = np.random.randn(7, 3).astype(np.float32)
    v = np.random.randn(7).astype(np.float32)
    np.testing.assert_allclose(tensor.to_numpy(tensor.mult(T(A, dev),
                                                           T(B, dev))),
                               A @ B, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.mult(T(A, dev),
                                                           T(v, dev))),
                               A @ v, rtol=1e-4, atol=1e-5)
    C = np.random.randn(5, 3).astype(np.float32)
    tcn = T(C, dev)
    tensor.mult(T(A, dev), T(B, dev), tcn, alpha=0.5, beta=2.0)
    np.testing.assert_allclose(tensor.to_numpy(tcn), 0.5 * (A @ B) + 2 * C,
                               rtol=1e-4, atol=1e-5)


def test_float_plus_scalar_keeps_float(dev):
    np.random.seed(0)
    x_val = np.random.random((2, 3)).astype(np.float32) * 10
    scalar = np.random.random((1,))[0] * 100
    y = T(x_val, dev) + scalar
    assert y.dtype == tensor.float32
    np.testing.assert_allclose(tensor.to_numpy(y), x_val + scalar, rtol=1e-6)


def test_int_plus_float_scalar_promotes(dev):
    np.random.seed(0)
    x_val = np.random.randint(0, 10, (2, 3)).astype(np.int32)
    scalar = random.random() * 100
    y = T(x_val, dev) + scalar
    assert y.dtype == tensor.float32
    np.testing.assert_allclose(tensor.to_numpy(y), x_val + scalar, rtol=1e-5)


def test_int_int(dev):
    a = np.random.randint(-25, 25, (3, 4, 5)).astype(np.int32)
    b = np.random.randint(-25, 25, (3, 4, 5)).astype(np.int32)
    y = T(a, dev) - T(b, dev)
    np.testing.assert_array_equal(tensor.to_numpy(y), a - b)
    bb = np.random.randint(-25, 25, (4, 5)).astype(np.int32)
    np.testing.assert_array_equal(tensor.to_numpy(T(a, dev) - T(bb, dev)),
                                  a - bb)


def test_broadcast_binary(dev):
    a = np.random.randn(3, 4, 5).astype(np.float32)
    for shp in [(5,), (4, 5), (1, 5), (4, 1), (3, 1, 1)]:
        b = np.random.randn(*shp).astype(np.float32) + 2
        np.testing.assert_allclose(tensor.to_numpy(T(a, dev) + T(b, dev)),
                                   a + b, rtol=1e-6)
        np.testing.assert_allclose(tensor.to_numpy(T(a, dev) * T(b, dev)),
                                   a * b, rtol=1e-6)
        np.testing.assert_allclose(tensor.to_numpy(T(a, dev) / T(b, dev)),
                                   a / b, rtol=1e-5)


def test_row_column_ops(dev):
    M = np.random.randn(3, 4).astype(np.float32)
    r = np.random.randn(4).astype(np.float32) + 3
    c = np.random.randn(3).astype(np.float32) + 3
    m = T(M, dev)
    m.add_row(T(r, dev))
    np.testing.assert_allclose(tensor.to_numpy(m), M + r, rtol=1e-6)
    m = T(M, dev)
    m.add_column(T(c, dev))
    np.testing.assert_allclose(tensor.to_numpy(m), M + c[:, None], rtol=1e-6)
    m = T(M, dev)
    m.mult_row(T(r, dev))
    np.testing.assert_allclose(tensor.to_numpy(m), M * r, rtol=1e-6)
    m = T(M, dev)
    m.div_column(T(c, dev))
    np.testing.assert_allclose(tensor.to_numpy(m), M / c[:, None], rtol=1e-5)
    m = T(M, dev)
    tensor.add_row(2.0, T(r, dev), 0.5, m)
    np.testing.assert_allclose(tensor.to_numpy(m), 2 * r + 0.5 * M, rtol=1e-5)


def test_axpy_norms(dev):
    x = np.random.randn(17).astype(np.float32)
    y = np.random.randn(17).astype(np.float32)
    ty = T(y, dev)
    tensor.axpy(0.5, T(x, dev), ty)
    np.testing.assert_allclose(tensor.to_numpy(ty), 0.5 * x + y, rtol=1e-5)
    tx = T(x, dev)
    assert tx.l1() == pytest.approx(np.abs(x).mean(), rel=1e-5)
    # reference tensor.cc:833-843: the 2-norm divided by the element count
    assert tx.l2() == pytest.approx(np.sqrt((x * x).sum()) / x.size, rel=1e-5)


def test_to_device_roundtrip(dev, cpu_dev):
    a = np.random.randn(4, 3).astype(np.float32)
    t = tensor.from_numpy(a)
    t.to_device(dev)
    assert t.device is dev
    u = t.clone()
    u.to_host()
    np.testing.assert_array_equal(tensor.to_numpy(u), a)
    z = tensor.zeros((2, 2), dev)
    o = tensor.ones((2, 2), dev)
    assert tensor.to_numpy(z).sum() == 0 and tensor.to_numpy(o).sum() == 4
    zl = tensor.zeros_like(t)
    assert zl.shape == t.shape and tensor.to_numpy(zl).sum() == 0


def test_setitem_and_contiguous(dev):
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    t = T(a, dev)
    tt = tensor.transpose(t, [2, 0, 1])
    c = tensor.contiguous(tt)
    assert not c.is_transpose()
    np.testing.assert_array_equal(tensor.to_numpy(c), a.transpose(2, 0, 1))
