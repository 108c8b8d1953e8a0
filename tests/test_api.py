"""The flat native-API namespace (``singa_wrap``) called the way the
reference's SWIG module is called: batch-norm / pooling handles with their
free functions, softmax along an axis, broadcast arithmetic on raw tensors,
erf, concat, rounding family, type conversion.  Cases follow reference
test/python/test_api.py; results are checked against numpy."""
import math

import numpy as np
import pytest

from singa_amd import device, tensor
from singa_amd import singa_wrap as singa_api


def _devices():
    out = [pytest.param("cpu", id="cpu")]
    out.append(pytest.param("gpu", id="gpu", marks=pytest.mark.gpu))
    return out


@pytest.fixture(params=_devices())
def dev(request):
    if request.param == "gpu":
        return device.create_cuda_gpu()
    return device.get_default_device()


def _to(a, dev):
    t = tensor.from_numpy(np.ascontiguousarray(a))
    t.to_device(dev)
    return t.data


def _np(raw):
    t = tensor.Tensor()
    t.data = raw
    t.shape = tuple(raw.shape())
    t.device = raw.device()
    t.dtype = raw.data_type()
    return tensor.to_numpy(t)


def _bn_fns(dev):
    gpu = dev.lang() != singa_api.kCpp if hasattr(dev, "lang") else False
    if gpu:
        return (singa_api.CudnnBatchNormHandle,
                singa_api.GpuBatchNormForwardTraining,
                singa_api.GpuBatchNormForwardInference)
    return (singa_api.BatchNormHandle, singa_api.CpuBatchNormForwardTraining,
            singa_api.CpuBatchNormForwardInference)


def test_batchnorm_training_and_testing(dev):
    rng = np.random.RandomState(0)
    x = rng.randn(4, 3, 5, 5).astype(np.float32) * 2 + 1
    scale = rng.rand(3).astype(np.float32) + 0.5
    bias = rng.randn(3).astype(np.float32)
    rm0 = rng.randn(3).astype(np.float32)
    rv0 = rng.rand(3).astype(np.float32) + 0.5
    momentum = 0.1
    Handle, train, infer = _bn_fns(dev)
    tx = _to(x, dev)
    rm, rv = _to(rm0, dev), _to(rv0, dev)
    h = Handle(momentum, tx)
    out = train(h, tx, _to(scale, dev), _to(bias, dev), rm, rv)
    y = _np(out[0])
    m = x.mean((0, 2, 3), keepdims=True)
    v = x.var((0, 2, 3), keepdims=True)
    ref = (x - m) / np.sqrt(v + 1e-5) * scale.reshape(1, 3, 1, 1) + \
        bias.reshape(1, 3, 1, 1)
    np.testing.assert_allclose(y, ref, atol=2e-5, rtol=1e-4)
    n = x.size // 3
    np.testing.assert_allclose(
        _np(rm), momentum * m.reshape(3) + (1 - momentum) * rm0, atol=1e-5)
    np.testing.assert_allclose(
        _np(rv), momentum * v.reshape(3) * n / (n - 1) + (1 - momentum) * rv0,
        atol=1e-4)
    yi = _np(infer(h, tx, _to(scale, dev), _to(bias, dev), _to(rm0, dev),
                   _to(rv0, dev)))
    refi = (x - rm0.reshape(1, 3, 1, 1)) / np.sqrt(
        rv0.reshape(1, 3, 1, 1) + 1e-5) * scale.reshape(1, 3, 1, 1) + \
        bias.reshape(1, 3, 1, 1)
    np.testing.assert_allclose(yi, refi, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("axis", [0, 1, 2, 3, -1])
def test_softmax_api(dev, axis):
    rng = np.random.RandomState(1)
    x = rng.randn(2, 3, 4, 5).astype(np.float32)
    y = _np(singa_api.SoftMax(_to(x, dev), axis))
    # the native call treats the tensor as a matrix: the dimensions before
    # `axis` are rows, everything from `axis` on is one softmax row
    ax = axis % x.ndim
    m = x.reshape(int(np.prod(x.shape[:ax])), -1)
    e = np.exp(m - m.max(1, keepdims=True))
    ref = (e / e.sum(1, keepdims=True)).reshape(x.shape)
    np.testing.assert_allclose(y, ref, atol=1e-6)


@pytest.mark.parametrize("shapes", [((2, 3, 4, 5), (5,)),
                                    ((3, 1, 2, 1), (3, 4, 2, 6)),
                                    ((4, 5), (2, 3, 4, 5)),
                                    ((1,), (3, 2))])
def test_tensor_arithmetic_broadcast(dev, shapes):
    rng = np.random.RandomState(2)
    a = rng.randn(*shapes[0]).astype(np.float32)
    b = rng.rand(*shapes[1]).astype(np.float32) + 0.5
    ta, tb = _to(a, dev), _to(b, dev)
    for fn, ref in ((singa_api.__add__, a + b), (singa_api.__sub__, a - b),
                    (singa_api.__mul__, a * b), (singa_api.__div__, a / b)):
        np.testing.assert_allclose(_np(fn(ta, tb)), ref, rtol=1e-5, atol=1e-6)


def test_transpose_then_broadcast(dev):
    rng = np.random.RandomState(3)
    a = rng.randn(3, 4).astype(np.float32)
    b = rng.randn(4, 1).astype(np.float32)
    ta = _to(a, dev).T() if callable(getattr(singa_api.Tensor, "T")) \
        else _to(a, dev).T
    np.testing.assert_allclose(_np(singa_api.__add__(ta, _to(b, dev))),
                               a.T + b, rtol=1e-6)


def test_erf_and_rounding_family(dev):
    x = np.array([-2.5, -1.5, -0.5, -0.2, 0.0, 0.2, 0.5, 1.5, 2.5, 3.7],
                 np.float32)
    t = _to(x, dev)
    np.testing.assert_allclose(_np(singa_api.Erf(t)),
                               np.array([math.erf(v) for v in x], np.float32),
                               atol=1e-6)
    np.testing.assert_array_equal(_np(singa_api.Ceil(t)), np.ceil(x))
    np.testing.assert_array_equal(_np(singa_api.Floor(t)), np.floor(x))
    # Round: halves away from zero; RoundE: halves to even
    np.testing.assert_array_equal(
        _np(singa_api.Round(t)), np.sign(x) * np.floor(np.abs(x) + 0.5))
    np.testing.assert_array_equal(_np(singa_api.RoundE(t)), np.round(x))


def test_concat(dev):
    rng = np.random.RandomState(4)
    parts = [rng.randn(2, n, 3).astype(np.float32) for n in (1, 4, 2)]
    out = singa_api.ConcatOn(singa_api.VecTensor([_to(p, dev) for p in parts]),
                             1)
    np.testing.assert_array_equal(_np(out), np.concatenate(parts, 1))


def test_as_type(dev):
    x = np.array([-1.7, -0.2, 0.0, 0.9, 2.6, 300.2], np.float32)
    t = _to(x, dev)
    ti = t.AsType(singa_api.kInt)
    assert ti.data_type() == singa_api.kInt
    np.testing.assert_array_equal(_np(ti), x.astype(np.int32))
    tf = ti.AsType(singa_api.kFloat32)
    assert tf.data_type() == singa_api.kFloat32
    np.testing.assert_array_equal(_np(tf), x.astype(np.int32).astype(np.float32))


def test_pooling_handle(dev):
    rng = np.random.RandomState(5)
    x = rng.randn(2, 3, 6, 6).astype(np.float32)
    gpu = dev.lang() != singa_api.kCpp if hasattr(dev, "lang") else False
    Handle = singa_api.CudnnPoolingHandle if gpu else singa_api.PoolingHandle
    fwd = singa_api.GpuPoolingForward if gpu else singa_api.CpuPoolingForward
    tx = _to(x, dev)
    for is_max in (True, False):
        h = Handle(tx, [2, 2], [2, 2], [0, 0], is_max)
        y = _np(fwd(h, tx))
        blocks = x.reshape(2, 3, 3, 2, 3, 2)
        ref = blocks.max((3, 5)) if is_max else blocks.mean((3, 5))
        np.testing.assert_allclose(y, ref, atol=1e-6)
