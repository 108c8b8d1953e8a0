"""Numerical gradient check of the differentiable autograd operators: for each
operator a scalar objective mse(op(x...), w) is differentiated by central
differences on the host and compared with ``autograd.backward``.  Complements
the value tests of test_operation.py (reference test/python/test_operation.py
checks the analytic gradients op by op against hand-written formulas)."""
import numpy as np
import pytest

from singa_amd import autograd, device, tensor

A = autograd
rs = np.random.RandomState(0)


def R(*shape, lo=-1.0, hi=1.0):
    return (rs.rand(*shape) * (hi - lo) + lo).astype(np.float32)


def away_from(x, points, margin=0.08):
    """Nudge entries away from non-differentiable points."""
    x = x.copy()
    for p in points:
        close = np.abs(x - p) < margin
        x[close] = p + margin * 2 * np.where(x[close] >= p, 1, -1)
    return x


# (name, function of tensors, list of input arrays)
CASES = [
    ("relu", A.relu, [away_from(R(3, 4), [0])]),
    ("leakyrelu", lambda x: A.leakyrelu(x, 0.1), [away_from(R(3, 4), [0])]),
    ("elu", lambda x: A.elu(x, 1.0), [away_from(R(3, 4), [0])]),
    ("selu", A.selu, [away_from(R(3, 4), [0])]),
    ("hardsigmoid", lambda x: A.hardsigmoid(x, 0.2, 0.5), [R(3, 4)]),
    ("prelu", A.prelu, [away_from(R(3, 4), [0]), R(4, lo=0.1, hi=0.5)]),
    ("clip", lambda x: A.clip(x, -0.5, 0.5),
     [away_from(R(3, 4), [-0.5, 0.5])]),
    ("tanh", A.tanh, [R(3, 4)]),
    ("sigmoid", A.sigmoid, [R(3, 4)]),
    ("softplus", A.softplus, [R(3, 4)]),
    ("softsign", A.softsign, [R(3, 4)]),
    ("cos", A.cos, [R(3, 4)]), ("sin", A.sin, [R(3, 4)]),
    ("tan", A.tan, [R(3, 4, lo=-0.8, hi=0.8)]),
    ("cosh", A.cosh, [R(3, 4)]), ("sinh", A.sinh, [R(3, 4)]),
    ("acos", A.acos, [R(3, 4, lo=-0.7, hi=0.7)]),
    ("asin", A.asin, [R(3, 4, lo=-0.7, hi=0.7)]),
    ("atan", A.atan, [R(3, 4)]),
    ("acosh", A.acosh, [R(3, 4, lo=1.3, hi=2.5)]),
    ("asinh", A.asinh, [R(3, 4)]),
    ("atanh", A.atanh, [R(3, 4, lo=-0.7, hi=0.7)]),
    ("abs", A.abs, [away_from(R(3, 4), [0])]),
    ("exp", A.exp, [R(3, 4)]),
    ("log", A.log, [R(3, 4, lo=0.5, hi=2.0)]),
    ("sqrt", A.sqrt, [R(3, 4, lo=0.5, hi=2.0)]),
    ("reciprocal", A.reciprocal, [R(3, 4, lo=0.5, hi=2.0)]),
    ("erf", A.erf, [R(3, 4)]),
    ("negative", A.negative, [R(3, 4)]),
    ("identity", A.identity, [R(3, 4)]),
    ("add", A.add, [R(3, 4), R(3, 4)]),
    ("add_bcast", A.add, [R(3, 4), R(4)]),
    ("sub", A.sub, [R(3, 4), R(3, 1)]),
    ("mul", A.mul, [R(3, 4), R(1, 4)]),
    ("div", A.div, [R(3, 4), R(3, 4, lo=0.5, hi=2.0)]),
    ("pow", A.pow, [R(3, 4, lo=0.5, hi=2.0), R(3, 4, lo=0.5, hi=2.0)]),
    ("min", A.min, [R(3, 4), R(3, 4) + 0.05]),
    ("max", A.max, [R(3, 4), R(3, 4) + 0.05]),
    ("mean", A.mean, [R(3, 4), R(3, 4), R(3, 4)]),
    ("sum", A.sum, [R(3, 4), R(3, 4)]),
    ("matmul", A.matmul, [R(3, 4), R(4, 5)]),
    ("matmul_batched", A.matmul, [R(2, 3, 4), R(2, 4, 5)]),
    ("add_bias", lambda x, b: A.add_bias(x, b, 0), [R(3, 4), R(4)]),
    ("gemm", lambda a, b, c: A.gemm(a, b, c, 0.5, 2.0, 0, 1),
     [R(3, 4), R(5, 4), R(3, 5)]),
    ("softmax", lambda x: A.softmax(x, 1), [R(3, 5)]),
    ("cossim", A.cossim, [R(3, 4), R(3, 4)]),
    ("reshape", lambda x: A.reshape(x, (4, 3)), [R(3, 4)]),
    ("flatten", lambda x: A.flatten(x, 1), [R(2, 3, 2)]),
    ("unsqueeze", lambda x: A.unsqueeze(x, [0, 2]), [R(3, 4)]),
    ("squeeze", lambda x: A.squeeze(x, [1]), [R(3, 1, 4)]),
    ("transpose", lambda x: A.transpose(x, (2, 0, 1)), [R(2, 3, 4)]),
    ("cat", lambda a, b: A.cat([a, b], 1), [R(2, 3), R(2, 2)]),
    ("slice", lambda x: A.slice(x, [0, 1], [2, 4], [0, 1], [1, 2]),
     [R(3, 5)]),
    ("split", lambda x: A.split(x, 1, [2, 3])[1], [R(2, 5)]),
    ("gather", lambda x: A.gather(x, 1, [2, 0, 2]), [R(3, 4)]),
    ("tile", lambda x: A.tile(x, [2, 3]), [R(2, 3)]),
    ("expand", lambda x: A.expand(x, [3, 2, 4]), [R(2, 1)]),
    ("pad", lambda x: A.pad(x, "constant", [1, 0, 2, 1], 0.0), [R(2, 3)]),
    ("upsample", lambda x: A.upsample(x, "nearest", [1.0, 1.0, 2.0, 2.0]),
     [R(1, 2, 2, 3)]),
    ("depth_to_space", lambda x: A.depth_to_space(x, 2, "DCR"),
     [R(1, 8, 2, 3)]),
    ("space_to_depth", lambda x: A.space_to_depth(x, 2), [R(1, 2, 4, 6)]),
    ("reduce_sum", lambda x: A.reduce_sum(x, [1], 1), [R(3, 4, 2)]),
    ("reduce_mean", lambda x: A.reduce_mean(x, [0, 2], 0), [R(3, 4, 2)]),
    ("globalaveragepool", A.globalaveragepool, [R(2, 3, 4, 4)]),
    ("where", lambda a, b: A.where(
        a, b, (np.arange(12).reshape(3, 4) % 2).astype(np.float32)),
     [R(3, 4), R(3, 4)]),
]


def _objective(fn, arrays, weights, dev):
    ts = [tensor.Tensor(device=dev, data=a, requires_grad=True,
                        stores_grad=True) for a in arrays]
    y = fn(*ts)
    w = tensor.Tensor(device=dev, data=weights)
    return A.mse_loss(y, w), ts, y


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_gradient_matches_finite_differences(case):
    name, fn, arrays = case
    dev = device.get_default_device()
    autograd.training = True
    try:
        arrays = [a.astype(np.float32) for a in arrays]
        y0 = fn(*[tensor.Tensor(device=dev, data=a) for a in arrays])
        weights = np.random.RandomState(1).randn(*y0.shape).astype(np.float32)
        loss, ts, _ = _objective(fn, arrays, weights, dev)
        grads = {id(p): tensor.to_numpy(g) for p, g in A.backward(loss)}
        eps = 1e-2
        for k, (a, t) in enumerate(zip(arrays, ts)):
            assert id(t) in grads, "no gradient for input %d of %s" % (k, name)
            g = grads[id(t)]
            assert g.shape == a.shape
            flat = a.reshape(-1)
            idxs = np.linspace(0, flat.size - 1, min(flat.size, 6)).astype(int)
            for i in idxs:
                vals = []
                for sgn in (+1, -1):
                    b = [x.copy() for x in arrays]
                    b[k].reshape(-1)[i] += sgn * eps
                    l, _, _ = _objective(fn, b, weights, dev)
                    vals.append(float(tensor.to_numpy(l).reshape(-1)[0]))
                num = (vals[0] - vals[1]) / (2 * eps)
                ana = float(g.reshape(-1)[i])
                assert abs(num - ana) <= 3e-2 * (1 + abs(num)), \
                    "%s input %d element %d: numeric %g analytic %g" % (
                        name, k, i, num, ana)
    finally:
        autograd.training = False
