"""autograd operators: forward against numpy, backward against central
differences (float64) of the numpy function.

The operator list follows the reference's test/python/test_operation.py; the
gradient check replaces its hand-written expected gradients.
"""
import numpy as np
import pytest

from singa_amd import autograd, tensor

RS = np.random.RandomState(7)


def to_t(a, dev, grad=True):
    t = tensor.Tensor(device=dev, data=np.asarray(a))
    t.requires_grad = grad
    t.stores_grad = grad
    return t


def scalar_loss(y, w):
    z = autograd.mul(y, w)
    l = autograd.reduce_sum(z, None, 0)
    return autograd.reshape(l, (1,))


def num_grad(fn, xs, i, w, eps=1e-3):
    xs = [np.array(x, dtype=np.float64) for x in xs]
    g = np.zeros_like(xs[i])
    it = np.nditer(xs[i], flags=["multi_index"])
    while not it.finished:
        idx = it.multi_index
        old = xs[i][idx]
        xs[i][idx] = old + eps
        fp = np.sum(fn(*xs) * w)
        xs[i][idx] = old - eps
        fm = np.sum(fn(*xs) * w)
        xs[i][idx] = old
        g[idx] = (fp - fm) / (2 * eps)
        it.iternext()
    return g


def check(dev, op, fn, xs, grad_of=None, rtol=1e-3, atol=1e-3, fwd_tol=1e-5):
    """op(*tensors) vs fn(*arrays); gradients of inputs listed in grad_of."""
    autograd.training = True
    xs = [np.asarray(x, dtype=np.float32) for x in xs]
    grad_of = list(range(len(xs))) if grad_of is None else grad_of
    ts = [to_t(x, dev, i in grad_of) for i, x in enumerate(xs)]
    y = op(*ts)
    ref = fn(*[x.astype(np.float64) for x in xs])
    out = tensor.to_numpy(y)
    assert out.shape == ref.shape, (out.shape, ref.shape)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=fwd_tol)
    if not grad_of:
        return
    # weights of the scalar loss: drawn from a stream of their own, so that a
    # case sees the same numbers whatever ran before it
    w = np.random.RandomState(1000 + ref.size).uniform(0.5, 1.5, ref.shape)
    grads = autograd.gradients(scalar_loss(y, to_t(w.astype(np.float32), dev,
                                                   False)))
    for i in grad_of:
        got = tensor.to_numpy(grads[id(ts[i])])
        want = num_grad(fn, xs, i, w)
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol,
                                   err_msg="grad of input %d" % i)


def sig(x):
    return 1 / (1 + np.exp(-x))


UNARY = [
    # name, numpy, low, high
    ("relu", lambda x: np.maximum(x, 0), -2, 2),
    ("tanh", np.tanh, -2, 2),
    ("sigmoid", sig, -3, 3),
    ("cos", np.cos, -3, 3), ("sin", np.sin, -3, 3), ("tan", np.tan, -1, 1),
    ("cosh", np.cosh, -2, 2), ("sinh", np.sinh, -2, 2),
    ("acos", np.arccos, -0.8, 0.8), ("asin", np.arcsin, -0.8, 0.8),
    ("atan", np.arctan, -2, 2), ("acosh", np.arccosh, 1.5, 3),
    ("asinh", np.arcsinh, -2, 2), ("atanh", np.arctanh, -0.8, 0.8),
    ("abs", np.abs, -2, 2), ("exp", np.exp, -2, 2), ("log", np.log, 0.5, 3),
    ("sqrt", np.sqrt, 0.5, 3),
    ("softplus", lambda x: np.log1p(np.exp(x)), -2, 2),
    ("softsign", lambda x: x / (1 + np.abs(x)), -2, 2),
    ("reciprocal", lambda x: 1 / x, 0.5, 3),
    ("negative", lambda x: -x, -2, 2),
    ("identity", lambda x: x, -2, 2),
    ("elu", lambda x: np.where(x > 0, x, np.exp(x) - 1), -2, 2),
    ("selu", lambda x: 1.0507 * np.where(x > 0, x, 1.67326 * (np.exp(x) - 1)),
     -2, 2),
    ("leakyrelu", lambda x: np.where(x > 0, x, 0.01 * x), -2, 2),
    ("hardsigmoid", lambda x: np.clip(0.2 * x + 0.5, 0, 1), -4, 4),
]


def _away_from_kinks(x):
    # keep samples off the points where the derivative jumps
    x = np.where(np.abs(x) < 0.05, 0.3, x)
    return np.where(np.abs(np.abs(x) - 2.5) < 0.05, 2.0, x)


@pytest.mark.parametrize("name,fn,lo,hi", UNARY, ids=[u[0] for u in UNARY])
def test_unary(dev, name, fn, lo, hi):
    x = _away_from_kinks(RS.uniform(lo, hi, (3, 4)))
    check(dev, getattr(autograd, name), fn, [x])


def test_erf(dev):
    from math import erf
    x = RS.uniform(-2, 2, (3, 4))
    check(dev, autograd.erf, np.vectorize(erf), [x])


@pytest.mark.parametrize("name,fn", [
    ("sign", np.sign), ("ceil", np.ceil), ("floor", np.floor),
    ("round", lambda x: np.sign(x) * np.floor(np.abs(x) + 0.5)),
    ("rounde", np.round)])
def test_step_functions(dev, name, fn):
    x = np.array([[-2.5, -1.5, -0.4, 0.5], [0.6, 1.5, 2.5, 3.49]])
    check(dev, getattr(autograd, name), fn, [x], grad_of=[])


BCAST = [((3, 4), (3, 4)), ((3, 4), (4,)), ((2, 3, 4), (3, 1)),
         ((1, 4), (3, 1))]


@pytest.mark.parametrize("sa,sb", BCAST)
@pytest.mark.parametrize("name,fn", [
    ("add", np.add), ("sub", np.subtract), ("mul", np.multiply),
    ("div", np.divide)])
def test_binary_broadcast(dev, name, fn, sa, sb):
    a = RS.uniform(0.5, 2, sa)
    b = RS.uniform(0.5, 2, sb)
    check(dev, getattr(autograd, name), fn, [a, b])


def test_pow(dev):
    a = RS.uniform(0.5, 2, (3, 4))
    b = RS.uniform(0.5, 2, (3, 4))
    check(dev, autograd.pow, np.power, [a, b])


@pytest.mark.parametrize("name,fn", [("less", np.less),
                                     ("greater", np.greater),
                                     ("equal", np.equal)])
def test_compare(dev, name, fn):
    a = np.array([[1., 2., 3.], [4., 5., 6.]])
    b = np.array([[1., 3., 2.], [4., 4., 7.]])
    check(dev, getattr(autograd, name), lambda x, y: fn(x, y).astype(float),
          [a, b], grad_of=[])


def test_min_max_mean_sum(dev):
    xs = [RS.uniform(-1, 1, (3, 4)) for _ in range(3)]
    check(dev, autograd.min, lambda *a: np.minimum.reduce(a), xs)
    check(dev, autograd.max, lambda *a: np.maximum.reduce(a), xs)
    check(dev, autograd.mean, lambda *a: sum(a) / len(a), xs)
    check(dev, autograd.sum, lambda *a: sum(a), xs)


def test_clip_prelu_where(dev):
    x = _away_from_kinks(RS.uniform(-2, 2, (3, 4)))
    x = np.where(np.abs(np.abs(x) - 1) < 0.05, 0.5, x)
    check(dev, lambda t: autograd.clip(t, -1, 1), lambda a: np.clip(a, -1, 1),
          [x])
    s = RS.uniform(0.1, 0.5, (4,))
    check(dev, autograd.prelu, lambda a, sl: np.where(a > 0, a, a * sl),
          [x, s])
    c = (RS.uniform(0, 1, (3, 4)) > 0.5).astype(np.float32)
    y = RS.uniform(-2, 2, (3, 4))
    check(dev, lambda a, b: autograd.where(a, b, to_t(c, dev, False)),
          lambda a, b: np.where(c > 0, a, b), [x, y])


def test_matmul_gemm_addbias(dev):
    a = RS.uniform(-1, 1, (3, 4))
    b = RS.uniform(-1, 1, (4, 5))
    c = RS.uniform(-1, 1, (3, 5))
    check(dev, autograd.matmul, np.matmul, [a, b])
    check(dev, lambda x, y, z: autograd.gemm(x, y, z, 0.5, 2.0, 0, 0),
          lambda x, y, z: 0.5 * x @ y + 2 * z, [a, b, c])
    check(dev, lambda x, y, z: autograd.gemm(x, y, z, 1.0, 1.0, 1, 1),
          lambda x, y, z: x.T @ y.T + z, [a.T, b.T, c])
    bias = RS.uniform(-1, 1, (5,))
    check(dev, lambda x, v: autograd.add_bias(x, v, 0), lambda x, v: x + v,
          [c, bias])
    ba = RS.uniform(-1, 1, (2, 3, 4))
    bb = RS.uniform(-1, 1, (2, 4, 5))
    check(dev, autograd.matmul, np.matmul, [ba, bb])


def _softmax(x, axis):
    e = np.exp(x - x.max(axis, keepdims=True))
    return e / e.sum(axis, keepdims=True)


def test_softmax_losses(dev):
    x = RS.uniform(-1, 1, (4, 5))
    check(dev, lambda t: autograd.softmax(t, 1), lambda a: _softmax(a, 1), [x])
    # axis 0 coerces the tensor to one row: soft-max over all elements
    check(dev, lambda t: autograd.softmax(t, 0),
          lambda a: _softmax(a.reshape(1, -1), 1).reshape(a.shape), [x])
    x3 = RS.uniform(-1, 1, (2, 3, 4))
    check(dev, lambda t: autograd.softmax(t, 2), lambda a: _softmax(a, 2),
          [x3])
    lab = np.array([1, 0, 4, 2], np.int32)
    onehot = np.eye(5)[lab]
    tl = tensor.Tensor(device=dev, data=lab)
    check(dev, lambda t: autograd.softmax_cross_entropy(t, tl),
          lambda a: np.reshape(
              -np.sum(onehot * np.log(_softmax(a, 1))) / 4, (1,)) if False
          else np.asarray(-np.sum(onehot * np.log(_softmax(a, 1))) / 4).reshape(
              tensor.to_numpy(autograd.softmax_cross_entropy(
                  to_t(x, dev), tl)).shape), [x])
    toh = to_t(onehot.astype(np.float32), dev, False)
    p = _softmax(x, 1)
    check(dev, lambda t: autograd.cross_entropy(t, toh),
          lambda a: np.asarray(-np.sum(onehot * np.log(a)) / 4).reshape(
              tensor.to_numpy(autograd.cross_entropy(to_t(p, dev),
                                                     toh)).shape), [p])
    t = RS.uniform(0, 1, (4, 5))
    tt = to_t(t.astype(np.float32), dev, False)
    check(dev, lambda a: autograd.mse_loss(a, tt),
          lambda a: np.asarray(np.sum((a - t) ** 2) / (2 * 4) if False else
                               np.sum((a - t) ** 2) / a.size).reshape(
              tensor.to_numpy(autograd.mse_loss(to_t(x, dev), tt)).shape), [x])
    q = RS.uniform(0.1, 0.9, (4, 5))
    check(dev, lambda a: autograd.binary_cross_entropy(a, tt),
          lambda a: np.asarray(
              -np.sum(t * np.log(a + 1e-4) + (1 - t) * np.log(1 - a + 1e-4)) / 4).reshape(
              tensor.to_numpy(autograd.binary_cross_entropy(to_t(q, dev),
                                                            tt)).shape), [q])


def test_shape_ops(dev):
    x = RS.uniform(-1, 1, (2, 3, 4))
    check(dev, lambda t: autograd.reshape(t, (4, 6)),
          lambda a: a.reshape(4, 6), [x])
    check(dev, lambda t: autograd.reshape(t, (0, -1)),
          lambda a: a.reshape(2, 12), [x])
    check(dev, lambda t: autograd.flatten(t, 2), lambda a: a.reshape(6, 4),
          [x])
    check(dev, lambda t: autograd.unsqueeze(t, [0, 2]),
          lambda a: a.reshape(1, 2, 1, 3, 4), [x])
    check(dev, lambda t: autograd.squeeze(t, [1]),
          lambda a: a.reshape(2, 4), [x[:, :1, :]])
    check(dev, lambda t: autograd.transpose(t, (2, 0, 1)),
          lambda a: a.transpose(2, 0, 1), [x])
    y = RS.uniform(-1, 1, (2, 2, 4))
    check(dev, lambda a, b: autograd.cat([a, b], 1),
          lambda a, b: np.concatenate([a, b], 1), [x, y])
    check(dev, lambda t: autograd.slice(t, [0, 1], [2, 3], [1, 2]),
          lambda a: a[:, 0:2, 1:3], [x])
    check(dev, lambda t: autograd.slice(t, [0], [4], [2], [2]),
          lambda a: a[:, :, 0:4:2], [x])
    parts = autograd.split(to_t(x, dev), 2, [1, 3])
    np.testing.assert_allclose(tensor.to_numpy(parts[0]), x[:, :, :1],
                               rtol=1e-6)
    np.testing.assert_allclose(tensor.to_numpy(parts[1]), x[:, :, 1:],
                               rtol=1e-6)


def test_gather_tile_expand_pad(dev):
    x = RS.uniform(-1, 1, (3, 4))
    check(dev, lambda t: autograd.gather(t, 1, [0, 3, 3]),
          lambda a: np.take(a, [0, 3, 3], 1), [x])
    check(dev, lambda t: autograd.gather(t, 0, [[2, 0], [1, 1]]),
          lambda a: np.take(a, [[2, 0], [1, 1]], 0), [x])
    check(dev, lambda t: autograd.tile(t, [2, 3]), lambda a: np.tile(a, (2, 3)),
          [x])
    check(dev, lambda t: autograd.expand(t, [2, 3, 4]),
          lambda a: np.broadcast_to(a, (2, 3, 4)) * 1.0, [x])
    check(dev, lambda t: autograd.pad(t, "constant", [1, 0, 2, 1], 0.5),
          lambda a: np.pad(a, ((1, 2), (0, 1)), "constant",
                           constant_values=0.5), [x])
    check(dev, lambda t: autograd.pad(t, "reflect", [1, 1, 1, 1]),
          lambda a: np.pad(a, ((1, 1), (1, 1)), "reflect"), [x])
    check(dev, lambda t: autograd.pad(t, "edge", [1, 1, 1, 1]),
          lambda a: np.pad(a, ((1, 1), (1, 1)), "edge"), [x])


def test_reduce_ops(dev):
    x = RS.uniform(-1, 1, (2, 3, 4))
    check(dev, lambda t: autograd.reduce_sum(t, [1], 1),
          lambda a: a.sum(1, keepdims=True), [x])
    check(dev, lambda t: autograd.reduce_sum(t, [0, 2], 0),
          lambda a: a.sum((0, 2)), [x])
    check(dev, lambda t: autograd.reduce_mean(t, [1, 2], 1),
          lambda a: a.mean((1, 2), keepdims=True), [x])
    x4 = RS.uniform(-1, 1, (2, 3, 4, 4))
    check(dev, autograd.globalaveragepool,
          lambda a: a.mean((2, 3), keepdims=True), [x4])


def test_resample_ops(dev):
    x = RS.uniform(-1, 1, (1, 4, 2, 3))
    check(dev, lambda t: autograd.upsample(t, "nearest", [1, 1, 2, 3]),
          lambda a: a.repeat(2, 2).repeat(3, 3), [x])

    def d2s(a, bs=2):
        b, c, h, w = a.shape
        t = a.reshape(b, bs, bs, c // (bs * bs), h, w)
        t = t.transpose(0, 3, 4, 1, 5, 2)
        return t.reshape(b, c // (bs * bs), h * bs, w * bs)

    check(dev, lambda t: autograd.depth_to_space(t, 2, "DCR"), d2s, [x])
    y = RS.uniform(-1, 1, (1, 1, 4, 6))

    def s2d(a, bs=2):
        b, c, h, w = a.shape
        t = a.reshape(b, c, h // bs, bs, w // bs, bs)
        t = t.transpose(0, 3, 5, 1, 2, 4)
        return t.reshape(b, c * bs * bs, h // bs, w // bs)

    check(dev, lambda t: autograd.space_to_depth(t, 2, "DCR"), s2d, [y])


def test_misc_ops(dev):
    x = RS.uniform(-1, 1, (2, 3))
    t = to_t(x, dev)
    np.testing.assert_array_equal(tensor.to_numpy(autograd.shape(t)), [2, 3])
    c = autograd.constant_of_shape(
        tensor.Tensor(device=dev, data=np.array([2, 2], np.int32)), 1.5)
    np.testing.assert_array_equal(tensor.to_numpy(c), np.full((2, 2), 1.5))
    ci = autograd.cast(t, tensor.int32)
    assert ci.dtype == tensor.int32
    np.testing.assert_array_equal(tensor.to_numpy(ci), x.astype(np.int32))
    idx = tensor.Tensor(device=dev, data=np.array([0, 2, 1], np.int32))
    oh = autograd.onehot(-1, idx, 3, np.array([0, 1], np.float32))
    np.testing.assert_array_equal(tensor.to_numpy(oh), np.eye(3)[[0, 2, 1]])
    z = np.array([[1., 0.], [0., 2.]], np.float32)
    nz = autograd.nonzero(to_t(z, dev, False))
    np.testing.assert_array_equal(tensor.to_numpy(nz), np.array(np.nonzero(z)))
    a = RS.uniform(-1, 1, (4, 5))
    b = RS.uniform(-1, 1, (4, 5))
    check(dev, autograd.cossim,
          lambda p, q: np.sum(p * q, 1) / np.sqrt(np.sum(p * p, 1) *
                                                   np.sum(q * q, 1)), [a, b])
    upd = RS.uniform(-1, 1, (2, 3))
    ind = np.array([[1, 0, 2], [0, 2, 1]], np.int32)
    ref = np.array(x.repeat(2, 0)[:3] * 0 + 0)
    base = RS.uniform(-1, 1, (3, 3))
    want = base.copy()
    for i in range(2):
        for j in range(3):
            want[ind[i, j], j] = upd[i, j]
    got = autograd.scatter_elements(to_t(base, dev), ind, to_t(upd, dev), 0)
    np.testing.assert_allclose(tensor.to_numpy(got), want, rtol=1e-6)


def test_embedding(dev):
    w = RS.uniform(-1, 1, (6, 3))
    ids = np.array([[0, 2, 5], [1, 1, 4]], np.int32)
    ti = tensor.Tensor(device=dev, data=ids)
    check(dev, lambda t: autograd.embedding(ti, t), lambda a: a[ids], [w])


def test_dropout(dev):
    autograd.training = True
    x = to_t(np.ones((64, 64), np.float32), dev)
    y = autograd.dropout(x, ratio=0.25)
    v = tensor.to_numpy(y)
    kept = (v != 0).mean()
    assert kept == pytest.approx(0.75, abs=0.05)
    np.testing.assert_allclose(v[v != 0], 1 / 0.75, rtol=1e-5)
    autograd.training = False
    np.testing.assert_array_equal(tensor.to_numpy(autograd.dropout(x)), 1.0)
    autograd.training = True


def test_ranking_loss(dev):
    pos = RS.uniform(0, 1, (6, 1))
    neg = RS.uniform(0, 1, (6, 1))
    neg = np.where(np.abs(0.2 - pos + neg) < 0.05, neg + 0.2, neg)
    check(dev, lambda p, n: autograd.ranking_loss(p, n, 0.2),
          lambda p, n: np.asarray(np.mean(np.maximum(0.2 - p + n, 0))).reshape(
              tensor.to_numpy(autograd.ranking_loss(
                  to_t(pos, dev), to_t(neg, dev), 0.2)).shape), [pos, neg])


def test_graph_keys_survive_garbage_collection():
    """Tensors are keyed in the autograd graph by a serial number, so a
    temporary that is collected during the forward pass cannot lend its
    identity to a parameter created afterwards (every parameter must get
    its gradient)."""
    import gc
    from singa_amd import layer
    autograd.training = True
    seen = set()
    for _ in range(50):
        t = tensor.Tensor((1,))
        assert t.uid not in seen
        seen.add(t.uid)
        del t
        gc.collect()
    x = tensor.from_numpy(np.random.randn(2, 4, 9, 9).astype(np.float32))
    c_in = layer.Conv2d(8, 3, padding=1, bias=False)
    c_a = layer.Conv2d(8, 3, padding=1, bias=False)
    c_b = layer.Conv2d(8, 1, bias=True)
    c_s = layer.Conv2d(8, 3, padding=1, stride=2, bias=False)
    x0 = c_in(x)
    y = autograd.add(autograd.add(c_a(x0), c_b(x0)), x0)
    z = c_s(x0)
    loss = autograd.add(autograd.mse_loss(y, tensor.Tensor(y.shape)),
                        autograd.mse_loss(z, tensor.Tensor(z.shape)))
    assert len(list(autograd.backward(loss))) == 5


def _layernorm_ref(x, g, b, eps):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + eps) * g + b


@pytest.mark.parametrize("shape", [(4, 7, 48), (3, 1000), (2, 5, 3000)])
def test_layer_norm_forward_backward(dev, shape):
    autograd.training = True
    rng = np.random.RandomState(1)
    x = rng.randn(*shape).astype(np.float32) * 2 + 0.5
    g = rng.rand(shape[-1]).astype(np.float32) + 0.5
    b = rng.randn(shape[-1]).astype(np.float32)
    w = rng.randn(*shape).astype(np.float32)
    tx = tensor.from_numpy(x, dev)
    tx.requires_grad = True
    tx.stores_grad = True
    tg, tb = tensor.from_numpy(g, dev), tensor.from_numpy(b, dev)
    for t in (tg, tb):
        t.requires_grad = True
        t.stores_grad = True
    y = autograd.layer_norm(tx, tg, tb, 1e-5)
    np.testing.assert_allclose(tensor.to_numpy(y), _layernorm_ref(x, g, b, 1e-5),
                               rtol=1e-4, atol=1e-4)
    loss = autograd.reduce_sum(autograd.mul(y, tensor.from_numpy(w, dev)))
    grads = {id(p): tensor.to_numpy(gr) for p, gr in autograd.backward(loss)}
    import torch
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = torch.nn.functional.layer_norm(xt, (shape[-1],), gt, bt, 1e-5)
    (yt * torch.tensor(w, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(grads[id(tx)], xt.grad.numpy(), rtol=2e-3,
                               atol=2e-4)
    np.testing.assert_allclose(grads[id(tg)], gt.grad.numpy(), rtol=2e-3,
                               atol=2e-3)
    np.testing.assert_allclose(grads[id(tb)], bt.grad.numpy(), rtol=2e-3,
                               atol=2e-3)


def test_gelu_forward_backward(dev):
    autograd.training = True
    import torch
    x = np.linspace(-4, 4, 1001).astype(np.float32).reshape(7, 143)
    tx = tensor.from_numpy(x, dev)
    tx.requires_grad = True
    tx.stores_grad = True
    y = autograd.gelu(tx)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = torch.nn.functional.gelu(xt)
    np.testing.assert_allclose(tensor.to_numpy(y), yt.detach().numpy(),
                               rtol=1e-5, atol=1e-6)
    loss = autograd.reduce_sum(y)
    (dx,) = [tensor.to_numpy(g) for _, g in autograd.backward(loss)]
    yt.sum().backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_layer_norm_layer_trains(dev):
    from singa_amd import layer, opt
    autograd.training = True
    rng = np.random.RandomState(2)
    x = tensor.from_numpy(rng.randn(16, 32).astype(np.float32), dev)
    t = tensor.from_numpy(rng.randn(16, 32).astype(np.float32), dev)
    ln, fc, act = layer.LayerNorm(), layer.Linear(32), layer.Gelu()
    sgd = opt.SGD(lr=0.1)
    losses = []
    for _ in range(60):
        loss = autograd.mse_loss(fc(act(ln(x))), t)
        sgd(loss)
        losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    assert losses[-1] < 0.95 * losses[0], losses
    assert set(ln.get_params()) == {ln.scale.name, ln.bias.name}
