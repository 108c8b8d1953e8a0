"""Per-operator cases that the reference's test/python/test_operation.py lists
by name and tests/test_operation.py folds into parametrised groups or leaves
out: each broadcasting binary operator on its own shape list, 1- and 3-input
min / max, the losses by value, SAME-padded convolution and pooling (1-d and
2-d), separable convolution, depth / space, scatter, cosine similarity,
non-zero, invalid inputs and the recurrent layers' shape and gradient checks.
Forward against numpy / torch, backward against central differences.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from singa_amd import autograd, layer, tensor
from test_operation import check, num_grad, scalar_loss, to_t

RS = np.random.RandomState(19)

# the reference's shape pairs (test_operation.py: *_broadcast tests)
BCAST = [((3, 4, 5), (5,)), ((3, 4, 5), (4, 5)), ((3, 4, 5), (1, 4, 1)),
         ((1, 4, 5, 6), (5, 6)), ((3, 4, 5, 6), (4, 5, 6)),
         ((2, 1, 5), (1, 4, 1)), ((5,), (3, 4, 5))]


def _pair(sa, sb, lo=-1.0, hi=1.0):
    return (RS.uniform(lo, hi, sa).astype(np.float32),
            RS.uniform(lo, hi, sb).astype(np.float32))


@pytest.mark.parametrize("sa,sb", BCAST)
def test_add_broadcast(dev, sa, sb):
    check(dev, autograd.add, np.add, _pair(sa, sb))


@pytest.mark.parametrize("sa,sb", BCAST)
def test_sub_broadcast(dev, sa, sb):
    check(dev, autograd.sub, np.subtract, _pair(sa, sb))


@pytest.mark.parametrize("sa,sb", BCAST)
def test_mul_broadcast(dev, sa, sb):
    check(dev, autograd.mul, np.multiply, _pair(sa, sb))


@pytest.mark.parametrize("sa,sb", BCAST)
def test_div_broadcast(dev, sa, sb):
    a, b = _pair(sa, sb, 0.5, 2.0)
    check(dev, autograd.div, np.divide, (a, b))


@pytest.mark.parametrize("sa,sb", BCAST)
def test_pow_broadcast(dev, sa, sb):
    a, b = _pair(sa, sb, 0.5, 2.0)
    check(dev, autograd.pow, np.power, (a, b), rtol=2e-3)


@pytest.mark.parametrize("sa,sb", BCAST[:5])
def test_prelu_broadcast(dev, sa, sb):
    x = RS.uniform(-1, 1, sa).astype(np.float32)
    x[np.abs(x) < 0.05] = 0.3          # keep clear of the kink
    slope = RS.uniform(0.1, 0.5, sb).astype(np.float32)
    check(dev, autograd.prelu, lambda a, s: np.where(a > 0, a, a * s),
          (x, slope))


@pytest.mark.parametrize("name,fn", [
    ("less", np.less), ("greater", np.greater), ("equal", np.equal)])
@pytest.mark.parametrize("sa,sb", BCAST)
def test_compare_broadcast(dev, name, fn, sa, sb):
    a = RS.randint(0, 3, sa).astype(np.float32)
    b = RS.randint(0, 3, sb).astype(np.float32)
    check(dev, getattr(autograd, name),
          lambda x, y: fn(x, y).astype(np.float64), (a, b), grad_of=[])


@pytest.mark.parametrize("name,fn", [
    ("_and", np.logical_and), ("_or", np.logical_or),
    ("_xor", np.logical_xor)])
@pytest.mark.parametrize("sa,sb", [((3, 4), (3, 4))] + BCAST)
def test_logical_and_or_xor_broadcast(dev, name, fn, sa, sb):
    a = (RS.rand(*sa) > 0.5).astype(np.float32)
    b = (RS.rand(*sb) > 0.5).astype(np.float32)
    check(dev, getattr(autograd, name),
          lambda x, y: fn(x, y).astype(np.float64), (a, b), grad_of=[])


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("name,fn", [("max", np.maximum), ("min", np.minimum)])
def test_min_max_n_inputs(dev, name, fn, n):
    # distinct values: the sub-gradient at ties is a convention
    base = RS.permutation(60 * n).astype(np.float32).reshape(n, 3, 4, 5) / 10
    xs = [base[i] for i in range(n)]

    def ref(*a):
        out = a[0]
        for b in a[1:]:
            out = fn(out, b)
        return out
    check(dev, getattr(autograd, name), ref, xs)


def test_mse_loss_value(dev):
    x = RS.randn(6, 5).astype(np.float32)
    t = RS.randn(6, 5).astype(np.float32)
    autograd.training = True
    tx = to_t(x, dev)
    loss = autograd.mse_loss(tx, to_t(t, dev, False))
    # reference convention: mean over all elements
    want = np.mean((x - t) ** 2)
    np.testing.assert_allclose(tensor.to_numpy(loss).reshape(-1)[0], want,
                               rtol=1e-5)
    g = autograd.gradients(loss)[id(tx)]
    np.testing.assert_allclose(tensor.to_numpy(g), 2 * (x - t) / x.size,
                               rtol=1e-5, atol=1e-6)


def test_mean_square_error_layer(dev):
    x = RS.randn(4, 3).astype(np.float32)
    t = RS.randn(4, 3).astype(np.float32)
    loss = layer.MeanSquareError()(to_t(x, dev), to_t(t, dev, False))
    np.testing.assert_allclose(tensor.to_numpy(loss).reshape(-1)[0],
                               np.mean((x - t) ** 2), rtol=1e-5)


def test_cossim_value_and_gradient(dev):
    a = RS.randn(5, 7).astype(np.float32)
    b = RS.randn(5, 7).astype(np.float32)

    def ref(x, y):
        return np.sum(x * y, 1) / (np.linalg.norm(x, axis=1) *
                                   np.linalg.norm(y, axis=1))
    check(dev, autograd.cossim, ref, (a, b))


def _conv_layer_vs_torch(dev, x, nb, k, stride, pad_mode, one_d=False):
    autograd.training = True
    tx = to_t(x, dev)
    conv = layer.Conv2d(nb, k, stride=stride, pad_mode=pad_mode, bias=True)
    y = conv(tx)
    W = tensor.to_numpy(conv.W)
    b = tensor.to_numpy(conv.b)
    n = x.shape[2:]
    ks = conv.kernel_size
    st = conv.stride
    pads = []
    for size, kk, s in zip(n, ks, st):
        total = max((-(-size // s) - 1) * s + kk - size, 0)
        small, big = total // 2, total - total // 2
        pads.append((small, big) if pad_mode == "SAME_UPPER" else (big, small))
    xt = torch.from_numpy(x).requires_grad_(True)
    xp = F.pad(xt, (pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    wt = torch.from_numpy(W).requires_grad_(True)
    yt = F.conv2d(xp, wt, torch.from_numpy(b), st)
    np.testing.assert_allclose(tensor.to_numpy(y), yt.detach().numpy(),
                               rtol=1e-4, atol=1e-4)
    w = RS.uniform(0.5, 1.5, yt.shape).astype(np.float32)
    (yt * torch.from_numpy(w)).sum().backward()
    grads = autograd.gradients(scalar_loss(y, to_t(w, dev, False)))
    np.testing.assert_allclose(tensor.to_numpy(grads[id(tx)]),
                               xt.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(tensor.to_numpy(grads[id(conv.W)]),
                               wt.grad.numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("pad_mode", ["SAME_UPPER", "SAME_LOWER"])
@pytest.mark.parametrize("k,stride,hw", [(3, 1, 7), (3, 2, 8), (4, 3, 9),
                                         (2, 2, 5)])
def test_conv2d_same_pad(dev, pad_mode, k, stride, hw):
    x = RS.randn(2, 3, hw, hw).astype(np.float32)
    _conv_layer_vs_torch(dev, x, 4, k, stride, pad_mode)


@pytest.mark.parametrize("pad_mode", ["SAME_UPPER", "SAME_LOWER"])
def test_conv1d_same_pad(dev, pad_mode):
    # a 1-d convolution is the (1, k) window over a (N, C, 1, L) tensor
    x = RS.randn(2, 3, 1, 9).astype(np.float32)
    _conv_layer_vs_torch(dev, x, 4, (1, 4), (1, 2), pad_mode)


def _pool_same(dev, x, k, stride, pad_mode, is_max, cls):
    autograd.training = True
    tx = to_t(x, dev)
    pool = cls(k, stride, pad_mode=pad_mode)
    y = pool(tx)
    four = x if x.ndim == 4 else x[:, :, None, :]
    ks = pool.kernel_size
    st = pool.stride
    pads = []
    for size, kk, s in zip(four.shape[2:], ks, st):
        total = max((-(-size // s) - 1) * s + kk - size, 0)
        small, big = total // 2, total - total // 2
        pads.append((small, big) if pad_mode == "SAME_UPPER" else (big, small))
    xt = torch.from_numpy(four)
    if is_max:
        xp = F.pad(xt, (pads[1][0], pads[1][1], pads[0][0], pads[0][1]),
                   value=float("-inf"))
        want = F.max_pool2d(xp, ks, st).numpy()
    else:
        # the odd cell is a replicated border cell (reference utils.py
        # handle_odd_pad_fwd), the symmetric part is excluded from the mean
        want = None
    got = tensor.to_numpy(y)
    out_sizes = [-(-size // s) for size, s in zip(four.shape[2:], st)]
    assert list(got.shape[2:]) == out_sizes[-(x.ndim - 2):]
    if want is not None:
        np.testing.assert_allclose(got.reshape(want.shape), want, rtol=1e-5)
    grads = autograd.gradients(scalar_loss(
        y, to_t(np.ones(got.shape, np.float32), dev, False)))
    g = tensor.to_numpy(grads[id(tx)])
    assert g.shape == x.shape and np.isfinite(g).all()
    # every output cell hands out a total gradient of one
    np.testing.assert_allclose(g.sum(), got.size, rtol=1e-4)


@pytest.mark.parametrize("pad_mode", ["SAME_UPPER", "SAME_LOWER"])
@pytest.mark.parametrize("is_max", [True, False])
def test_pooling2d_same_pad(dev, pad_mode, is_max):
    x = RS.randn(2, 3, 8, 7).astype(np.float32)
    cls = layer.MaxPool2d if is_max else layer.AvgPool2d
    _pool_same(dev, x, 3, 2, pad_mode, is_max, cls)


@pytest.mark.parametrize("pad_mode", ["SAME_UPPER", "SAME_LOWER"])
@pytest.mark.parametrize("is_max", [True, False])
def test_pooling1d_same_pad(dev, pad_mode, is_max):
    x = RS.randn(2, 3, 9).astype(np.float32)
    cls = layer.MaxPool1d if is_max else layer.AvgPool1d
    _pool_same(dev, x, 4, 2, pad_mode, is_max, cls)


def test_separable_conv2d(dev):
    autograd.training = True
    x = RS.randn(2, 8, 6, 6).astype(np.float32)
    tx = to_t(x, dev)
    sep = layer.SeparableConv2d(16, 3, padding=1)
    y = sep(tx)
    assert y.shape == (2, 16, 6, 6)
    dw = tensor.to_numpy(sep.depthwise_conv.W)
    pw = tensor.to_numpy(sep.point_conv.W)
    want = F.conv2d(F.conv2d(torch.from_numpy(x), torch.from_numpy(dw), None,
                             1, 1, 1, 8), torch.from_numpy(pw)).numpy()
    np.testing.assert_allclose(tensor.to_numpy(y), want, rtol=1e-4, atol=1e-4)
    grads = autograd.gradients(scalar_loss(
        y, to_t(np.ones(y.shape, np.float32), dev, False)))
    assert id(sep.depthwise_conv.W) in grads and id(sep.point_conv.W) in grads


@pytest.mark.parametrize("mode", ["DCR", "CRD"])
def test_depth_space(dev, mode):
    x = RS.randn(2, 8, 3, 4).astype(np.float32)

    def d2s(a):
        n, c, h, w = a.shape
        if mode == "DCR":
            t = a.reshape(n, 2, 2, c // 4, h, w).transpose(0, 3, 4, 1, 5, 2)
        else:
            t = a.reshape(n, c // 4, 2, 2, h, w).transpose(0, 1, 4, 2, 5, 3)
        return t.reshape(n, c // 4, h * 2, w * 2)
    check(dev, lambda t: autograd.depth_to_space(t, 2, mode), d2s, (x,))
    y = d2s(x).astype(np.float32)
    # space_to_depth is its inverse
    autograd.training = False
    back = autograd.space_to_depth(to_t(y, dev, False), 2, mode)
    np.testing.assert_allclose(tensor.to_numpy(back), x, rtol=1e-6)
    autograd.training = True


@pytest.mark.parametrize("axis", [0, 1, -1])
def test_scatter_elements(dev, axis):
    x = RS.randn(4, 5).astype(np.float32)
    idx = np.stack([RS.permutation(4) for _ in range(5)], 1)[:3] \
        if axis == 0 else np.stack([RS.permutation(5)[:3] for _ in range(4)])
    upd = RS.randn(*idx.shape).astype(np.float32)
    want = x.copy()
    np.put_along_axis(want, idx, upd, axis)
    autograd.training = False
    y = autograd.scatter_elements(to_t(x, dev, False), idx, upd, axis)
    autograd.training = True
    np.testing.assert_allclose(tensor.to_numpy(y), want, rtol=1e-6)


def test_scatter_elements_rank3_repeats_and_gradient(dev):
    """Any rank, negative indices, a cell written twice (the later update
    stays), updates on the device; the input's gradient is zero exactly where
    a value was replaced."""
    x = RS.randn(3, 4, 5).astype(np.float32)
    idx = RS.randint(-4, 4, (2, 4, 3))
    idx[1, 2, 1] = idx[0, 2, 1]                 # no clash: different rows
    idx[0, 1, :] = [2, -2, 2]                   # axis 1: cells written twice
    upd = RS.randn(2, 4, 3).astype(np.float32)
    want, touched = x.copy(), np.zeros(x.shape, bool)
    for i in range(2):
        for j in range(4):
            for k in range(3):
                want[i, idx[i, j, k] % 4, k] = upd[i, j, k]
                touched[i, idx[i, j, k] % 4, k] = True
    tx = to_t(x, dev)
    y = autograd.scatter_elements(tx, idx, to_t(upd, dev, False), 1)
    np.testing.assert_allclose(tensor.to_numpy(y), want, rtol=1e-6)
    w = to_t(RS.uniform(0.5, 1.5, x.shape).astype(np.float32), dev, False)
    grads = {id(p): g for p, g in autograd.backward(scalar_loss(y, w))}
    g = tensor.to_numpy(grads[id(tx)])
    assert (g[touched] == 0).all() and (g[~touched] != 0).all()


def test_nonzero(dev):
    x = np.array([[1, 0, 2], [0, 0, 3]], np.float32)
    autograd.training = False
    y = autograd.nonzero(to_t(x, dev, False))
    autograd.training = True
    np.testing.assert_array_equal(tensor.to_numpy(y),
                                  np.array(np.nonzero(x)))


def test_invalid_inputs(dev):
    """Shape errors surface as exceptions, not as silent wrong results
    (reference test_operation.py: test_invalid_inputs)."""
    a = to_t(RS.randn(3, 4).astype(np.float32), dev)
    b = to_t(RS.randn(5, 6).astype(np.float32), dev)
    with pytest.raises(Exception):
        autograd.matmul(a, b)
    with pytest.raises(Exception):
        autograd.reshape(a, (5, 5))
    with pytest.raises(Exception):
        autograd.cat([a, b], 0)
    with pytest.raises(Exception):
        autograd.Clip(None, None)(3.0)          # not a Tensor


def test_rnn_lstm_layers_tiny_ops_shape_check(dev):
    """Unrolled python RNN / LSTM layers on tiny inputs: output shapes and a
    gradient for every parameter (reference: test_vanillaRNN_gpu_tiny_ops_
    shape_check, test_LSTM_gpu_tiny_ops_shape_check)."""
    autograd.training = True
    xs = [to_t(RS.randn(2, 3).astype(np.float32), dev, False)
          for _ in range(3)]
    h0 = to_t(np.zeros((2, 2), np.float32), dev, False)
    c0 = to_t(np.zeros((2, 2), np.float32), dev, False)
    for net, state in ((layer.RNN(3, 2), (h0,)),
                       (layer.LSTM(3, 2), ((h0, c0),))):
        out = net(xs, *state)
        hs = out[0]
        assert len(hs) == 3 and all(h.shape == (2, 2) for h in hs)
        loss = autograd.softmax_cross_entropy(
            autograd.cat(hs, 0),
            to_t(np.array([0, 1, 0, 1, 0, 1], np.int32), dev, False))
        got = {id(p) for p, _ in autograd.backward(loss)}
        for p in net.get_params().values():
            assert id(p) in got


def _layer_loss(net, xs, state, target):
    out = net(xs, *state)
    return autograd.softmax_cross_entropy(autograd.cat(out[0], 0), target)


@pytest.mark.parametrize("kind", ["rnn", "lstm"])
def test_numerical_gradients_check_for_rnn_layers(dev, kind):
    """Central differences of the loss w.r.t. single parameter entries
    (reference: test_numerical_gradients_check_for_vallina_rnn / _lstm)."""
    autograd.training = True
    xs = [to_t(RS.randn(2, 3).astype(np.float32), dev, False)
          for _ in range(3)]
    h0 = to_t(RS.randn(2, 2).astype(np.float32) * 0.1, dev, False)
    c0 = to_t(RS.randn(2, 2).astype(np.float32) * 0.1, dev, False)
    target = to_t(np.array([0, 1, 1, 0, 1, 0], np.int32), dev, False)
    net = layer.RNN(3, 2) if kind == "rnn" else layer.LSTM(3, 2)
    state = (h0,) if kind == "rnn" else ((h0, c0),)
    net(xs, *state)                                   # creates the parameters
    for p in net.get_params().values():
        p.copy_from_numpy((RS.randn(*p.shape) * 0.5).astype(np.float32))
    loss = _layer_loss(net, xs, state, target)
    grads = {id(p): tensor.to_numpy(g) for p, g in autograd.backward(loss)}
    eps = 1e-2
    for p in net.get_params().values():
        value = tensor.to_numpy(p).copy()
        idx = tuple(RS.randint(0, n) for n in value.shape)
        ends = []
        for sign in (1, -1):
            bumped = value.copy()
            bumped[idx] += sign * eps
            p.copy_from_numpy(bumped)
            ends.append(float(tensor.to_numpy(
                _layer_loss(net, xs, state, target)).reshape(-1)[0]))
        p.copy_from_numpy(value)
        np.testing.assert_allclose(grads[id(p)][idx],
                                   (ends[0] - ends[1]) / (2 * eps),
                                   rtol=5e-2, atol=2e-3)


@pytest.mark.parametrize("mode", ["vanilla", "lstm", "gru"])
def test_cudnn_rnn_operation_and_gradient_check(dev, mode):
    """``layer.CudnnRNN`` forward shapes and a central-difference check of its
    flat weight gradient (reference: test_cudnn_rnn_operation,
    test_gradient_check_cudnn_rnn_vanilla / _lstm)."""
    autograd.training = True
    x = RS.randn(2, 3, 4).astype(np.float32)             # (batch, seq, feature)
    tx = to_t(x, dev, False)
    rnn = layer.CudnnRNN(3, rnn_mode=mode, batch_first=True,
                         return_sequences=True)
    y = rnn(tx)
    assert y.shape == (2, 3, 3)
    rnn.W.copy_from_numpy(
        (RS.randn(rnn.W.size()) * 0.4).astype(np.float32))
    w = RS.uniform(0.5, 1.5, (2, 3, 3)).astype(np.float32)

    def loss_value():
        return float(tensor.to_numpy(scalar_loss(
            rnn(tx), to_t(w, dev, False))).reshape(-1)[0])

    loss = scalar_loss(rnn(tx), to_t(w, dev, False))
    g = {id(p): tensor.to_numpy(v) for p, v in autograd.backward(loss)}[
        id(rnn.W)]
    value = tensor.to_numpy(rnn.W).copy()
    eps = 1e-2
    for idx in RS.choice(value.size, 6, replace=False):
        ends = []
        for sign in (1, -1):
            bumped = value.copy()
            bumped[idx] += sign * eps
            rnn.W.copy_from_numpy(bumped)
            ends.append(loss_value())
        rnn.W.copy_from_numpy(value)
        np.testing.assert_allclose(g[idx], (ends[0] - ends[1]) / (2 * eps),
                                   rtol=5e-2, atol=2e-3)
