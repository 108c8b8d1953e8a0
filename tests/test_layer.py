"""Layers and the native-handle NN operators (convolution, batch norm, pooling,
recurrent layers) against plain PyTorch float32 references of the same maths.
Cases follow reference test/python/test_layer.py, test_operation.py (conv /
bn / pool parts) and test/singa/test_operation_*.cc.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from singa_amd import autograd, layer, tensor

RS = np.random.RandomState(11)


def T(a, dev, grad=False):
    t = tensor.Tensor(device=dev, data=np.asarray(a, dtype=np.float32))
    t.requires_grad = grad
    t.stores_grad = grad
    return t


def loss_of(y, dev):
    w = RS.uniform(0.5, 1.5, y.shape).astype(np.float32)
    l = autograd.reduce_sum(autograd.mul(y, T(w, dev)), None, 0)
    return autograd.reshape(l, (1,)), torch.from_numpy(w)


def set_param(p, a):
    p.copy_from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def test_linear(dev):
    autograd.training = True
    x = RS.randn(5, 7).astype(np.float32)
    lin = layer.Linear(4)
    tx = T(x, dev, True)
    y = lin(tx)
    W, b = tensor.to_numpy(lin.W), tensor.to_numpy(lin.b)
    assert W.shape == (7, 4) and b.shape == (4,)
    np.testing.assert_allclose(tensor.to_numpy(y), x @ W + b, rtol=1e-5,
                               atol=1e-5)
    l, w = loss_of(y, dev)
    g = autograd.gradients(l)
    np.testing.assert_allclose(tensor.to_numpy(g[id(lin.W)]), x.T @ w.numpy(),
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(g[id(lin.b)]), w.numpy().sum(0),
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(g[id(tx)]), w.numpy() @ W.T,
                               rtol=1e-4, atol=1e-5)
    assert set(lin.get_params().keys()) == {lin.W.name, lin.b.name}


def test_linear_no_bias_and_3d_input(dev):
    x = RS.randn(2, 3, 6).astype(np.float32)
    lin = layer.Linear(5, bias=False)
    y = lin(T(x, dev))
    W = tensor.to_numpy(lin.W)
    np.testing.assert_allclose(tensor.to_numpy(y), x @ W, rtol=1e-5, atol=1e-5)
    assert len(lin.get_params()) == 1


CONVS = [
    # cin, cout, k, stride, pad, group, bias, hw
    (3, 8, 3, 1, 1, 1, True, 9),
    (4, 6, 3, 2, 1, 1, False, 10),
    (4, 8, 1, 1, 0, 1, True, 6),
    (4, 8, 3, 1, 0, 2, True, 7),
    (8, 8, 3, 1, 1, 8, False, 6),
    (3, 4, 5, 2, 2, 1, True, 11),
]


@pytest.mark.parametrize("cin,cout,k,st,pd,grp,bias,hw", CONVS)
def test_conv2d(dev, cin, cout, k, st, pd, grp, bias, hw):
    autograd.training = True
    x = RS.randn(2, cin, hw, hw).astype(np.float32)
    conv = layer.Conv2d(cout, k, stride=st, padding=pd, group=grp, bias=bias)
    tx = T(x, dev, True)
    y = conv(tx)
    W = tensor.to_numpy(conv.W)
    assert W.shape == (cout, cin // grp, k, k)
    b = tensor.to_numpy(conv.b) if bias else None
    xt = torch.from_numpy(x).requires_grad_(True)
    Wt = torch.from_numpy(W).requires_grad_(True)
    bt = torch.from_numpy(b).requires_grad_(True) if bias else None
    yt = F.conv2d(xt, Wt, bt, st, pd, 1, grp)
    np.testing.assert_allclose(tensor.to_numpy(y), yt.detach().numpy(),
                               rtol=1e-4, atol=1e-4)
    l, w = loss_of(y, dev)
    g = autograd.gradients(l)
    (yt * w).sum().backward()
    np.testing.assert_allclose(tensor.to_numpy(g[id(conv.W)]), Wt.grad.numpy(),
                               rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(tensor.to_numpy(g[id(tx)]), xt.grad.numpy(),
                               rtol=1e-3, atol=1e-3)
    if bias:
        np.testing.assert_allclose(tensor.to_numpy(g[id(conv.b)]),
                                   bt.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_conv2d_same_padding(dev):
    x = RS.randn(1, 2, 7, 7).astype(np.float32)
    for mode in ("SAME_UPPER", "SAME_LOWER"):
        conv = layer.Conv2d(3, 2, stride=1, pad_mode=mode, bias=False)
        y = conv(T(x, dev))
        assert y.shape == (1, 3, 7, 7)
        W = torch.from_numpy(tensor.to_numpy(conv.W))
        pad = (0, 1, 0, 1) if mode == "SAME_UPPER" else (1, 0, 1, 0)
        ref = F.conv2d(F.pad(torch.from_numpy(x), pad), W)
        np.testing.assert_allclose(tensor.to_numpy(y), ref.numpy(), rtol=1e-4,
                                   atol=1e-4)


def test_separable_conv(dev):
    x = RS.randn(2, 4, 8, 8).astype(np.float32)
    sep = layer.SeparableConv2d(6, 3, padding=1)
    y = sep(T(x, dev))
    assert y.shape == (2, 6, 8, 8)
    dw = torch.from_numpy(tensor.to_numpy(sep.depthwise_conv.W))
    pw = torch.from_numpy(tensor.to_numpy(sep.point_conv.W))
    ref = F.conv2d(F.conv2d(torch.from_numpy(x), dw, None, 1, 1, 1, 4), pw)
    np.testing.assert_allclose(tensor.to_numpy(y), ref.numpy(), rtol=1e-4,
                               atol=1e-4)


def test_batchnorm2d(dev):
    autograd.training = True
    x = (RS.randn(4, 3, 5, 5) * 2 + 1).astype(np.float32)
    bn = layer.BatchNorm2d(momentum=0.9)
    tx = T(x, dev, True)
    y = bn(tx)
    sc = RS.uniform(0.5, 1.5, 3).astype(np.float32)
    bi = RS.uniform(-1, 1, 3).astype(np.float32)
    set_param(bn.scale, sc)
    set_param(bn.bias, bi)
    y = bn(tx)
    xt = torch.from_numpy(x).requires_grad_(True)
    st = torch.from_numpy(sc).requires_grad_(True)
    bt = torch.from_numpy(bi).requires_grad_(True)
    yt = F.batch_norm(xt, None, None, st, bt, True, 0.1, 1e-5)
    np.testing.assert_allclose(tensor.to_numpy(y), yt.detach().numpy(),
                               rtol=1e-4, atol=1e-4)
    l, w = loss_of(y, dev)
    g = autograd.gradients(l)
    (yt * w).sum().backward()
    np.testing.assert_allclose(tensor.to_numpy(g[id(tx)]), xt.grad.numpy(),
                               rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(tensor.to_numpy(g[id(bn.scale)]),
                               st.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(tensor.to_numpy(g[id(bn.bias)]),
                               bt.grad.numpy(), rtol=1e-3, atol=1e-3)
    # running statistics move towards the batch statistics
    rm = tensor.to_numpy(bn.running_mean)
    assert np.all(np.abs(rm) > 0)
    autograd.training = False
    yi = bn(T(x, dev))
    rv = tensor.to_numpy(bn.running_var)
    ref = (x - rm[None, :, None, None]) / np.sqrt(
        rv[None, :, None, None] + 1e-5) * sc[None, :, None, None] + \
        bi[None, :, None, None]
    np.testing.assert_allclose(tensor.to_numpy(yi), ref, rtol=1e-4, atol=1e-4)
    autograd.training = True
    assert len(bn.get_params()) == 2 and len(bn.get_states()) == 4


POOLS = [("max", 2, 2, 0, 8), ("max", 3, 2, 1, 9), ("avg", 2, 2, 0, 8),
         ("avg", 3, 1, 1, 7)]


@pytest.mark.parametrize("kind,k,st,pd,hw", POOLS)
def test_pooling(dev, kind, k, st, pd, hw):
    autograd.training = True
    x = RS.randn(2, 3, hw, hw).astype(np.float32)
    cls = layer.MaxPool2d if kind == "max" else layer.AvgPool2d
    pool = cls(k, st, pd)
    tx = T(x, dev, True)
    y = pool(tx)
    xt = torch.from_numpy(x).requires_grad_(True)
    if kind == "max":
        yt = F.max_pool2d(xt, k, st, pd)
    else:
        yt = F.avg_pool2d(xt, k, st, pd, count_include_pad=False)
    np.testing.assert_allclose(tensor.to_numpy(y), yt.detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    l, w = loss_of(y, dev)
    g = autograd.gradients(l)
    (yt * w).sum().backward()
    np.testing.assert_allclose(tensor.to_numpy(g[id(tx)]), xt.grad.numpy(),
                               rtol=1e-4, atol=1e-4)


def test_pool1d(dev):
    x = RS.randn(2, 3, 10).astype(np.float32)
    y = layer.MaxPool1d(2, 2)(T(x, dev))
    ref = F.max_pool1d(torch.from_numpy(x), 2, 2)
    np.testing.assert_allclose(tensor.to_numpy(y).reshape(ref.shape),
                               ref.numpy(), rtol=1e-5)
    y = layer.AvgPool1d(2, 2)(T(x, dev))
    ref = F.avg_pool1d(torch.from_numpy(x), 2, 2)
    np.testing.assert_allclose(tensor.to_numpy(y).reshape(ref.shape),
                               ref.numpy(), rtol=1e-5)


def test_activation_and_utility_layers(dev):
    x = RS.randn(3, 4).astype(np.float32)
    np.testing.assert_allclose(tensor.to_numpy(layer.ReLU()(T(x, dev))),
                               np.maximum(x, 0))
    np.testing.assert_allclose(tensor.to_numpy(layer.Sigmoid()(T(x, dev))),
                               1 / (1 + np.exp(-x)), rtol=1e-5)
    np.testing.assert_allclose(
        tensor.to_numpy(layer.Add()(T(x, dev), T(x, dev))), 2 * x)
    x4 = RS.randn(2, 3, 4, 5).astype(np.float32)
    assert layer.Flatten()(T(x4, dev)).shape == (2, 60)
    assert layer.Reshape()(T(x4, dev), (6, 20)).shape == (6, 20)
    c = layer.Cat(1)([T(x, dev), T(x, dev)])
    np.testing.assert_allclose(tensor.to_numpy(c), np.concatenate([x, x], 1))
    e = np.exp(x - x.max(1, keepdims=True))
    np.testing.assert_allclose(tensor.to_numpy(layer.SoftMax()(T(x, dev))),
                               e / e.sum(1, keepdims=True), rtol=1e-5)
    lab = tensor.Tensor(device=dev, data=np.array([0, 3, 1], np.int32))
    l = layer.SoftMaxCrossEntropy()(T(x, dev), lab)
    ref = F.cross_entropy(torch.from_numpy(x), torch.tensor([0, 3, 1]))
    np.testing.assert_allclose(tensor.to_numpy(l).reshape(()), ref.numpy(),
                               rtol=1e-5)
    t = RS.randn(3, 4).astype(np.float32)
    l = layer.MeanSquareError()(T(x, dev), T(t, dev))
    assert tensor.to_numpy(l).size == 1


def test_embedding_layer(dev):
    emb = layer.Embedding(10, 4)
    ids = tensor.Tensor(device=dev, data=np.array([[1, 2], [9, 0]], np.int32))
    y = emb(ids)
    W = tensor.to_numpy(emb.W)
    np.testing.assert_allclose(tensor.to_numpy(y), W[[[1, 2], [9, 0]]],
                               rtol=1e-6)


def test_gemm_layer(dev):
    x = RS.randn(3, 5).astype(np.float32)
    gm = layer.Gemm(4, alpha=0.5, beta=2.0, transA=False, transB=True)
    y = gm(T(x, dev))
    W, b = tensor.to_numpy(gm.W), tensor.to_numpy(gm.b)
    np.testing.assert_allclose(tensor.to_numpy(y), 0.5 * x @ W.T + 2.0 * b,
                               rtol=1e-4, atol=1e-5)


def _torch_lstm_ref(x, h0, c0, Wx, Wh, b):
    # gate order of singa_amd.layer.LSTM: i, f, o, g
    T_, B, _ = x.shape
    H = h0.shape[1]
    h, c = torch.from_numpy(h0), torch.from_numpy(c0)
    outs = []
    for t in range(T_):
        z = torch.from_numpy(x[t]) @ torch.from_numpy(Wx) + h @ \
            torch.from_numpy(Wh) + torch.from_numpy(b)
        i, f, o, g = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs), h, c


def test_lstm_layer(dev):
    autograd.training = True
    T_, B, D, H = 4, 3, 5, 6
    x = RS.randn(T_, B, D).astype(np.float32)
    lstm = layer.LSTM(D, H)
    xs = [T(x[t], dev) for t in range(T_)]
    h0 = T(np.zeros((B, H)), dev)
    c0 = T(np.zeros((B, H)), dev)
    hs, hT, cT = lstm(xs, (h0, c0))
    assert len(hs) == T_ and hs[0].shape == (B, H)
    params = lstm.get_params()
    assert len(params) > 0
    # gradients flow to every parameter
    loss = autograd.reshape(autograd.reduce_sum(hs[-1], None, 0), (1,))
    got = {id(p) for p, g in autograd.backward(loss)}
    assert got == {id(p) for p in params.values()}


def test_rnn_layer(dev):
    autograd.training = True
    T_, B, D, H = 3, 2, 4, 5
    x = RS.randn(T_, B, D).astype(np.float32)
    rnn = layer.RNN(D, H)
    xs = [T(x[t], dev) for t in range(T_)]
    h0 = T(np.zeros((B, H)), dev)
    hs, hT = rnn(xs, h0)
    Wx, Wh, b = [tensor.to_numpy(p) for p in (rnn.Wx, rnn.Wh, rnn.b)]
    h = np.zeros((B, H), np.float32)
    for t in range(T_):
        h = np.tanh(x[t] @ Wx + h @ Wh + b)
    np.testing.assert_allclose(tensor.to_numpy(hs[-1]), h, rtol=1e-4,
                               atol=1e-5)


def test_dropout_layer(dev):
    autograd.training = True
    y = layer.Dropout(0.5)(T(np.ones((50, 50)), dev))
    v = tensor.to_numpy(y)
    assert 0.35 < (v == 0).mean() < 0.65
    autograd.training = False
    y = layer.Dropout(0.5)(T(np.ones((5, 5)), dev))
    assert np.all(tensor.to_numpy(y) == 1)
    autograd.training = True


def test_layer_naming_and_states(dev):
    class Block(layer.Layer):
        def __init__(self):
            super().__init__()
            self.fc1 = layer.Linear(4)
            self.fc2 = layer.Linear(2)

        def forward(self, x):
            return self.fc2(self.fc1(x))

    blk = Block()
    blk(T(RS.randn(3, 5), dev))
    names = sorted(blk.get_params().keys())
    assert len(names) == 4
    assert all(n.count(layer.Layer.sep) >= 1 for n in names)
    states = blk.get_states()
    new = {k: tensor.to_numpy(v) * 0 + 1 for k, v in states.items()}
    blk.set_states({k: tensor.from_numpy(v) for k, v in new.items()})
    for v in blk.get_params().values():
        assert np.all(tensor.to_numpy(v) == 1)
