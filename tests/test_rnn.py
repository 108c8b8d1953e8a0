"""Flat-weight recurrent stacks (``CudnnRNNHandle`` + RNNForward / Backward,
``layer.CudnnRNN``) against torch.nn.RNN / LSTM / GRU float32 on the host, with
the weights copied gate by gate through ``weights_mapping`` (reference
test/python/test_operation.py: test_rnn_*, test/singa/test_operation_rnn.cc).
"""
import numpy as np
import pytest
import torch

from singa_amd import autograd, layer, tensor
from singa_amd import singa_wrap as singa

RS = np.random.RandomState(5)
MODES = {"relu": 0, "tanh": 1, "lstm": 2, "gru": 3}


def _torch_net(kind, n_in, H, layers, bidir):
    if kind == "lstm":
        return torch.nn.LSTM(n_in, H, layers, bidirectional=bidir)
    if kind == "gru":
        return torch.nn.GRU(n_in, H, layers, bidirectional=bidir)
    return torch.nn.RNN(n_in, H, layers, nonlinearity=kind,
                        bidirectional=bidir)


def _fill_weights(handle, net, W_np):
    G, D, H = handle.gates, handle.directions, handle.hidden_size
    for l in range(handle.num_layers):
        for d in range(D):
            sfx = "_l%d%s" % (l, "_reverse" if d else "")
            p = l * D + d
            parts = [("weight_ih", 0, False), ("weight_hh", G, False),
                     ("bias_ih", 0, True), ("bias_hh", G, True)]
            for name, base, is_bias in parts:
                w = getattr(net, name + sfx).detach().numpy()
                for g in range(G):
                    off, size = handle.weights_mapping[(base + g, p, is_bias)]
                    W_np[off:off + size] = w[g * H:(g + 1) * H].reshape(-1)


def _torch_grads(handle, net):
    G, D, H = handle.gates, handle.directions, handle.hidden_size
    out = np.zeros(handle.weights_size, np.float32)
    for l in range(handle.num_layers):
        for d in range(D):
            sfx = "_l%d%s" % (l, "_reverse" if d else "")
            p = l * D + d
            for name, base, is_bias in [("weight_ih", 0, False),
                                        ("weight_hh", G, False),
                                        ("bias_ih", 0, True),
                                        ("bias_hh", G, True)]:
                w = getattr(net, name + sfx).grad.numpy()
                for g in range(G):
                    off, size = handle.weights_mapping[(base + g, p, is_bias)]
                    out[off:off + size] = w[g * H:(g + 1) * H].reshape(-1)
    return out


CASES = [("lstm", 1, False), ("lstm", 2, True), ("gru", 1, False),
         ("gru", 2, True), ("tanh", 2, False), ("relu", 1, True)]


@pytest.mark.parametrize("kind,layers,bidir", CASES)
def test_rnn_forward_backward(dev, kind, layers, bidir):
    T, B, n_in, H = 5, 3, 4, 6
    D = 2 if bidir else 1
    x = RS.randn(T, B, n_in).astype(np.float32)
    hx = RS.randn(layers * D, B, H).astype(np.float32) * 0.5
    cx = RS.randn(layers * D, B, H).astype(np.float32) * 0.5
    tx = tensor.Tensor(device=dev, data=x)
    handle = singa.CudnnRNNHandle(tx.data, H, MODES[kind], layers, 1, 0.0,
                                  1 if bidir else 0)
    torch.manual_seed(0)
    net = _torch_net(kind, n_in, H, layers, bidir)
    W_np = np.zeros(handle.weights_size, np.float32)
    _fill_weights(handle, net, W_np)
    assert handle.weights_size == sum(p.numel() for p in net.parameters())
    tW = tensor.Tensor(device=dev, data=W_np)
    thx = tensor.Tensor(device=dev, data=hx)
    tcx = tensor.Tensor(device=dev, data=cx)
    y, hy, cy = singa.RNNForwardTraining(tx.data, thx.data, tcx.data, tW.data,
                                         handle)
    xt = torch.from_numpy(x).requires_grad_(True)
    h0 = torch.from_numpy(hx).requires_grad_(True)
    c0 = torch.from_numpy(cx).requires_grad_(True)
    if kind == "lstm":
        yt, (hn, cn) = net(xt, (h0, c0))
    else:
        yt, hn = net(xt, h0)
    np.testing.assert_allclose(tensor.to_numpy(tensor.from_raw_tensor(y)),
                               yt.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tensor.to_numpy(tensor.from_raw_tensor(hy)),
                               hn.detach().numpy(), rtol=1e-4, atol=1e-5)
    if kind == "lstm":
        np.testing.assert_allclose(tensor.to_numpy(tensor.from_raw_tensor(cy)),
                                   cn.detach().numpy(), rtol=1e-4, atol=1e-5)
    # backward with random upstream gradients on y, hy (and cy)
    dy = RS.randn(*yt.shape).astype(np.float32)
    dhy = RS.randn(*hn.shape).astype(np.float32)
    dcy = RS.randn(*hn.shape).astype(np.float32)
    loss = (yt * torch.from_numpy(dy)).sum() + (hn * torch.from_numpy(dhy)).sum()
    if kind == "lstm":
        loss = loss + (cn * torch.from_numpy(dcy)).sum()
    loss.backward()
    T_ = lambda a: tensor.Tensor(device=dev, data=a).data
    dx, dhx, dcx, dW = singa.RNNBackward(y, T_(dy), T_(dhy), T_(dcy), tW.data,
                                         thx.data, tcx.data, tx.data, handle)
    n = lambda c: tensor.to_numpy(tensor.from_raw_tensor(c))
    np.testing.assert_allclose(n(dx), xt.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(n(dhx), h0.grad.numpy(), rtol=1e-3, atol=1e-4)
    if kind == "lstm":
        np.testing.assert_allclose(n(dcx), c0.grad.numpy(), rtol=1e-3,
                                   atol=1e-4)
    np.testing.assert_allclose(n(dW), _torch_grads(handle, net), rtol=1e-3,
                               atol=1e-4)


@pytest.mark.parametrize("kind", ["lstm", "gru"])
def test_rnn_variable_lengths(dev, kind):
    """Packed-sequence semantics: outputs past a sample's length are zero and
    the final state is the one at its last valid step."""
    T, B, n_in, H = 6, 3, 4, 5
    lens = np.array([6, 3, 1], np.int32)
    x = RS.randn(T, B, n_in).astype(np.float32)
    tx = tensor.Tensor(device=dev, data=x)
    handle = singa.CudnnRNNHandle(tx.data, H, MODES[kind], 1, 1, 0.0, 1)
    torch.manual_seed(1)
    net = _torch_net(kind, n_in, H, 1, True)
    W_np = np.zeros(handle.weights_size, np.float32)
    _fill_weights(handle, net, W_np)
    tW = tensor.Tensor(device=dev, data=W_np)
    z = np.zeros((2, B, H), np.float32)
    thx = tensor.Tensor(device=dev, data=z)
    tcx = tensor.Tensor(device=dev, data=z)
    tl = tensor.Tensor(device=dev, data=lens)
    y, hy, cy = singa.RNNForwardTraining(tx.data, thx.data, tcx.data, tW.data,
                                         handle, tl.data)
    xt = torch.from_numpy(x).requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(
        xt, torch.from_numpy(lens.astype(np.int64)), enforce_sorted=True)
    out, hn = net(packed)
    if kind == "lstm":
        hn = hn[0]
    yt, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=T)
    n = lambda c: tensor.to_numpy(tensor.from_raw_tensor(c))
    np.testing.assert_allclose(n(y), yt.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(n(hy), hn.detach().numpy(), rtol=1e-4,
                               atol=1e-5)
    dy = RS.randn(T, B, 2 * H).astype(np.float32)
    (yt * torch.from_numpy(dy)).sum().backward()
    T_ = lambda a: tensor.Tensor(device=dev, data=a).data
    dx, dhx, dcx, dW = singa.RNNBackward(y, T_(dy), T_(z), T_(z), tW.data,
                                         thx.data, tcx.data, tx.data, handle,
                                         tl.data)
    np.testing.assert_allclose(n(dx), xt.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(n(dW), _torch_grads(handle, net), rtol=1e-3,
                               atol=1e-4)


def test_rnn_param_accessors(dev):
    tx = tensor.Tensor((4, 2, 3), dev)
    handle = singa.CudnnRNNHandle(tx.data, 5, 2, 1, 1, 0.0, 0)
    W = tensor.Tensor((handle.weights_size,), dev)
    W.set_value(0.0)
    v = np.arange(15, dtype=np.float32)
    singa.RNNSetParam(2, 0, W.data, tensor.Tensor(device=dev, data=v).data,
                      False, handle)
    back = singa.RNNGetParamCopy(2, 0, W.data, False, handle)
    np.testing.assert_array_equal(
        tensor.to_numpy(tensor.from_raw_tensor(back)), v)
    off, size = handle.weights_mapping[(2, 0, False)]
    np.testing.assert_array_equal(tensor.to_numpy(W)[off:off + size], v)


@pytest.mark.parametrize("mode", ["lstm", "gru", "vanilla"])
@pytest.mark.parametrize("return_sequences", [True, False])
def test_cudnn_rnn_layer_trains(dev, mode, return_sequences):
    autograd.training = True
    B, T, n_in, H = 4, 5, 3, 6
    x = RS.randn(B, T, n_in).astype(np.float32)
    rnn = layer.CudnnRNN(H, rnn_mode=mode, num_layers=2, bidirectional=True,
                         batch_first=True, return_sequences=return_sequences)
    tx = tensor.Tensor(device=dev, data=x)
    y = rnn(tx)
    assert y.shape == ((B, T, 2 * H) if return_sequences else (B, 2 * H))
    loss = autograd.reshape(autograd.reduce_sum(
        autograd.mul(y, y), None, 0), (1,))
    grads = {id(p): g for p, g in autograd.backward(loss)}
    assert id(rnn.W) in grads
    g = tensor.to_numpy(grads[id(rnn.W)])
    assert g.shape == (rnn.handle.weights_size,)
    assert np.isfinite(g).all() and np.abs(g).sum() > 0
