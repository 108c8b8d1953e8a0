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
    _check_against_torch(dev, kind, layers, bidir, 5, 3, 4, 6)


# the shapes of the character-level example (batch 32, 64 steps, hidden 128),
# a hidden size / batch that needs two row tiles and masked tile edges, and the
# smallest hidden size of the single-launch LSTM kernels
LARGE = [("lstm", 1, False, 64, 32, 128, 128), ("lstm", 2, True, 64, 32, 128, 128),
         ("lstm", 1, True, 17, 40, 24, 256), ("lstm", 1, False, 9, 5, 7, 32),
         # many steps at the largest hidden size and batch of the single-launch
         # kernels (32 workgroups, four row tiles): every step crosses the grid
         # barrier, so a hand-off that lets a store trail its signal shows here
         ("lstm", 1, False, 96, 128, 64, 256),
         ("gru", 1, True, 64, 32, 128, 128), ("tanh", 1, True, 64, 32, 128, 128),
         ("relu", 1, False, 64, 32, 128, 128)]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,layers,bidir,T,B,n_in,H", LARGE)
def test_rnn_example_sized_gpu(gpu_dev, kind, layers, bidir, T, B, n_in, H):
    handle = _check_against_torch(gpu_dev, kind, layers, bidir, T, B, n_in, H,
                                  tol=4.0, scale=0.3)
    if kind == "lstm":
        # the single-launch kernels ran and no barrier wait hit its bound
        assert all(s.get("seq") for per_dir in handle.reserve["layers"]
                   for s in per_dir)
        assert handle._bar.cpu().tolist()[1:3] == [0, 0]


def _check_against_torch(dev, kind, layers, bidir, T, B, n_in, H, tol=1.0,
                         scale=1.0):
    D = 2 if bidir else 1
    x = RS.randn(T, B, n_in).astype(np.float32)
    hx = RS.randn(layers * D, B, H).astype(np.float32) * 0.5
    cx = RS.randn(layers * D, B, H).astype(np.float32) * 0.5
    tx = tensor.Tensor(device=dev, data=x)
    handle = singa.CudnnRNNHandle(tx.data, H, MODES[kind], layers, 1, 0.0,
                                  1 if bidir else 0)
    torch.manual_seed(0)
    net = _torch_net(kind, n_in, H, layers, bidir)
    if kind == "relu" and scale != 1.0:
        with torch.no_grad():       # keep 64 un-squashed steps bounded
            for p_ in net.parameters():
                p_.mul_(scale)
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
                               yt.detach().numpy(), rtol=1e-4 * tol,
                               atol=1e-5 * tol)
    np.testing.assert_allclose(tensor.to_numpy(tensor.from_raw_tensor(hy)),
                               hn.detach().numpy(), rtol=1e-4 * tol,
                               atol=1e-5 * tol)
    if kind == "lstm":
        np.testing.assert_allclose(tensor.to_numpy(tensor.from_raw_tensor(cy)),
                                   cn.detach().numpy(), rtol=1e-4 * tol,
                                   atol=1e-5 * tol)
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
    np.testing.assert_allclose(n(dx), xt.grad.numpy(), rtol=1e-3 * tol,
                               atol=1e-4 * tol)
    np.testing.assert_allclose(n(dhx), h0.grad.numpy(), rtol=1e-3 * tol,
                               atol=1e-4 * tol)
    if kind == "lstm":
        np.testing.assert_allclose(n(dcx), c0.grad.numpy(), rtol=1e-3 * tol,
                                   atol=1e-4 * tol)
    # weight gradients sum T * B products: compare against their own scale
    ref = _torch_grads(handle, net)
    np.testing.assert_allclose(n(dW), ref, rtol=1e-3 * tol,
                               atol=1e-4 * tol * max(1.0, np.abs(ref).max()))
    return handle


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


def _lstm_layer_run(dev, unrolled, T, B, n_in, H):
    autograd.training = True
    rs = np.random.RandomState(11)
    xs = [tensor.Tensor(device=dev, data=rs.randn(B, n_in).astype(np.float32))
          for _ in range(T)]
    h0 = tensor.Tensor(device=dev, data=rs.randn(B, H).astype(np.float32))
    c0 = tensor.Tensor(device=dev, data=rs.randn(B, H).astype(np.float32))
    h0.requires_grad = c0.requires_grad = True
    h0.stores_grad = c0.stores_grad = True
    lstm = layer.LSTM(n_in, H)
    lstm.initialize(xs, (h0, c0))
    for name, p in sorted(lstm.get_params().items()):
        p.copy_from_numpy(rs.randn(*p.shape).astype(np.float32) * 0.2)
    run = lstm.forward_unrolled if unrolled else lstm.forward
    out, h, c = run(xs, (h0, c0))
    w = [rs.randn(B, H).astype(np.float32) for _ in range(T + 1)]
    total = autograd.mul(c, tensor.Tensor(device=dev, data=w[T]))
    for o, wt in zip(out, w):
        total = autograd.add(total, autograd.mul(
            o, tensor.Tensor(device=dev, data=wt)))
    total = autograd.add(total, autograd.mul(h, h))   # h is out[-1] again
    loss = autograd.reshape(autograd.reduce_sum(total, None, 0), (1,))
    grads = {p.name: tensor.to_numpy(g) for p, g in autograd.backward(loss)}
    return [tensor.to_numpy(o) for o in out] + [tensor.to_numpy(c)], grads


@pytest.mark.parametrize("T,B,n_in,H", [(4, 3, 5, 6), (7, 2, 3, 32)])
def test_lstm_layer_sequence_operator_matches_unrolled(dev, T, B, n_in, H):
    """``layer.LSTM.forward`` (one operator for the whole sequence) equals the
    chain of per-step operators, outputs and every gradient."""
    y_seq, g_seq = _lstm_layer_run(dev, False, T, B, n_in, H)
    y_ref, g_ref = _lstm_layer_run(dev, True, T, B, n_in, H)
    for a, b in zip(y_seq, y_ref):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)
    assert len(g_seq) == len(g_ref) == 18          # 16 parameters, h0, c0
    for (ka, a), (kb, b) in zip(sorted(g_seq.items()), sorted(g_ref.items())):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_lstm_layer_sequence_example_sized_gpu(gpu_dev):
    y_seq, g_seq = _lstm_layer_run(gpu_dev, False, 64, 32, 101, 128)
    y_ref, g_ref = _lstm_layer_run(gpu_dev, True, 64, 32, 101, 128)
    for a, b in zip(y_seq, y_ref):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
    for (ka, a), (kb, b) in zip(sorted(g_seq.items()), sorted(g_ref.items())):
        np.testing.assert_allclose(a, b, rtol=2e-3,
                                   atol=2e-4 * max(1.0, np.abs(b).max()))
