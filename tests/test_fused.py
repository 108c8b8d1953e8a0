"""Fused operators against their unfused compositions: relu(bn(x)) and
relu(bn(x) + residual) computed inside the BatchNorm kernels must match the
separate BatchNorm / Add / ReLU operators in forward, every gradient and the
running statistics (host and MI355X)."""
import numpy as np
import pytest

from singa_amd import autograd, layer, tensor


def _grads(loss_fn, params):
    loss = loss_fn()
    got = {id(p): tensor.to_numpy(g) for p, g in autograd.backward(loss)}
    return [got[id(p)] for p in params]


@pytest.mark.parametrize("shape", [(4, 6, 5, 8), (3, 5, 7, 7), (8, 16),
                                   (8, 3, 64, 64)])
@pytest.mark.parametrize("with_residual", [False, True])
def test_bn_act_matches_unfused(dev, shape, with_residual):
    if len(shape) == 2 and with_residual:
        pytest.skip("residual form is used on feature maps")
    rs = np.random.RandomState(0)
    xn = rs.randn(*shape).astype(np.float32)
    rn = rs.randn(*shape).astype(np.float32)
    wn = rs.randn(*shape).astype(np.float32)   # weights of the scalar loss
    sn = (rs.rand(shape[1]) + 0.5).astype(np.float32)
    bn_ = (rs.randn(shape[1]) * 0.3).astype(np.float32)
    autograd.training = True
    try:
        results = []
        for fused in (True, False):
            x = tensor.Tensor(device=dev, data=xn, requires_grad=True,
                              stores_grad=True)
            r = tensor.Tensor(device=dev, data=rn, requires_grad=True,
                              stores_grad=True)
            w = tensor.Tensor(device=dev, data=wn, requires_grad=False)
            bn = layer.BatchNorm2d(shape[1], momentum=0.3,
                                   activation="RELU" if fused else None)
            y0 = bn(x, r) if (fused and with_residual) else bn(x)
            bn.scale.copy_from_numpy(sn)
            bn.bias.copy_from_numpy(bn_)
            bn.running_mean.set_value(0.0)
            bn.running_var.set_value(1.0)

            def fwd():
                if fused:
                    y = bn(x, r) if with_residual else bn(x)
                else:
                    y = bn(x)
                    if with_residual:
                        y = autograd.add(y, r)
                    y = autograd.relu(y)
                fwd.y = y
                return autograd.mse_loss(y, w)

            params = [x, bn.scale, bn.bias] + ([r] if with_residual else [])
            g = _grads(fwd, params)
            results.append((tensor.to_numpy(fwd.y), g,
                            tensor.to_numpy(bn.running_mean),
                            tensor.to_numpy(bn.running_var)))
        (yf, gf, mf, vf), (yu, gu, mu, vu) = results
        np.testing.assert_allclose(yf, yu, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(mf, mu, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(vf, vu, rtol=1e-5, atol=1e-6)
        for a, b in zip(gf, gu):
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4)
    finally:
        autograd.training = False


def test_bn_act_inference(dev):
    rs = np.random.RandomState(1)
    xn = rs.randn(2, 4, 3, 3).astype(np.float32)
    rn = rs.randn(2, 4, 3, 3).astype(np.float32)
    x = tensor.Tensor(device=dev, data=xn)
    r = tensor.Tensor(device=dev, data=rn)
    fused = layer.BatchNorm2d(4, activation="RELU")
    plain = layer.BatchNorm2d(4)
    autograd.training = False
    y1 = tensor.to_numpy(fused(x, r))
    y2 = tensor.to_numpy(autograd.relu(autograd.add(plain(x), r)))
    np.testing.assert_allclose(y1, y2, rtol=1e-5, atol=1e-6)
    fused2 = layer.BatchNorm2d(4, activation="RELU")
    np.testing.assert_allclose(tensor.to_numpy(fused2(x)),
                               np.maximum(tensor.to_numpy(plain(x)), 0),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("momentum,nesterov", [(0.0, False), (0.9, False),
                                               (0.9, True)])
def test_multi_tensor_sgd_matches_per_tensor(dev, momentum, nesterov):
    """One multi-tensor launch per 48 parameters == the per-parameter kernel,
    for sizes that straddle the 4096-element chunks and the 16-byte vector
    width, over two steps (so the momentum buffers are exercised)."""
    from singa_amd import opt
    rs = np.random.RandomState(0)
    sizes = [1, 3, 64, 4095, 4096, 4097, 10001] + [7 + i for i in range(50)]
    vals = [rs.randn(n).astype(np.float32) for n in sizes]
    grads = [[rs.randn(n).astype(np.float32) for n in sizes] for _ in range(2)]

    def run(many):
        sgd = opt.SGD(lr=0.1, momentum=momentum, weight_decay=1e-3,
                      nesterov=nesterov)
        ps = [tensor.Tensor(device=dev, data=v.copy(), name="p%d" % i)
              for i, v in enumerate(vals)]
        for step in range(2):
            gs = [tensor.Tensor(device=dev, data=g) for g in grads[step]]
            if many:
                sgd.apply_many(list(zip(ps, gs)), scale=0.5)
            else:
                for p, g in zip(ps, gs):
                    sgd.apply(p.name, p, g, scale=0.5)
            sgd.step()
        return [tensor.to_numpy(p) for p in ps]

    for a, b in zip(run(True), run(False)):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)


def _run_fork(dev, fold):
    from singa_amd import autograd, layer, tensor
    import numpy as np
    autograd.training = True
    saved = autograd._Conv2d.__init__
    if not fold:
        def no_fold(self, handle, odd_padding=(0, 0, 0, 0)):
            saved(self, handle, odd_padding)
            self.fuses_dx_addend = False
        autograd._Conv2d.__init__ = no_fold
    try:
        rng = np.random.RandomState(3)
        x = tensor.from_numpy(rng.randn(2, 4, 9, 9).astype(np.float32), dev)
        x.requires_grad = True
        x.stores_grad = True
        dev.SetRandSeed(11)
        c_in = layer.Conv2d(8, 3, padding=1, bias=False)
        c_a = layer.Conv2d(8, 3, padding=1, stride=1, bias=False)
        c_b = layer.Conv2d(8, 1, bias=True)
        c_s = layer.Conv2d(8, 3, padding=1, stride=2, bias=False)
        x0 = c_in(x)
        y = autograd.add(autograd.add(c_a(x0), c_b(x0)), x0)
        z = c_s(x0)
        t = tensor.from_numpy(rng.randn(*y.shape).astype(np.float32), dev)
        tz = tensor.from_numpy(rng.randn(*z.shape).astype(np.float32), dev)
        loss = autograd.add(autograd.mse_loss(y, t), autograd.mse_loss(z, tz))
        return [tensor.to_numpy(g) for _, g in autograd.backward(loss)]
    finally:
        autograd._Conv2d.__init__ = saved


def _check_fold(dev, tol):
    import numpy as np
    a = _run_fork(dev, True)
    b = _run_fork(dev, False)
    assert len(a) == len(b) and len(a) >= 5
    for u, v in zip(a, b):
        np.testing.assert_allclose(u, v, rtol=tol, atol=tol)


def test_conv_backward_folds_fork_gradient_cpu():
    from singa_amd import device
    _check_fold(device.get_default_device(), 1e-5)


@pytest.mark.gpu
def test_conv_backward_folds_fork_gradient_gpu():
    from singa_amd import device
    _check_fold(device.create_cuda_gpu(), 1e-4)


@pytest.mark.gpu
def test_conv_bwd_data_acc_kernel_matches_reference():
    """Every launch plan of the accumulate path: direct epilogue add, split
    reduction seeded from the addend, strided classes, untouched classes."""
    import torch
    from singa_amd import device
    dev = device.create_cuda_gpu()
    m = dev.math
    cases = [  # N, C, H, W, K, R, stride, pad
        (4, 64, 56, 56, 64, 1, 1, 0),     # many tiles, no split
        (2, 512, 7, 7, 512, 3, 1, 1),     # few tiles: split reduction
        (2, 32, 28, 28, 64, 3, 2, 1),     # stride classes
        (2, 16, 16, 16, 32, 1, 2, 0),     # classes no tap reaches
        (2, 8, 15, 15, 8, 3, 3, 1),       # nine classes, two launches
    ]
    for (N, C, H, W, K, R, st, pd) in cases:
        P = (H + 2 * pd - R) // st + 1
        dy = torch.randn(N, K, P, P, device="cuda")
        w = torch.randn(K, C, R, R, device="cuda") * 0.1
        add = torch.randn(N, C, H, W, device="cuda")
        dx = torch.full((N, C, H, W), float("nan"), device="cuda")
        m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1, add)
        ref = torch.nn.grad.conv2d_input((N, C, H, W), w.double().cpu(),
                                         dy.double().cpu(), stride=st,
                                         padding=pd) + add.double().cpu()
        err = (dx.double().cpu() - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), \
            ((N, C, H, W, K, R, st, pd), err)


def _half_sgd_drift(dev, master):
    """200 steps with a gradient whose step (lr * g = 1e-4) is far below the
    rounding unit of fp16 around 1.0 (2^-10): without a master copy nothing
    moves, with it the parameter follows the exact trajectory."""
    import numpy as np
    from singa_amd import opt, tensor
    p = tensor.from_numpy(np.ones((64, 33), dtype=np.float16), dev)
    g = tensor.from_numpy(np.full((64, 33), 1e-2, dtype=np.float16), dev)
    p.name = "w"
    sgd = opt.SGD(lr=0.01, momentum=0.0, master_weights=master)
    for _ in range(200):
        sgd.apply_many([(p, g)])
        sgd.step()
    return float(tensor.to_numpy(p).astype(np.float32).mean())


def test_sgd_master_weights_accumulate_small_steps_cpu():
    from singa_amd import device
    dev = device.get_default_device()
    assert abs(_half_sgd_drift(dev, True) - (1.0 - 200 * 1e-4)) < 2e-3
    assert _half_sgd_drift(dev, False) == 1.0


@pytest.mark.gpu
def test_sgd_master_weights_accumulate_small_steps_gpu():
    from singa_amd import device
    dev = device.create_cuda_gpu()
    assert abs(_half_sgd_drift(dev, True) - (1.0 - 200 * 1e-4)) < 2e-3
    assert _half_sgd_drift(dev, False) == 1.0


def _sgd16_vs_fp32(dev, master):
    """Ten momentum steps on 16-bit parameters with 16-bit hyper-parameter
    tensors (lr, momentum, dampening, weight decay are converted on the fly);
    distance to the float32 trajectory on the host."""
    import numpy as np
    from singa_amd import device, opt, tensor

    def run(d, dt, master_weights):
        rng = np.random.RandomState(0)
        npdt = np.float16 if dt == tensor.float16 else np.float32
        ps = []
        for i, shp in enumerate([(50, 20, 5, 5), (500,), (80, 50), (7,)]):
            p = tensor.from_numpy((rng.randn(*shp) * 0.1).astype(npdt), d)
            p.name = "p%d" % i
            ps.append(p)
        sgd = opt.SGD(lr=0.005, momentum=0.9, weight_decay=1e-5, dtype=dt,
                      master_weights=master_weights)
        for _ in range(10):
            gs = [tensor.from_numpy(
                (rng.randn(*p.shape) * 0.05).astype(npdt), d) for p in ps]
            sgd.apply_many(list(zip(ps, gs)))
            sgd.step()
        return [tensor.to_numpy(p).astype(np.float32) for p in ps]

    ref = run(device.get_default_device(), tensor.float32, False)
    out = run(dev, tensor.float16, master)
    return max(float(np.abs(a - b).max()) for a, b in zip(out, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("master", [False, True])
def test_sgd_16bit_hyper_parameters_gpu(master):
    from singa_amd import device
    assert _sgd16_vs_fp32(device.create_cuda_gpu(), master) < 2e-3


@pytest.mark.parametrize("master", [False, True])
def test_sgd_16bit_hyper_parameters_cpu(master):
    from singa_amd import device
    assert _sgd16_vs_fp32(device.get_default_device(), master) < 2e-3


def _conv_bn_net(seed):
    """conv -> bn -> relu -> conv(+bias) -> bn, trained for a few steps;
    returns the loss trajectory, the running statistics and the handles."""
    from singa_amd import layer, model, opt, tensor, device
    from singa_amd.core import nn_ops

    class Net(model.Model):

        def __init__(self):
            super(Net, self).__init__()
            self.c1 = layer.Conv2d(8, 3, padding=1, bias=False)
            self.b1 = layer.BatchNorm2d()
            self.r1 = layer.ReLU()
            self.c2 = layer.Conv2d(6, 1, bias=True)
            self.b2 = layer.BatchNorm2d()
            self.flat = layer.Flatten()
            self.fc = layer.Linear(5)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            y = self.b2(self.c2(self.r1(self.b1(self.c1(x)))))
            return self.fc(self.flat(y))

        def train_one_batch(self, x, t):
            out = self.forward(x)
            l = self.loss(out, t)
            self.optimizer(l)
            return out, l

    dev = device.get_default_device()
    dev.SetRandSeed(seed)
    rs = np.random.RandomState(seed)
    x = tensor.from_numpy(rs.randn(4, 3, 6, 6).astype(np.float32) + 3.0)
    t = tensor.from_numpy(rs.randint(0, 5, 4).astype(np.int32))
    m = Net()
    m.set_optimizer(opt.SGD(lr=0.05, momentum=0.9))
    m.compile([x], is_train=True, use_graph=False, sequential=False)
    losses = []
    for _ in range(4):
        _, l = m(x, t)
        losses.append(float(tensor.to_numpy(l).reshape(-1)[0]))
    stats = [tensor.to_numpy(v) for v in (m.b1.running_mean, m.b1.running_var,
                                          m.b2.running_mean, m.b2.running_var)]
    return losses, stats, (m.c1.handle, m.c2.handle), nn_ops


def test_conv_hands_batchnorm_statistics_cpu(monkeypatch):
    """A batch normalisation that reads a convolution's output takes its
    statistics from the sums the convolution produced; trajectory and running
    statistics equal those of the two-pass normalisation."""
    from singa_amd.core import nn_ops as ops
    monkeypatch.setattr(ops, "CONV_BN_STATS", True)
    losses, stats, handles, nn_ops = _conv_bn_net(3)
    for h in handles:
        link = h.stat_link
        assert link is not None and link.token == 4 and not link.pending
        # consumed and cleared
        assert float(link.stats.view().abs().max()) == 0.0
    monkeypatch.setattr(nn_ops, "CONV_BN_STATS", False)
    ref_losses, ref_stats, ref_handles, _ = _conv_bn_net(3)
    assert all(h.stat_link is None for h in ref_handles)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for a, b in zip(stats, ref_stats):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6)


# ---- linear layer with the bias in the product's epilogue ----------------------
@pytest.mark.parametrize("dtype_name,tol", [("float32", 2e-5),
                                            ("float16", 2e-2),
                                            ("bfloat16", 1e-1)])
@pytest.mark.parametrize("shape", [(6, 5, 24), (33, 40), (2, 3, 4, 8)])
@pytest.mark.parametrize("streams", [False, True])
def test_linear_bias_forward_backward(dev, dtype_name, tol, shape, streams,
                                      monkeypatch):
    """``streams``: the three gradient kernels on streams of their own (GPU
    only; one stream on the host either way)."""
    from singa_amd import autograd, tensor
    from singa_amd.core import tensor_core
    monkeypatch.setattr(tensor_core, "_LINEAR_BWD_STREAMS", streams)
    dt = getattr(tensor, dtype_name)
    rs = np.random.RandomState(0)
    K, N = shape[-1], 16
    x = rs.randn(*shape).astype(np.float32)
    w = (rs.randn(K, N) * 0.3).astype(np.float32)
    b = rs.randn(N).astype(np.float32)
    dy = rs.randn(*shape[:-1], N).astype(np.float32)

    def T(a, grad=True):
        t = tensor.from_numpy(a, dev).as_type(dt)
        t.requires_grad = grad
        t.stores_grad = grad
        return t
    prev = autograd.training
    autograd.training = True
    try:
        tx, tw, tb = T(x), T(w), T(b)
        y = autograd.linear_bias(tx, tw, tb)
        assert y.shape == tuple(shape[:-1]) + (N,) and y.dtype == dt
        dx, dw, db = y.creator.backward(T(dy, False).data)
    finally:
        autograd.training = prev

    def N_(core):
        return tensor.to_numpy(tensor.from_raw_tensor(core).as_type(
            tensor.float32)) if not isinstance(core, tensor.Tensor) else \
            tensor.to_numpy(core.as_type(tensor.float32))
    x2, dy2 = x.reshape(-1, K).astype(np.float64), dy.reshape(-1, N)
    scale = max(1.0, float(np.abs(x2 @ w).max()))
    np.testing.assert_allclose(N_(y), x @ w + b, atol=tol * scale)
    np.testing.assert_allclose(N_(dx), (dy2 @ w.T).reshape(shape),
                               atol=tol * scale)
    np.testing.assert_allclose(N_(dw), x2.T @ dy2,
                               atol=tol * max(1.0, np.abs(x2.T @ dy2).max()))
    np.testing.assert_allclose(N_(db), dy2.sum(0),
                               atol=tol * max(1.0, np.abs(dy2.sum(0)).max()))


@pytest.mark.parametrize("dtype_name", ["float32", "float16", "bfloat16"])
@pytest.mark.parametrize("rows,cols", [(1, 8), (300, 768), (1025, 36),
                                       (700, 13), (8192, 768), (2048, 3072),
                                       (600, 70001)])
def test_column_sums_read_the_input_type(dev, dtype_name, rows, cols):
    """Sum over axis 0 of a matrix of any float type (float accumulation;
    16-byte loads when the row length allows, single elements otherwise).
    On the GPU the row slices of one launch meet in a scratch buffer that the
    launch must leave clean: the sum is taken three times in a row.  70001
    columns exceed that scratch (separate fill / sum / convert passes)."""
    from singa_amd import singa_wrap as singa
    from singa_amd import tensor
    dt = getattr(tensor, dtype_name)
    a = np.random.RandomState(1).randn(rows, cols).astype(np.float32)
    t = tensor.from_numpy(a, dev).as_type(dt)
    ref = tensor.to_numpy(t.as_type(tensor.float32)).astype(np.float64).sum(0)
    tol = {"float32": 1e-4, "float16": 4e-3, "bfloat16": 3e-2}[dtype_name]
    for _ in range(3):
        got = tensor.to_numpy(tensor.from_raw_tensor(
            singa.Sum(t.data, 0)).as_type(tensor.float32))
        np.testing.assert_allclose(got, ref,
                                   atol=tol * max(1.0, np.abs(ref).max()))
