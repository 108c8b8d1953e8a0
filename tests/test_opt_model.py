"""Optimizers, learning-rate schedules, initializers and the Model API (graph
buffering, the four execution modes, checkpoint save / load).

Cases follow reference test/python/test_opt.py, test_initializer.py and
test_model.py; the optimizer maths is checked against plain PyTorch / numpy.
"""
import os

import numpy as np
import pytest
import torch

from singa_amd import autograd, initializer, layer, model, opt, tensor

RS = np.random.RandomState(3)


def P(a, dev, name):
    t = tensor.Tensor(device=dev, data=np.asarray(a, np.float32),
                      requires_grad=True, stores_grad=True, name=name)
    return t


# ---------------------------------------------------------------- schedulers
def test_decay_schedulers(dev):
    step = tensor.Tensor((1,), dev)
    step.set_value(0)
    c = opt.Constant(0.1)
    assert tensor.to_numpy(c(step))[0] == pytest.approx(0.1)
    e = opt.ExponentialDecay(0.1, 2, 0.5)
    step.set_value(4)
    assert tensor.to_numpy(e(step))[0] == pytest.approx(0.1 * 0.5 ** 2)
    e = opt.ExponentialDecay(0.1, 2, 0.5, True)
    step.set_value(3)
    assert tensor.to_numpy(e(step))[0] == pytest.approx(0.1 * 0.5 ** 1)


# ----------------------------------------------------------------- optimizers
def _run(dev, singa_opt, torch_opt_fn, steps=4):
    w0 = RS.randn(4, 3).astype(np.float32)
    p = P(w0, dev, "w")
    pt = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    to = torch_opt_fn([pt])
    for i in range(steps):
        g = RS.randn(4, 3).astype(np.float32)
        singa_opt.apply("w", p, tensor.Tensor(device=dev, data=g))
        singa_opt.step()
        pt.grad = torch.from_numpy(g.copy())
        to.step()
    np.testing.assert_allclose(tensor.to_numpy(p), pt.detach().numpy(),
                               rtol=1e-4, atol=1e-5)


def test_sgd_plain(dev):
    _run(dev, opt.SGD(lr=0.1), lambda ps: torch.optim.SGD(ps, lr=0.1))


def test_sgd_momentum_decay(dev):
    _run(dev, opt.SGD(lr=0.1, momentum=0.9, weight_decay=1e-2),
         lambda ps: torch.optim.SGD(ps, lr=0.1, momentum=0.9,
                                    weight_decay=1e-2))


def test_sgd_dampening(dev):
    # the first step also applies the dampening (reference opt.py:273-281);
    # torch skips it when it creates the buffer, so compare with numpy
    w0 = RS.randn(5).astype(np.float32)
    p = P(w0, dev, "w")
    o = opt.SGD(lr=0.1, momentum=0.9, dampening=0.1)
    w, v = w0.copy(), np.zeros(5, np.float32)
    for _ in range(3):
        g = RS.randn(5).astype(np.float32)
        o.apply("w", p, tensor.Tensor(device=dev, data=g))
        o.step()
        v = 0.9 * v + 0.9 * g
        w = w - 0.1 * v
    np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=1e-4, atol=1e-5)


def test_sgd_nesterov(dev):
    _run(dev, opt.SGD(lr=0.05, momentum=0.8, nesterov=True),
         lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.8, nesterov=True))
    with pytest.raises(ValueError):
        opt.SGD(lr=0.1, momentum=0.0, nesterov=True)


def test_sgd_lr_schedule(dev):
    w0 = np.ones(3, np.float32)
    p = P(w0, dev, "w")
    o = opt.SGD(lr=opt.ExponentialDecay(0.1, 1, 0.5))
    g = np.ones(3, np.float32)
    w = w0.copy()
    for i in range(3):
        o.apply("w", p, tensor.Tensor(device=dev, data=g))
        o.step()
        w = w - 0.1 * 0.5 ** i * g
    np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=1e-5)


def test_rmsprop(dev):
    w0 = RS.randn(6).astype(np.float32)
    p = P(w0, dev, "w")
    o = opt.RMSProp(lr=0.01, rho=0.9, epsilon=1e-8)
    w, r = w0.copy(), np.zeros(6, np.float32)
    for _ in range(3):
        g = RS.randn(6).astype(np.float32)
        o.apply("w", p, tensor.Tensor(device=dev, data=g))
        o.step()
        r = 0.9 * r + 0.1 * g * g
        w = w - 0.01 * g / np.sqrt(r + 1e-8)
    np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=1e-4, atol=1e-5)


def test_adagrad(dev):
    w0 = RS.randn(6).astype(np.float32)
    p = P(w0, dev, "w")
    o = opt.AdaGrad(lr=0.1, epsilon=1e-8)
    w, h = w0.copy(), np.zeros(6, np.float32)
    for _ in range(3):
        g = RS.randn(6).astype(np.float32)
        o.apply("w", p, tensor.Tensor(device=dev, data=g))
        o.step()
        h = h + g * g
        w = w - 0.1 * g / np.sqrt(h + 1e-8)
    np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=1e-4, atol=1e-5)


def test_adam(dev):
    w0 = RS.randn(6).astype(np.float32)
    p = P(w0, dev, "w")
    o = opt.Adam(lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-8)
    w = w0.copy()
    m, v = np.zeros(6, np.float32), np.zeros(6, np.float32)
    for t in range(1, 4):
        g = RS.randn(6).astype(np.float32)
        o.apply("w", p, tensor.Tensor(device=dev, data=g))
        o.step()
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w = w - 0.01 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) +
                                               1e-8)
    np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=1e-3, atol=1e-5)


def test_optimizer_states_roundtrip(dev):
    o = opt.SGD(lr=0.1, momentum=0.9)
    p = P(np.ones(3), dev, "w")
    o.apply("w", p, tensor.Tensor(device=dev, data=np.ones(3, np.float32)))
    o.step()
    st = o.get_states()
    assert st["step_counter"] == 1 and "w" in st["moments"]
    o2 = opt.SGD(lr=0.1, momentum=0.9)
    o2.set_states(st)
    assert tensor.to_numpy(o2.step_counter)[0] == 1


def test_adaptive_rules_fused_many(dev):
    """One launch updates several tensors; weight decay and the data-parallel
    gradient scale are part of the kernel."""
    shapes = [(7,), (3, 5), (4100,)]
    for make, rule in [
            (lambda: opt.RMSProp(lr=0.01, rho=0.8, weight_decay=0.1), "rms"),
            (lambda: opt.AdaGrad(lr=0.05, weight_decay=0.1), "ada"),
            (lambda: opt.Adam(lr=0.01, weight_decay=0.1), "adam")]:
        o = make()
        ws = [RS.randn(*s).astype(np.float32) for s in shapes]
        ps = [P(w, dev, "p%d" % i) for i, w in enumerate(ws)]
        s1 = [np.zeros_like(w) for w in ws]
        s2 = [np.zeros_like(w) for w in ws]
        for t in range(1, 4):
            gs = [RS.randn(*s).astype(np.float32) for s in shapes]
            o.apply_many([(p, tensor.Tensor(device=dev, data=g))
                          for p, g in zip(ps, gs)], scale=0.5)
            o.step()
            for i, g in enumerate(gs):
                g = 0.5 * g + 0.1 * ws[i]
                if rule == "rms":
                    s1[i] = 0.8 * s1[i] + 0.2 * g * g
                    ws[i] = ws[i] - 0.01 * g / np.sqrt(s1[i] + 1e-8)
                elif rule == "ada":
                    s1[i] = s1[i] + g * g
                    ws[i] = ws[i] - 0.05 * g / np.sqrt(s1[i] + 1e-8)
                else:
                    s1[i] = 0.9 * s1[i] + 0.1 * g
                    s2[i] = 0.999 * s2[i] + 0.001 * g * g
                    ws[i] = ws[i] - 0.01 * (s1[i] / (1 - 0.9 ** t)) / (
                        np.sqrt(s2[i] / (1 - 0.999 ** t)) + 1e-8)
        for p, w in zip(ps, ws):
            np.testing.assert_allclose(tensor.to_numpy(p), w, rtol=2e-3,
                                       atol=2e-5)


def test_set_states_keeps_tensor_identity(dev):
    """Restoring a checkpoint writes into the tensors a recorded step reads;
    it never rebinds them."""
    for make, tables in [
            (lambda: opt.SGD(lr=0.1, momentum=0.9), ["moments"]),
            (lambda: opt.Adam(lr=0.01), ["m", "v"]),
            (lambda: opt.RMSProp(lr=0.01), ["running_average"]),
            (lambda: opt.AdaGrad(lr=0.01), ["history"])]:
        src, dst = make(), make()
        for o, val in [(src, 2.0), (dst, 1.0)]:
            p = P(np.ones(3), dev, "w")
            for _ in range(int(val)):
                o.apply("w", p, tensor.Tensor(
                    device=dev, data=np.full(3, val, np.float32)))
                o.step()
        held = {n: getattr(dst, n)["w"] for n in tables}
        counter, lr_value = dst.step_counter, dst.lr_value
        dst.set_states(src.get_states())
        assert dst.step_counter is counter and dst.lr_value is lr_value
        assert tensor.to_numpy(dst.step_counter)[0] == 2
        for n in tables:
            assert getattr(dst, n)["w"] is held[n]
            np.testing.assert_array_equal(
                tensor.to_numpy(held[n]),
                tensor.to_numpy(getattr(src, n)["w"]))


# --------------------------------------------------------------- initializers
def test_initializers(dev):
    t = tensor.Tensor((40, 60), dev)
    initializer.he_uniform(t)
    a = tensor.to_numpy(t)
    lim = np.sqrt(6.0 / 40)
    assert np.abs(a).max() <= lim + 1e-6 and a.std() > 0.1 * lim
    initializer.he_normal(t)
    assert tensor.to_numpy(t).std() == pytest.approx(np.sqrt(2.0 / 40), rel=0.2)
    initializer.glorot_uniform(t)
    assert np.abs(tensor.to_numpy(t)).max() <= np.sqrt(6.0 / 100) + 1e-6
    initializer.glorot_normal(t)
    assert tensor.to_numpy(t).std() == pytest.approx(np.sqrt(2.0 / 100),
                                                     rel=0.2)
    initializer.lecun_uniform(t)
    assert np.abs(tensor.to_numpy(t)).max() <= np.sqrt(3.0 / 40) + 1e-6
    initializer.lecun_normal(t)
    assert tensor.to_numpy(t).std() == pytest.approx(np.sqrt(1.0 / 40), rel=0.2)
    initializer.gaussian(t, 40, 60)  # legacy form: std = sqrt(4 / fans)
    assert tensor.to_numpy(t).std() == pytest.approx(np.sqrt(4.0 / 100),
                                                     rel=0.2)
    initializer.xavier(t)
    initializer.glorot(t)
    initializer.msra(t)
    sq = tensor.Tensor((5, 7), dev)
    initializer.eye(sq)
    np.testing.assert_array_equal(tensor.to_numpy(sq), np.eye(5, 7))
    o = tensor.Tensor((6, 6), dev)
    initializer.orthogonal(o)
    q = tensor.to_numpy(o)
    np.testing.assert_allclose(q @ q.T, np.eye(6), atol=1e-4)


# ----------------------------------------------------------------------- model
class MLP(model.Model):

    def __init__(self, hidden=8, classes=3):
        super(MLP, self).__init__()
        self.fc1 = layer.Linear(hidden)
        self.relu = layer.ReLU()
        self.fc2 = layer.Linear(classes)
        self.loss = layer.SoftMaxCrossEntropy()

    def forward(self, x):
        return self.fc2(self.relu(self.fc1(x)))

    def train_one_batch(self, x, y):
        out = self.forward(x)
        l = self.loss(out, y)
        self.optimizer(l)
        return out, l


def _data(dev, n=16, d=5):
    rs = np.random.RandomState(0)
    x = rs.randn(n, d).astype(np.float32)
    y = rs.randint(0, 3, n).astype(np.int32)
    return (tensor.Tensor(device=dev, data=x),
            tensor.Tensor(device=dev, data=y))


def _train(dev, use_graph, sequential, steps=5, seed=0, after=None):
    dev.SetRandSeed(seed)
    m = MLP()
    m.set_optimizer(opt.SGD(lr=0.1, momentum=0.9))
    tx, ty = _data(dev)
    m.compile([tx], is_train=True, use_graph=use_graph, sequential=sequential)
    # same start for every mode
    rs = np.random.RandomState(1)
    for k, v in sorted(m.get_params().items()):
        v.copy_from_numpy(rs.randn(*v.shape).astype(np.float32) * 0.3)
    losses = []
    for _ in range(steps):
        out, l = m(tx, ty)
        losses.append(float(tensor.to_numpy(l).reshape(-1)[0]))
    params = {k: tensor.to_numpy(v) for k, v in m.get_params().items()}
    if after is not None:
        after()
    dev.EnableGraph(False)
    dev.ResetGraph()
    return losses, params, m


def test_model_modes_agree(dev):
    """eager == graph (breadth-first) == graph (serial)."""
    ref_l, ref_p, _ = _train(dev, False, False)
    assert ref_l[-1] < ref_l[0]
    for use_graph, seq in [(True, False), (True, True)]:
        l, p, _ = _train(dev, use_graph, seq)
        np.testing.assert_allclose(l, ref_l, rtol=1e-4, atol=1e-5)
        for (k, v), (k2, v2) in zip(sorted(p.items()), sorted(ref_p.items())):
            np.testing.assert_allclose(v, v2, rtol=1e-4, atol=1e-5)


def test_model_eval_and_param_names(dev):
    _, _, m = _train(dev, False, False, steps=1)
    tx, _ = _data(dev)
    m.eval()
    out = m(tx)
    assert out.shape == (16, 3)
    m.train()
    names = sorted(m.get_params().keys())
    assert names == sorted(["fc1.W", "fc1.b", "fc2.W", "fc2.b"])


def test_model_save_load(dev, tmp_path):
    _, params, m = _train(dev, False, False, steps=2)
    f = os.path.join(str(tmp_path), "ckpt.zip")
    m.save_states(f, aux_states={"epoch": tensor.from_numpy(
        np.array([3], np.float32))})
    m2 = MLP()
    m2.set_optimizer(opt.SGD(lr=0.1, momentum=0.9))
    tx, ty = _data(dev)
    m2.compile([tx], is_train=True, use_graph=False, sequential=False)
    aux = m2.load_states(f)
    assert "epoch" in aux
    assert float(tensor.to_numpy(aux["epoch"]).reshape(-1)[0]) == 3
    for k, v in m2.get_params().items():
        np.testing.assert_array_equal(tensor.to_numpy(v), params[k])


def test_load_states_under_recorded_graph(dev, tmp_path):
    """A checkpoint loaded after the step was recorded takes effect in the
    replayed step (parameters, momentum and step counter are overwritten in
    place)."""
    l_ref, p_ref, m = _train(dev, True, False, steps=3)
    # _train reset the device graph; record again and keep replaying
    tx, ty = _data(dev)
    m.compile([tx], is_train=True, use_graph=True, sequential=False)
    m(tx, ty)
    f = os.path.join(str(tmp_path), "mid.zip")
    aux = {"opt/" + k: v for k, v in m.optimizer.get_states()["moments"]
           .items()}
    m.save_states(f, aux_states=aux)
    snap = {k: tensor.to_numpy(v) for k, v in m.get_params().items()}
    m(tx, ty)
    after_one = {k: tensor.to_numpy(v) for k, v in m.get_params().items()}
    m(tx, ty)                              # drift away
    st = m.optimizer.get_states()
    loaded = m.load_states(f)
    m.optimizer.set_states({
        "step_counter": st["step_counter"] - 2,
        "moments": {k[len("opt/"):]: v for k, v in loaded.items()}})
    for k, v in m.get_params().items():
        np.testing.assert_array_equal(tensor.to_numpy(v), snap[k])
    m(tx, ty)                              # replay of the recorded step
    for k, v in m.get_params().items():
        np.testing.assert_allclose(tensor.to_numpy(v), after_one[k],
                                   rtol=1e-5, atol=1e-6)
    dev.EnableGraph(False)
    dev.ResetGraph()


def test_graph_profiling_output(dev, capsys):
    dev.SetVerbosity(1)
    dev.SetSkipIteration(1)
    try:
        _train(dev, True, False, steps=4, after=dev.PrintTimeProfiling)
    finally:
        dev.SetVerbosity(0)
    assert "Time Profiling" in capsys.readouterr().out


def test_recorded_inference_matches_eager(dev):
    """Evaluation through a recorded forward pass: per input signature one
    recording; later inputs (other tensor objects, other values, another
    batch size) give the eager results; a training step recorded on the same
    device keeps working next to it."""
    from singa_amd import layer, model, opt, tensor

    class Net(model.Model):
        def __init__(self):
            super().__init__()
            self.fc1 = layer.Linear(16)
            self.act = layer.ReLU()
            self.fc2 = layer.Linear(3)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

        def train_one_batch(self, x, y):
            out = self.forward(x)
            loss = self.loss(out, y)
            self.optimizer(loss)
            return out, loss

    rs = np.random.RandomState(0)
    m = Net()
    m.set_optimizer(opt.SGD(lr=0.05))
    x8 = tensor.from_numpy(rs.randn(8, 5).astype(np.float32), dev)
    y8 = tensor.from_numpy(rs.randint(0, 3, 8).astype(np.int32), dev)
    m.compile([x8], is_train=True, use_graph=True, sequential=False)
    m.graph_inference = True

    def eager(xn):
        m.graph_inference = False
        try:
            return tensor.to_numpy(m(tensor.from_numpy(xn, dev)))
        finally:
            m.graph_inference = True

    for rnd in range(3):
        m.train()
        for _ in range(3):
            _, loss = m(x8, y8)
        m.eval()
        for batch in (8, 8, 3, 8, 3):
            xn = rs.randn(batch, 5).astype(np.float32)
            got = tensor.to_numpy(m(tensor.from_numpy(xn, dev)))
            np.testing.assert_allclose(got, eager(xn), rtol=1e-5, atol=1e-6)
        assert len(m._inference.plans) == 2
    first = float(tensor.to_numpy(loss)[0])
    m.train()
    for _ in range(20):
        _, loss = m(x8, y8)
    assert float(tensor.to_numpy(loss)[0]) < first
    dev.ResetGraph()
