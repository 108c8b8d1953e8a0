"""Training-free architecture scores computed with this framework's autograd
on one mini-batch: no weight update, one forward and one backward pass each.

    synflow        sum |w * d R / d w| of the linearised network (|w|, all-ones
                   input), the score the reference's singa build uses
                   (src/eva_engine/phase1/algo/prune_synflow.py)
    grad_norm      sum of the gradient norms of the loss
    snip           sum |w * d L / d w|
    weight_norm    sum of the weight norms
    nas_wot        log-determinant of the kernel of ReLU activation patterns
    express_flow   per-neuron flow |z * d y / d z|, summed over the batch by a
                   gate of ones on every hidden layer and then over neurons
    fisher         sum over hidden neurons of the batch mean of
                   (activation * d L / d activation)^2 / 2
    grasp          - sum (H g) * w with the Hessian-vector product taken as a
                   finite difference of two gradients
    ntk_trace      trace of the empirical neural tangent kernel of the summed
                   logits over the first samples of the batch
    ntk_cond_num   minus the condition number of that kernel (a well
                   conditioned kernel trains fast: higher is better)

Role: reference src/eva_engine/phase1/evaluator.py + algo/."""
import time

import numpy as np

from singa_amd import autograd, opt, tensor


def build(space, arch_id, dev, batch):
    """(model, input tensors, label tensor) of ``arch_id`` compiled for the
    shapes of ``batch`` = (*inputs, labels) in numpy."""
    *xs, y = batch
    m = space.new_architecture(arch_id)
    m.set_optimizer(opt.SGD(lr=0.05, momentum=0.9))
    ts = space.inputs(dev, *xs)
    m.compile(ts, is_train=True, use_graph=False, sequential=False)
    ty = tensor.from_numpy(np.ascontiguousarray(y, np.int32), dev)
    return m, ts, ty


class _training(object):
    def __enter__(self):
        self.prev = autograd.training
        autograd.training = True

    def __exit__(self, *a):
        autograd.training = self.prev


def _np(t):
    return tensor.to_numpy(t).astype(np.float64)


def _weights(m):
    return {k: v for k, v in m.get_params().items()}


def grad_norm(m, xs, y):
    with _training():
        loss = m.loss_fn(m.forward(*xs), y)
        return float(sum(np.sqrt((_np(g) ** 2).sum())
                         for _, g in autograd.backward(loss)))


def snip(m, xs, y):
    with _training():
        loss = m.loss_fn(m.forward(*xs), y)
        return float(sum(np.abs(_np(p) * _np(g)).sum()
                         for p, g in autograd.backward(loss)))


def weight_norm(m, xs, y):
    return float(sum(np.sqrt((_np(p) ** 2).sum())
                     for p in _weights(m).values()))


def synflow(m, xs, y):
    params = _weights(m)
    saved = {k: tensor.to_numpy(v) for k, v in params.items()}
    try:
        for k, v in params.items():                 # linearise: w <- |w|
            v.copy_from_numpy(np.abs(saved[k]))
        ones = []
        for t in xs:
            a = tensor.to_numpy(t)
            ones.append(t if a.dtype.kind in "iu" else tensor.from_numpy(
                np.ones_like(a), t.device))
        with _training():
            out = m.forward(*ones)
            r = autograd.reshape(autograd.reduce_sum(out, None, 0), (1,))
            score = sum(np.abs(_np(p) * _np(g)).sum()
                        for p, g in autograd.backward(r))
        return float(np.log1p(score))
    finally:
        for k, v in params.items():
            v.copy_from_numpy(saved[k])


def nas_wot(m, xs, y):
    if not hasattr(m, "taps"):
        raise ValueError("nas_wot needs a model that records its hidden "
                         "activations (taps)")
    m.taps = []
    try:
        m.forward(*xs)
        codes = np.concatenate([(_np(t) > 0).reshape(len(_np(t)), -1)
                                for t in m.taps], axis=1).astype(np.float64)
    finally:
        m.taps = None
    k = codes @ codes.T + (1 - codes) @ (1 - codes.T)
    sign, logdet = np.linalg.slogdet(k + 1e-3 * np.eye(len(k)))
    return float(logdet)


def express_flow(m, xs, y):
    if not hasattr(m, "taps"):
        raise ValueError("express_flow needs a model with hidden taps")
    gates = []
    acts = m.acts
    originals = [a.forward for a in acts]

    def gated(fwd):
        def run(x):
            h = fwd(x)
            g = tensor.from_numpy(np.ones((h.shape[-1],), np.float32),
                                  h.device)
            g.requires_grad = True
            g.stores_grad = True
            gates.append(g)
            return autograd.mul(h, g)
        return run
    try:
        for a, f in zip(acts, originals):
            a.forward = gated(f)
        with _training():
            out = m.forward(*xs)
            r = autograd.reshape(autograd.reduce_sum(out, None, 0), (1,))
            grads = {id(p): g for p, g in autograd.backward(r)}
    finally:
        for a, f in zip(acts, originals):
            a.forward = f
    score = 0.0
    for g in gates:
        score += np.abs(_np(grads[id(g)])).sum()
    return float(np.log1p(score))


def _hidden_gates(m, xs, y, per_sample, with_loss):
    """d target / d gate for a gate of ones multiplied into every hidden
    activation: (batch, neurons) per layer when ``per_sample`` else
    (neurons,).  The target is the loss or the sum of the logits."""
    if not hasattr(m, "acts"):
        raise ValueError("this score needs a model that lists its hidden "
                         "activation layers (acts)")
    gates = []
    originals = [a.forward for a in m.acts]

    def gated(fwd):
        def run(x):
            h = fwd(x)
            shape = tuple(h.shape) if per_sample else (h.shape[-1],)
            g = tensor.from_numpy(np.ones(shape, np.float32), h.device)
            g.requires_grad = True
            g.stores_grad = True
            gates.append(g)
            return autograd.mul(h, g)
        return run
    try:
        for a, f in zip(m.acts, originals):
            a.forward = gated(f)
        with _training():
            out = m.forward(*xs)
            r = m.loss_fn(out, y) if with_loss else autograd.reshape(
                autograd.reduce_sum(out, None, 0), (1,))
            grads = {id(p): g for p, g in autograd.backward(r)}
    finally:
        for a, f in zip(m.acts, originals):
            a.forward = f
    return [_np(grads[id(g)]) for g in gates]


def fisher(m, xs, y):
    total = 0.0
    for g in _hidden_gates(m, xs, y, per_sample=True, with_loss=True):
        g = g.reshape(len(g), -1) * len(g)      # the loss is a batch mean
        total += 0.5 * (g ** 2).mean(axis=0).sum()
    return float(np.log1p(total))


def _loss_grads(m, xs, y):
    with _training():
        loss = m.loss_fn(m.forward(*xs), y)
        return {id(p): (p, _np(g)) for p, g in autograd.backward(loss)}


def grasp(m, xs, y, eps=1e-2):
    base = _loss_grads(m, xs, y)
    saved = {k: tensor.to_numpy(p) for k, (p, _) in base.items()}
    norm = np.sqrt(sum((g ** 2).sum() for _, g in base.values())) or 1.0
    try:
        for k, (p, g) in base.items():          # w <- w + eps * g / |g|
            p.copy_from_numpy((saved[k] + eps * g / norm).astype(np.float32))
        moved = _loss_grads(m, xs, y)
    finally:
        for k, (p, _) in base.items():
            p.copy_from_numpy(saved[k])
    score = 0.0
    for k, (p, g) in base.items():
        hg = (moved[k][1] - g) * (norm / eps)
        score += -(hg * saved[k]).sum()
    return float(score)


def _ntk(m, xs, y, samples=8):
    n = min(samples, xs[0].shape[0])
    rows = []
    for i in range(n):
        one = []
        for t in xs:
            a = tensor.to_numpy(t)[i:i + 1]
            one.append(tensor.from_numpy(np.ascontiguousarray(a), t.device))
        with _training():
            out = m.forward(*one)
            r = autograd.reshape(autograd.reduce_sum(out, None, 0), (1,))
            grads = sorted(((p.name if hasattr(p, "name") else str(id(p)),
                             _np(g).reshape(-1))
                            for p, g in autograd.backward(r)),
                           key=lambda kv: kv[0])
        rows.append(np.concatenate([g for _, g in grads]))
    j = np.stack(rows)
    return j @ j.T


def ntk_trace(m, xs, y):
    return float(np.log1p(np.trace(_ntk(m, xs, y))))


def ntk_cond_num(m, xs, y):
    ev = np.linalg.eigvalsh(_ntk(m, xs, y))
    return float(-ev[-1] / max(ev[0], 1e-12 * max(ev[-1], 1e-30)))


METRICS = {"synflow": synflow, "grad_norm": grad_norm, "snip": snip,
           "weight_norm": weight_norm, "nas_wot": nas_wot,
           "express_flow": express_flow, "fisher": fisher, "grasp": grasp,
           "ntk_trace": ntk_trace, "ntk_cond_num": ntk_cond_num}


class P1Evaluator(object):
    """Scores architecture ids on one fixed mini-batch and keeps the time each
    evaluation took (the coordinator's ``score_time_per_model``)."""

    def __init__(self, space, dev, batch, metric="express_flow", cache=None,
                 dataset="data"):
        self.space, self.dev, self.batch = space, dev, batch
        self.metric, self.cache, self.dataset = metric, cache, dataset
        self.times = []

    def p1_evaluate(self, arch_id):
        if self.cache is not None:
            hit = self.cache.get(self.dataset, self.metric, arch_id)
            if hit is not None:
                return hit
        t0 = time.perf_counter()
        m, xs, y = build(self.space, arch_id, self.dev, self.batch)
        score = METRICS[self.metric](m, xs, y)
        self.times.append(time.perf_counter() - t0)
        if self.cache is not None:
            self.cache.put(self.dataset, self.metric, arch_id, score)
        return score

    def mean_time(self):
        return float(np.mean(self.times)) if self.times else 0.0
