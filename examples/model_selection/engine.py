"""Two-phase model selection (filter with a training-free score, refine with
successive halving), on this framework's autograd.

Phase 1 ("filtering"): candidate architectures are drawn from a search space
by a sampler (random or regularised evolution) and ranked by a proxy that
needs no training: the gradient-norm score (sum of |dL/dW| on one batch) or a
SynFlow-style score (sum of |W * dR/dW| with R = sum of the outputs of the
network evaluated on an all-ones input with |W|).  Phase 2 ("refinement"): the
best K candidates enter successive halving -- train each for a small budget,
keep the better half, double the budget -- until one remains.

Counterpart of the Python engine of the reference's
examples/model_selection/Trails (samplers, training-free evaluators,
successive halving, coordinator); the search space here is MLPs over tabular
data so the example runs in seconds.
"""
import os
import random
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", ".."))

from singa_amd import autograd, device, layer, model, opt, tensor  # noqa: E402


# ---------------------------------------------------------------- search space
class MlpSpace(object):
    """Architectures = tuples of hidden widths, 1..max_depth layers."""

    def __init__(self, widths=(8, 16, 32, 64, 128), max_depth=3):
        self.widths, self.max_depth = tuple(widths), max_depth

    def random(self, rng):
        depth = rng.randint(1, self.max_depth)
        return tuple(rng.choice(self.widths) for _ in range(depth))

    def mutate(self, arch, rng):
        arch = list(arch)
        move = rng.choice(["widen", "add", "drop"])
        if move == "add" and len(arch) < self.max_depth:
            arch.insert(rng.randint(0, len(arch)), rng.choice(self.widths))
        elif move == "drop" and len(arch) > 1:
            arch.pop(rng.randint(0, len(arch) - 1))
        else:
            arch[rng.randint(0, len(arch) - 1)] = rng.choice(self.widths)
        return tuple(arch)


class Mlp(model.Model):

    def __init__(self, arch, num_classes):
        super().__init__()
        self.hidden = [layer.Linear(w) for w in arch]
        self.register_layers(*self.hidden)
        self.relu = layer.ReLU()
        self.out = layer.Linear(num_classes)
        self.loss = layer.SoftMaxCrossEntropy()

    def forward(self, x):
        for h in self.hidden:
            x = self.relu(h(x))
        return self.out(x)

    def train_one_batch(self, x, y):
        out = self.forward(x)
        loss = self.loss(out, y)
        self.optimizer(loss)
        return out, loss


# -------------------------------------------------------------------- samplers
class RandomSampler(object):

    def __init__(self, space, seed=0):
        self.space, self.rng = space, random.Random(seed)

    def sample(self):
        return self.space.random(self.rng)

    def feedback(self, arch, score):
        pass


class EvolutionSampler(object):
    """Regularised evolution: mutate the best of a random tournament drawn
    from a sliding population of the most recent candidates."""

    def __init__(self, space, population=8, tournament=3, seed=0):
        self.space, self.rng = space, random.Random(seed)
        self.population, self.tournament = population, tournament
        self.history = []

    def sample(self):
        if len(self.history) < self.population:
            return self.space.random(self.rng)
        pool = self.rng.sample(self.history[-self.population:],
                               self.tournament)
        parent = max(pool, key=lambda t: t[1])[0]
        return self.space.mutate(parent, self.rng)

    def feedback(self, arch, score):
        self.history.append((arch, score))


# ---------------------------------------------------------- training-free scores
def _build(arch, x, num_classes, dev):
    m = Mlp(arch, num_classes)
    m.set_optimizer(opt.SGD(lr=0.05, momentum=0.9))
    tx = tensor.from_numpy(x, dev)
    m.compile([tx], is_train=True, use_graph=False, sequential=False)
    return m


def grad_norm_score(arch, x, y, num_classes, dev):
    m = _build(arch, x, num_classes, dev)
    autograd.training = True
    loss = m.loss(m.forward(tensor.from_numpy(x, dev)),
                  tensor.from_numpy(y, dev))
    score = 0.0
    for _, g in autograd.backward(loss):
        score += float(np.abs(tensor.to_numpy(g)).sum())
    autograd.training = False
    return score


def synflow_score(arch, x, y, num_classes, dev):
    m = _build(arch, x, num_classes, dev)
    params = m.get_params()
    saved = {k: tensor.to_numpy(v) for k, v in params.items()}
    for k, v in params.items():                 # linearise: W <- |W|
        v.copy_from_numpy(np.abs(saved[k]))
    autograd.training = True
    ones = tensor.from_numpy(np.ones((1, x.shape[1]), np.float32), dev)
    out = m.forward(ones)
    r = autograd.reduce_sum(out, None, 0)
    r = autograd.reshape(r, (1,))
    score = 0.0
    for p, g in autograd.backward(r):
        score += float(np.abs(tensor.to_numpy(p) * tensor.to_numpy(g)).sum())
    autograd.training = False
    return float(np.log1p(score))


SCORES = {"grad_norm": grad_norm_score, "synflow": synflow_score}


# --------------------------------------------------------------------- phases
def filtering_phase(space, data, n_candidates=20, sampler="evolution",
                    score="synflow", seed=0, dev=None):
    dev = dev or device.get_default_device()
    x, y, num_classes = data
    smp = (EvolutionSampler if sampler == "evolution" else RandomSampler)(
        space, seed=seed)
    seen = {}
    while len(seen) < n_candidates:
        arch = smp.sample()
        if arch in seen:
            smp.feedback(arch, seen[arch])
            if len(seen) >= len(space.widths) ** space.max_depth:
                break
            arch = space.random(smp.rng)
            if arch in seen:
                continue
        s = SCORES[score](arch, x[:64], y[:64], num_classes, dev)
        seen[arch] = s
        smp.feedback(arch, s)
    return sorted(seen.items(), key=lambda kv: -kv[1])


def _train_eval(arch, data, val, epochs, dev, bs=64):
    x, y, num_classes = data
    vx, vy = val
    m = _build(arch, x[:bs], num_classes, dev)
    m.train()
    for _ in range(epochs):
        for b in range(len(x) // bs):
            m(tensor.from_numpy(x[b * bs:(b + 1) * bs], dev),
              tensor.from_numpy(y[b * bs:(b + 1) * bs], dev))
    m.eval()
    pred = np.argmax(tensor.to_numpy(m(tensor.from_numpy(vx, dev))), 1)
    return float((pred == vy).mean())


def refinement_phase(candidates, data, val, min_epochs=1, eta=2, dev=None):
    """Successive halving; returns (best arch, its accuracy, the trace)."""
    dev = dev or device.get_default_device()
    alive, budget, trace = list(candidates), min_epochs, []
    best = (alive[0], 0.0)
    while True:
        scored = [(a, _train_eval(a, data, val, budget, dev)) for a in alive]
        scored.sort(key=lambda t: -t[1])
        trace.append((budget, scored))
        best = scored[0]
        if len(scored) == 1:
            break
        alive = [a for a, _ in scored[:max(1, len(scored) // eta)]]
        budget *= eta
    return best[0], best[1], trace


def model_selection(data, val, space=None, n_candidates=16, top_k=4,
                    sampler="evolution", score="synflow", seed=0):
    """Coordinator: filter N candidates down to K, refine to one."""
    space = space or MlpSpace()
    ranked = filtering_phase(space, data, n_candidates, sampler, score, seed)
    top = [a for a, _ in ranked[:top_k]]
    arch, acc, trace = refinement_phase(top, data, val)
    return {"best_arch": arch, "val_acc": acc, "filtered": ranked[:top_k],
            "rounds": [(b, [(list(a), round(s, 4)) for a, s in sc])
                       for b, sc in trace]}


def toy_data(n=768, dim=12, classes=3, seed=0):
    rng = np.random.RandomState(seed)
    centers = rng.randn(classes, dim).astype(np.float32) * 1.5
    y = rng.randint(0, classes, n).astype(np.int32)
    x = centers[y] + rng.randn(n, dim).astype(np.float32)
    cut = int(0.75 * n)
    return (x[:cut], y[:cut], classes), (x[cut:], y[cut:])


if __name__ == "__main__":
    import json
    data, val = toy_data()
    print(json.dumps(model_selection(data, val), indent=1, default=list))
