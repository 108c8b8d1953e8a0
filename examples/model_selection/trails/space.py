"""Search spaces.  An architecture is identified by a string (``"64-128-32"``)
so that it can travel through score caches, database tables and JSON.

Role: reference src/search_space/core/space.py (SpaceWrapper contract),
src/search_space/mlp_api/space.py (MlpSpace, DNNModel with a feature
embedding) and the CNN models of phase1/algo/singa_ms/cnn_ms."""
import itertools

import numpy as np

from singa_amd import autograd, layer, model, tensor


class SpaceWrapper(object):
    """What samplers, evaluators and the driver need from a search space."""
    name = "space"

    def decisions(self):
        """List of option lists, one per categorical decision."""
        raise NotImplementedError

    def encode(self, arch_id):
        """Option index per decision."""
        raise NotImplementedError

    def decode(self, choice):
        raise NotImplementedError

    def new_architecture(self, arch_id, **kw):
        raise NotImplementedError

    # ---- generic helpers built on the three methods above -----------------
    def __len__(self):
        return int(np.prod([len(d) for d in self.decisions()]))

    def random_architecture_id(self, rng):
        return self.decode([rng.randint(len(d)) for d in self.decisions()])

    def sample_all_models(self):
        for choice in itertools.product(*[range(len(d))
                                          for d in self.decisions()]):
            yield self.decode(list(choice))

    def mutate_architecture(self, arch_id, rng):
        """Change exactly one decision to a different option."""
        choice = self.encode(arch_id)
        open_slots = [i for i, d in enumerate(self.decisions()) if len(d) > 1]
        i = open_slots[rng.randint(len(open_slots))]
        options = [o for o in range(len(self.decisions()[i]))
                   if o != choice[i]]
        choice[i] = options[rng.randint(len(options))]
        return self.decode(choice)


# --------------------------------------------------------------------- MLP
class DNNModel(model.Model):
    """Embedding of the (id, value) features, then the hidden layers an
    architecture id names, then the class logits."""

    def __init__(self, nfield, nfeat, nemb, hidden, num_classes=2):
        super(DNNModel, self).__init__()
        self.nfield, self.nemb = nfield, nemb
        self.embedding = layer.Embedding(nfeat, nemb)
        self.hidden = [layer.Linear(h) for h in hidden]
        self.acts = [layer.ReLU() for _ in hidden]
        for i, (l, a) in enumerate(zip(self.hidden, self.acts)):
            setattr(self, "fc%d" % i, l)
            setattr(self, "act%d" % i, a)
        self.out = layer.Linear(num_classes)
        self.loss_fn = layer.SoftMaxCrossEntropy()
        self.taps = None      # filled with the hidden activations on request

    def forward(self, ids, values):
        e = self.embedding(ids)                          # (B, F, E)
        B = ids.shape[0]
        v = autograd.reshape(values, (B, self.nfield, 1))
        h = autograd.reshape(autograd.mul(e, v), (B, self.nfield * self.nemb))
        taps = []
        for l, a in zip(self.hidden, self.acts):
            h = a(l(h))
            taps.append(h)
        if self.taps is not None:
            self.taps = taps
        return self.out(h)

    def train_one_batch(self, ids, values, y):
        out = self.forward(ids, values)
        loss = self.loss_fn(out, y)
        self.optimizer(loss)
        return out, loss


class MlpSpace(SpaceWrapper):
    name = "mlp_sp"

    def __init__(self, nfield, nfeat, nemb=8, num_layers=3,
                 layer_choices=(8, 16, 32, 64, 128), num_classes=2):
        self.nfield, self.nfeat, self.nemb = nfield, nfeat, nemb
        self.num_layers = num_layers
        self.layer_choices = tuple(layer_choices)
        self.num_classes = num_classes

    def decisions(self):
        return [list(self.layer_choices)] * self.num_layers

    def encode(self, arch_id):
        return [self.layer_choices.index(int(t)) for t in arch_id.split("-")]

    def decode(self, choice):
        return "-".join(str(self.layer_choices[c]) for c in choice)

    def hidden_sizes(self, arch_id):
        return [int(t) for t in arch_id.split("-")]

    def new_architecture(self, arch_id, **kw):
        return DNNModel(self.nfield, self.nfeat, self.nemb,
                        self.hidden_sizes(arch_id), self.num_classes)

    def inputs(self, dev, ids, values):
        return [tensor.from_numpy(np.ascontiguousarray(ids, np.int32), dev),
                tensor.from_numpy(np.ascontiguousarray(values, np.float32),
                                  dev)]

    def profiling(self, arch_id):
        """(parameters, multiply-adds per example) of an architecture."""
        sizes = [self.nfield * self.nemb] + self.hidden_sizes(arch_id) + \
            [self.num_classes]
        macs = sum(a * b for a, b in zip(sizes, sizes[1:]))
        return self.nfeat * self.nemb + macs + sum(sizes[1:]), macs


# --------------------------------------------------------------------- CNN
class StageCNN(model.Model):
    """Stages of 3x3 convolution + batch norm + ReLU, each closed by 2x2 max
    pooling, then global pooling and the classifier."""

    def __init__(self, stages, num_classes=10):
        super(StageCNN, self).__init__()
        self.blocks = []
        n = 0
        for depth, width in stages:
            for _ in range(depth):
                conv = layer.Conv2d(width, 3, padding=1, bias=False)
                bn, act = layer.BatchNorm2d(), layer.ReLU()
                for kind, l in (("conv", conv), ("bn", bn), ("act", act)):
                    setattr(self, "%s%d" % (kind, n), l)
                self.blocks.append((conv, bn, act, False))
                n += 1
            pool = layer.MaxPool2d(2, 2)
            setattr(self, "pool%d" % n, pool)
            self.blocks.append((pool, None, None, True))
        self.gap = layer.AvgPool2d(1)   # replaced at first forward
        self.flat = layer.Flatten()
        self.out = layer.Linear(num_classes)
        self.loss_fn = layer.SoftMaxCrossEntropy()

    def forward(self, x):
        h = x
        for a, b, c, is_pool in self.blocks:
            h = a(h) if is_pool else c(b(a(h)))
        h = autograd.reduce_mean(h, [2, 3], 0)
        return self.out(self.flat(h))

    def train_one_batch(self, x, y):
        out = self.forward(x)
        loss = self.loss_fn(out, y)
        self.optimizer(loss)
        return out, loss


class CnnSpace(SpaceWrapper):
    """Per stage a depth and a width: ids like ``"1x16-2x32-2x64"``."""
    name = "cnn_sp"

    def __init__(self, num_stages=3, depths=(1, 2), widths=(8, 16, 32, 64),
                 num_classes=10):
        self.num_stages, self.depths = num_stages, tuple(depths)
        self.widths, self.num_classes = tuple(widths), num_classes

    def decisions(self):
        return [list(self.depths), list(self.widths)] * self.num_stages

    def encode(self, arch_id):
        out = []
        for tok in arch_id.split("-"):
            d, w = tok.split("x")
            out += [self.depths.index(int(d)), self.widths.index(int(w))]
        return out

    def decode(self, choice):
        return "-".join("%dx%d" % (self.depths[choice[2 * i]],
                                   self.widths[choice[2 * i + 1]])
                        for i in range(self.num_stages))

    def new_architecture(self, arch_id, **kw):
        stages = [tuple(int(t) for t in tok.split("x"))
                  for tok in arch_id.split("-")]
        return StageCNN(stages, self.num_classes)

    def inputs(self, dev, x):
        return [tensor.from_numpy(np.ascontiguousarray(x, np.float32), dev)]
