"""A predicate-conditioned mixture of experts for tabular data.

A query ``SELECT ... WHERE col_a = v AND col_c = w`` names a sub-population.
The predicate is encoded as one id per column (the value's feature id, or the
column's wildcard id when the column is not constrained); a gate network maps
its embedding to a sparse distribution over K expert MLPs (all but the
``top`` largest gate weights are dropped and the rest renormalised), and the
prediction is the gate-weighted sum of the experts' logits.  For one
predicate the gate is computed once, so serving a query runs only the
selected experts: the model is *sliced* by the query.  Role: the reference's
model_slicing (sparse-attention mixture served from SQL through
model_slicing/pg_interface.py).
"""
import os
import sys

import numpy as np

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..", "..")]

from singa_amd import autograd, layer, model, opt, tensor  # noqa: E402


class SlicedMoE(model.Model):

    def __init__(self, nfield, nfeat, nemb=8, experts=4, hidden=32, top=2,
                 num_classes=2):
        super(SlicedMoE, self).__init__()
        self.nfield, self.nfeat, self.nemb = nfield, nfeat, nemb
        self.K, self.top = experts, min(top, experts)
        # feature ids [0, nfeat) and one wildcard id per column after them
        self.embedding = layer.Embedding(nfeat + nfield, nemb)
        self.gate = layer.Linear(experts)
        self.fc1 = [layer.Linear(hidden) for _ in range(experts)]
        self.fc2 = [layer.Linear(num_classes) for _ in range(experts)]
        for i, (a, b) in enumerate(zip(self.fc1, self.fc2)):
            setattr(self, "expert%d_fc1" % i, a)
            setattr(self, "expert%d_fc2" % i, b)
        self.relu = layer.ReLU()
        self.loss_fn = layer.SoftMaxCrossEntropy()
        self._built = False    # every expert has seen one forward pass

    def wildcard(self, column):
        return self.nfeat + column

    def _flat(self, ids):
        B = ids.shape[0]
        return autograd.reshape(self.embedding(ids),
                                (B, self.nfield * self.nemb))

    def gate_weights(self, pred_ids):
        """(B, K) sparse gate of the predicate ids (B, nfield)."""
        logits = self.gate(self._flat(pred_ids))
        g = np.nan_to_num(tensor.to_numpy(autograd.softmax(logits, 1)),
                          nan=1.0 / self.K)     # the shape-inference pass runs
        # on parameters that are not initialised yet
        if self.top < self.K:
            drop = np.argsort(g, axis=1, kind="stable")[:, :self.K - self.top]
            np.put_along_axis(g, drop, 0.0, axis=1)
            total = g.sum(axis=1, keepdims=True)
            g = g / np.where(total > 0, total, 1.0)
        return g.astype(np.float32)

    def forward(self, ids, values, pred_ids):
        B = ids.shape[0]
        v = autograd.reshape(values, (B, self.nfield, 1))
        h = autograd.reshape(autograd.mul(self.embedding(ids), v),
                             (B, self.nfield * self.nemb))
        # the sparse gate is a constant of the step (its argmax set is not
        # differentiable); the dense softmax keeps the gate trainable
        dense = autograd.softmax(self.gate(self._flat(pred_ids)), 1)
        mask = (self.gate_weights(pred_ids) > 0).astype(np.float32)
        gate = autograd.mul(dense, tensor.from_numpy(mask, ids.device))
        out = None
        for k in range(self.K):
            if self._built and not mask[:, k].any():
                continue                 # an expert nobody selected is skipped
            # (the first pass visits all of them: it creates their parameters)
            y = self.fc2[k](self.relu(self.fc1[k](h)))
            col = tensor.from_numpy(np.eye(self.K, dtype=np.float32)[:, k:k + 1],
                                    ids.device)
            w = autograd.matmul(gate, col)                       # (B, 1)
            y = autograd.mul(y, w)
            out = y if out is None else autograd.add(out, y)
        self._built = True
        return out

    def train_one_batch(self, ids, values, pred_ids, y):
        out = self.forward(ids, values, pred_ids)
        loss = self.loss_fn(out, y)
        self.optimizer(loss)
        return out, loss

    def set_optimizer(self, o):
        self.optimizer = o


def predicate_ids(m, where):
    """{column index: feature id} -> (nfield,) ids with wildcards."""
    ids = np.array([m.wildcard(c) for c in range(m.nfield)], np.int32)
    for c, fid in where.items():
        ids[int(c)] = int(fid)
    return ids


def train(m, dev, data, epochs=2, batch_size=64, lr=0.05, seed=0):
    """Train on rows paired with random predicates that each row satisfies
    (a random subset of its own columns is constrained)."""
    rng = np.random.RandomState(seed)
    ids0 = np.zeros((batch_size, m.nfield), np.int32)
    tx = [tensor.from_numpy(ids0, dev),
          tensor.from_numpy(np.zeros((batch_size, m.nfield), np.float32), dev),
          tensor.from_numpy(ids0, dev)]
    m.set_optimizer(opt.SGD(lr=lr, momentum=0.9))
    m.compile(tx, is_train=True, use_graph=False, sequential=False)
    m.train()
    ty = tensor.Tensor((batch_size,), dev, tensor.int32)
    losses = []
    for ep in range(epochs):
        for ids, values, y in data.batches(batch_size, shuffle=True,
                                           seed=seed + ep):
            keep = rng.rand(*ids.shape) < 0.3
            wild = np.arange(m.nfield, dtype=np.int32)[None, :] + m.nfeat
            pred = np.where(keep, ids, wild).astype(np.int32)
            tx[0].copy_from_numpy(ids.astype(np.int32))
            tx[1].copy_from_numpy(values.astype(np.float32))
            tx[2].copy_from_numpy(pred)
            ty.copy_from_numpy(y.astype(np.int32))
            _, loss = m(tx[0], tx[1], tx[2], ty)
            losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    m.eval()
    return losses
