"""Structured (tabular) data in libsvm coding: every row is
``label id:value id:value ...`` with exactly ``nfield`` pairs whose ids index a
shared vocabulary of ``nfeat`` features.  Role: reference
src/dataset_utils/structure_data_loader.py (libsvm readers, batching) and
stream_dataloader.py (mini-batches decoded from database rows)."""
import json

import numpy as np


class TabularData(object):
    """ids (n, nfield) int32, values (n, nfield) float32, labels (n,) int32."""

    def __init__(self, ids, values, labels, nfeat):
        self.ids = np.ascontiguousarray(ids, np.int32)
        self.values = np.ascontiguousarray(values, np.float32)
        self.labels = np.ascontiguousarray(labels, np.int32)
        self.nfeat = int(nfeat)

    @property
    def nfield(self):
        return self.ids.shape[1]

    def __len__(self):
        return len(self.labels)

    def split(self, fraction, seed=0):
        order = np.random.RandomState(seed).permutation(len(self))
        cut = int(len(self) * fraction)
        a, b = order[:cut], order[cut:]
        return (TabularData(self.ids[a], self.values[a], self.labels[a],
                            self.nfeat),
                TabularData(self.ids[b], self.values[b], self.labels[b],
                            self.nfeat))

    def batches(self, batch_size, shuffle=False, seed=0, drop_last=True):
        order = np.arange(len(self))
        if shuffle:
            np.random.RandomState(seed).shuffle(order)
        stop = len(self) - (len(self) % batch_size if drop_last else 0)
        for i in range(0, stop, batch_size):
            j = order[i:i + batch_size]
            if len(j):
                yield self.ids[j], self.values[j], self.labels[j]


def parse_libsvm_lines(lines, nfield=None, nfeat=None):
    ids, vals, labels = [], [], []
    for line in lines:
        parts = line.split()
        if not parts:
            continue
        labels.append(int(float(parts[0])))
        pairs = [p.split(":") for p in parts[1:]]
        ids.append([int(a) for a, _ in pairs])
        vals.append([float(b) for _, b in pairs])
    width = nfield or max(len(r) for r in ids)
    ids_a = np.zeros((len(ids), width), np.int32)
    val_a = np.zeros((len(ids), width), np.float32)   # missing pairs: value 0
    for r, (i, v) in enumerate(zip(ids, vals)):
        ids_a[r, :len(i)] = i[:width]
        val_a[r, :len(v)] = v[:width]
    return TabularData(ids_a, val_a, labels,
                       nfeat or (int(ids_a.max()) + 1 if len(ids) else 1))


def load_libsvm(path, nfield=None, nfeat=None):
    with open(path) as f:
        return parse_libsvm_lines(f, nfield, nfeat)


def save_libsvm(data, path):
    with open(path, "w") as f:
        for i, v, y in zip(data.ids, data.values, data.labels):
            f.write("%d %s\n" % (y, " ".join("%d:%g" % p for p in zip(i, v))))


def from_mini_batch_json(text, nfield=None, nfeat=None):
    """The mini-batch format the database functions pass: a JSON list of
    rows, each ``{"label": 1, "col1": "12:1", ...}`` (libsvm pairs as strings)
    or a JSON list of libsvm lines."""
    rows = json.loads(text) if isinstance(text, str) else text
    lines = []
    for r in rows:
        if isinstance(r, str):
            lines.append(r)
            continue
        label = r.get("label", 0)
        pairs = [str(v) for k, v in r.items() if k not in ("label", "id")]
        lines.append("%s %s" % (label, " ".join(pairs)))
    return parse_libsvm_lines(lines, nfield, nfeat)


def synthetic(n=2048, nfield=6, vocab_per_field=12, seed=0, noise=0.1):
    """A binary task whose label depends on pairwise interactions of a few
    fields, so that wider / deeper networks do fit it better."""
    rs = np.random.RandomState(seed)
    local = rs.randint(0, vocab_per_field, (n, nfield))
    ids = local + np.arange(nfield) * vocab_per_field
    emb = rs.randn(nfield * vocab_per_field, 4)
    e = emb[ids]                                     # (n, nfield, 4)
    score = (e[:, 0] * e[:, 1]).sum(1) + (e[:, 2] * e[:, 3]).sum(1) \
        + 0.5 * e[:, 4:, 0].sum(1)
    score += noise * rs.randn(n)
    labels = (score > np.median(score)).astype(np.int32)
    return TabularData(ids, np.ones((n, nfield), np.float32), labels,
                       nfield * vocab_per_field)
