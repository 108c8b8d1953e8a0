"""Tabular binary-classification data for the federated example, split among
clients.  Reads ``$SINGA_DATA/bank/bank_<rank>.csv`` style shards (numeric
features, last column label) when present; otherwise generates a linearly
separable-with-noise problem and gives each client a disjoint shard."""
import os

import numpy as np


def load(rank, num_clients=2, n=4000, dim=16, seed=7):
    path = os.path.join(os.environ.get("SINGA_DATA", "/tmp"), "bank",
                        "bank_%d.csv" % rank)
    if os.path.isfile(path):
        a = np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.float32)
        x, y = a[:, :-1], a[:, -1].astype(np.int32)
    else:
        rng = np.random.RandomState(seed)
        w = rng.randn(dim).astype(np.float32)
        x = rng.randn(n, dim).astype(np.float32)
        y = ((x @ w + 0.3 * rng.randn(n)) > 0).astype(np.int32)
        per = n // num_clients
        x, y = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    cut = int(0.8 * len(x))
    return x[:cut], y[:cut], x[cut:], y[cut:]
