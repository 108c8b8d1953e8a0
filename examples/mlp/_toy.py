"""2-d toy problem shared by the MLP examples: points scattered around the
line y = 5x + 1, labelled by the side of the line they fall on."""
import numpy as np


def make(n=400, seed=0):
    rng = np.random.RandomState(seed)
    x = rng.uniform(-1, 1, n)
    y = 5 * x + 1 + 2 * rng.randn(n)
    data = np.stack([x, y], axis=1).astype(np.float32)
    label = (5 * x + 1 > y).astype(np.int32)
    return data, label
