"""Dataset loaders shared by the example scripts.

Each loader reads the dataset's standard files from ``$SINGA_DATA/<name>``
(default ``/tmp/<name>``) when they are there.  The examples must also run on
machines without network access, so when the files are absent the loader
returns a deterministic synthetic stand-in of the same shapes and dtypes
(class-conditional patterns plus noise, so that training still converges and
accuracy numbers are meaningful as a smoke signal, not as benchmarks).
"""
import os

import numpy as np


def data_dir(name):
    return os.path.join(os.environ.get("SINGA_DATA", "/tmp"), name)


def synthetic_images(num_train, num_val, channels, size, num_classes, seed=0):
    """Images whose class decides a low-frequency template; N(0, 0.5) noise."""
    rng = np.random.RandomState(seed)
    templates = rng.randn(num_classes, channels, 4, 4).astype(np.float32)
    templates = np.kron(templates, np.ones((size // 4, size // 4),
                                           np.float32))

    def make(n):
        y = rng.randint(0, num_classes, n).astype(np.int32)
        x = templates[y] + 0.5 * rng.randn(n, channels, size,
                                           size).astype(np.float32)
        return x, y

    tx, ty = make(num_train)
    vx, vy = make(num_val)
    return tx, ty, vx, vy


def load(name, **kw):
    from . import cifar10, cifar100, mnist
    return {"mnist": mnist, "cifar10": cifar10,
            "cifar100": cifar100}[name].load(**kw)
