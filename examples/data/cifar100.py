"""CIFAR-100 python version -> (N,3,32,32) float32, fine labels."""
import os

import numpy as np

from . import data_dir, synthetic_images
from .cifar10 import _load_batch

MEAN = np.array([0.5071, 0.4865, 0.4409], np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.2673, 0.2564, 0.2762], np.float32).reshape(1, 3, 1, 1)


def load(dir_path=None, num_train=4096, num_val=512):
    d = dir_path or data_dir("cifar-100-python")
    tr, te = os.path.join(d, "train"), os.path.join(d, "test")
    if os.path.exists(tr) and os.path.exists(te):
        tx, ty = _load_batch(tr)
        vx, vy = _load_batch(te)
        f = lambda x: (x.astype(np.float32) / 255 - MEAN) / STD  # noqa: E731
        return f(tx), ty, f(vx), vy
    print("cifar100 files not found under %s: using synthetic data" % d)
    return synthetic_images(num_train, num_val, 3, 32, 100, seed=3)
