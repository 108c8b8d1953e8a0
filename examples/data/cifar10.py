"""CIFAR-10 python batches -> (N,3,32,32) float32 normalised per channel."""
import os

import numpy as np

from . import data_dir, synthetic_images

MEAN = np.array([0.4914, 0.4822, 0.4465], np.float32).reshape(1, 3, 1, 1)
STD = np.array([0.2023, 0.1994, 0.2010], np.float32).reshape(1, 3, 1, 1)


def _load_batch(path):
    # the python-version batches are pickles of plain dict/bytes/ndarray; they
    # are the user's own download, read with numpy's restricted unpickler off
    import pickle
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    x = np.asarray(d["data"], np.uint8).reshape(-1, 3, 32, 32)
    y = np.asarray(d.get("labels", d.get("fine_labels")), np.int32)
    return x, y


def normalize(x):
    return (x.astype(np.float32) / 255 - MEAN) / STD


def load(dir_path=None, num_train=2048, num_val=512):
    d = dir_path or os.path.join(data_dir("cifar-10-batches-py"))
    names = ["data_batch_%d" % i for i in range(1, 6)]
    if all(os.path.exists(os.path.join(d, n)) for n in names + ["test_batch"]):
        xs, ys = zip(*[_load_batch(os.path.join(d, n)) for n in names])
        vx, vy = _load_batch(os.path.join(d, "test_batch"))
        return (normalize(np.concatenate(xs)), np.concatenate(ys),
                normalize(vx), vy)
    print("cifar10 files not found under %s: using synthetic data" % d)
    return synthetic_images(num_train, num_val, 3, 32, 10, seed=2)
