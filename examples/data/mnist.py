"""MNIST: idx-ubyte(.gz) files -> (N,1,28,28) float32 in [0,1], int32 labels.
Layout as produced by the reference's examples/cnn/data/download_mnist.py."""
import gzip
import os

import numpy as np

from . import data_dir, synthetic_images

_FILES = {
    "train_x": "train-images-idx3-ubyte", "train_y": "train-labels-idx1-ubyte",
    "val_x": "t10k-images-idx3-ubyte", "val_y": "t10k-labels-idx1-ubyte",
}


def _read(path, offset):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        return np.frombuffer(f.read(), np.uint8, offset=offset)


def _find(d, stem):
    for cand in (stem, stem + ".gz"):
        p = os.path.join(d, cand)
        if os.path.exists(p):
            return p
    return None


def load(dir_path=None, num_train=2048, num_val=512):
    d = dir_path or data_dir("mnist")
    paths = {k: _find(d, v) for k, v in _FILES.items()}
    if all(paths.values()):
        tx = _read(paths["train_x"], 16).reshape(-1, 1, 28, 28)
        vx = _read(paths["val_x"], 16).reshape(-1, 1, 28, 28)
        ty = _read(paths["train_y"], 8).astype(np.int32)
        vy = _read(paths["val_y"], 8).astype(np.int32)
        return (tx.astype(np.float32) / 255, ty,
                vx.astype(np.float32) / 255, vy)
    print("mnist files not found under %s: using synthetic data" % d)
    return synthetic_images(num_train, num_val, 1, 28, 10, seed=1)
