"""Data and logging helpers of the GAN examples."""
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]


def load_data(filepath=None):
    """-> train images (N, 784) float32 in [0, 1].  ``filepath`` may be an
    ``.npz`` with array ``x``; otherwise the shared MNIST loader is used (which
    falls back to synthetic digits-like data when the files are absent)."""
    if filepath and filepath.endswith(".npz") and os.path.exists(filepath):
        x = np.load(filepath, allow_pickle=False)["x"].astype(np.float32)
        return x.reshape(x.shape[0], -1)
    import data as datasets
    tx, _, _, _ = datasets.load("mnist")
    tx = tx.reshape(tx.shape[0], -1)
    lo, hi = tx.min(), tx.max()
    return ((tx - lo) / max(hi - lo, 1e-6)).astype(np.float32)


def print_log(s):
    print("[%s] %s" % (time.strftime("%Y-%m-%d %H:%M:%S"), s), flush=True)


def save_grid(images, rows, cols, path):
    """Tile (rows*cols, H, W) images in [0, 1] into one grayscale PNG."""
    from PIL import Image
    n, h, w = images.shape
    grid = images[:rows * cols].reshape(rows, cols, h, w) \
        .transpose(0, 2, 1, 3).reshape(rows * h, cols * w)
    Image.fromarray((np.clip(grid, 0, 1) * 255).astype(np.uint8),
                    "L").save(path)
