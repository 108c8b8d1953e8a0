"""LIME explanation of an image classifier: which regions of an image push the
model towards a class?  The image is cut into a grid of segments, random
subsets of segments are greyed out, the model scores every perturbed copy, and
a proximity-weighted linear model fitted to (mask -> score) ranks the segments.
Counterpart of the reference's examples/singa_easy (explanations/lime).

    python lime.py            # trains a small CNN on synthetic data, explains it
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", "cnn"), os.path.join(_HERE, ".."),
                os.path.join(_HERE, "..", "..")]

from singa_amd import tensor  # noqa: E402


def grid_segments(h, w, cells):
    """Label map (h, w) of a cells x cells grid."""
    ys = np.minimum(np.arange(h) * cells // h, cells - 1)
    xs = np.minimum(np.arange(w) * cells // w, cells - 1)
    return ys[:, None] * cells + xs[None, :]


def explain(predict, image, label, cells=4, samples=200, baseline=0.0,
            kernel_width=0.25, rng=None):
    """-> weights (cells*cells,): contribution of each segment to ``label``.

    predict: callable((N,C,H,W) float32) -> (N, classes) probabilities.
    """
    rng = rng or np.random.RandomState(0)
    c, h, w = image.shape
    seg = grid_segments(h, w, cells)
    nseg = cells * cells
    masks = rng.randint(0, 2, (samples, nseg)).astype(np.float32)
    masks[0] = 1.0                                   # the unperturbed image
    keep = masks[:, seg]                             # (samples, h, w)
    batch = image[None] * keep[:, None] + baseline * (1 - keep[:, None])
    scores = predict(batch.astype(np.float32))[:, label]
    # proximity kernel on the cosine distance to the full image
    dist = 1.0 - masks.sum(1) / np.sqrt(nseg * np.maximum(masks.sum(1), 1))
    weights = np.exp(-(dist ** 2) / kernel_width ** 2)
    X = np.concatenate([masks, np.ones((samples, 1), np.float32)], 1)
    sw = np.sqrt(weights)[:, None]
    coef, *_ = np.linalg.lstsq(X * sw, scores * sw[:, 0], rcond=None)
    return coef[:nseg]


def model_predictor(model, dev, batch_size=32):
    """Wrap a compiled singa_amd Model as a numpy probability function."""
    def predict(x):
        outs = []
        for i in range(0, len(x), batch_size):
            chunk = x[i:i + batch_size]
            logits = tensor.to_numpy(model(tensor.from_numpy(chunk, dev)))
            e = np.exp(logits - logits.max(1, keepdims=True))
            outs.append(e / e.sum(1, keepdims=True))
        return np.concatenate(outs)
    return predict


def demo():
    """Class k lights up quadrant k of a 2x2 layout: a faithful explanation
    must rank that quadrant's segments first."""
    from singa_amd import models, opt
    import train_cnn as base
    rng = np.random.RandomState(0)

    def make(n):
        y = rng.randint(0, 4, n).astype(np.int32)
        x = 0.1 * rng.randn(n, 1, 28, 28).astype(np.float32)
        for i, k in enumerate(y):
            r, c = divmod(int(k), 2)
            x[i, 0, r * 14:(r + 1) * 14, c * 14:(c + 1) * 14] += 1.0
        return x, y

    dev = base.create_device(0)
    tx_, ty_ = make(512)
    model = models.create("cnn", num_channels=1, num_classes=4)
    model.set_optimizer(opt.SGD(lr=0.01, momentum=0.9))
    tx = tensor.Tensor((32, 1, 28, 28), dev)
    ty = tensor.Tensor((32,), dev, tensor.int32)
    model.compile([tx], is_train=True, use_graph=False, sequential=False)
    model.train()
    for epoch in range(3):
        for b in range(16):
            tx.copy_from_numpy(tx_[b * 32:(b + 1) * 32])
            ty.copy_from_numpy(ty_[b * 32:(b + 1) * 32])
            model(tx, ty, "plain", None)
    model.eval()
    x, y = make(1)
    w = explain(model_predictor(model, dev), x[0], int(y[0]), cells=2,
                samples=64)
    print("label", int(y[0]), "segment weights", np.round(w, 3))
    return int(y[0]), w


if __name__ == "__main__":
    demo()
