"""BloodMNIST demo: 8-class blood-cell classification on 28x28 RGB images.
Reads ``bloodmnist.npz`` (MedMNIST layout: train_images (N,28,28,3) uint8,
train_labels (N,1), val_*, test_*) from ``$SINGA_DATA``; a synthetic stand-in of
the same shapes is used when the file is absent.  Counterpart of the reference's
examples/demos/Classification/BloodMnist/ClassDemo.py.

    python ClassDemo.py -m 5 -b 64
"""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_EX = os.path.join(_HERE, "..", "..", "..")
sys.path[:0] = [os.path.join(_EX, "cnn"), _EX, os.path.join(_EX, "..")]

from singa_amd import models, opt, tensor  # noqa: E402
import data as datasets  # noqa: E402
import train_cnn as base  # noqa: E402


def load():
    path = os.path.join(datasets.data_dir(""), "bloodmnist.npz")
    if os.path.isfile(path):
        z = np.load(path, allow_pickle=False)

        def prep(x, y):
            x = x.astype(np.float32).transpose(0, 3, 1, 2) / 255.0 - 0.5
            return x, y.reshape(-1).astype(np.int32)

        tx, ty = prep(z["train_images"], z["train_labels"])
        vx, vy = prep(z["val_images"], z["val_labels"])
        return tx, ty, vx, vy
    print("bloodmnist.npz not found: using synthetic data")
    return datasets.synthetic_images(2048, 512, 3, 28, 8, seed=5)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--max-epoch", type=int, default=5)
    ap.add_argument("-b", "--batch-size", type=int, default=64)
    ap.add_argument("-l", "--lr", type=float, default=0.01)
    ap.add_argument("--max-batches", type=int, default=None)
    args = ap.parse_args(argv)
    tx_, ty_, vx_, vy_ = load()
    dev = base.create_device(0)
    bs = args.batch_size
    model = models.create("cnn", num_channels=3, num_classes=8)
    model.set_optimizer(opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5))
    tx = tensor.Tensor((bs, 3, 28, 28), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    model.compile([tx], is_train=True, use_graph=True, sequential=False)
    nb, nvb = len(tx_) // bs, len(vx_) // bs
    if args.max_batches:
        nb, nvb = min(nb, args.max_batches), min(nvb, args.max_batches)
    acc = 0.0
    for epoch in range(args.max_epoch):
        idx = np.random.permutation(len(tx_))
        model.train()
        loss_sum = 0.0
        for b in range(nb):
            sel = idx[b * bs:(b + 1) * bs]
            tx.copy_from_numpy(base.augmentation(tx_[sel]))
            ty.copy_from_numpy(ty_[sel])
            _, loss = model(tx, ty, "plain", None)
            loss_sum += float(tensor.to_numpy(loss).reshape(-1)[0])
        model.eval()
        correct = 0
        for b in range(nvb):
            tx.copy_from_numpy(vx_[b * bs:(b + 1) * bs])
            correct += base.accuracy(tensor.to_numpy(model(tx)),
                                     vy_[b * bs:(b + 1) * bs])
        acc = correct / max(nvb * bs, 1)
        print("epoch %d: loss %.4f val acc %.3f" % (epoch, loss_sum / nb, acc),
              flush=True)
    return acc


if __name__ == "__main__":
    main()
