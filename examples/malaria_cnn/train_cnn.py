"""Malaria cell-image classification (parasitized / uninfected).

    python train_cnn.py cnn malaria -dir /data/malaria -m 10
    python train_cnn.py mlp malaria --synthetic -m 2

Command line of the reference's examples/malaria_cnn (README.md:39-41,
run.sh:20).  The dataset (``training_set`` / ``testing_set`` folders of cell
thumbnails) is decoded once into a uint8 array that stays in host memory; each
step standardises and flips only its own batch, and uploads it while the
previous step's kernels are still running (the step is a replayed graph, so
the host is idle in between).  ``--synthetic`` writes a small stand-in dataset
of drawn cells first, for machines without the download.
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402
from data import malaria  # noqa: E402
from model import cnn, mlp  # noqa: E402


def create_device(device_id=0):
    if device.get_num_gpus() > 0:
        return device.create_cuda_gpu_on(device_id)
    return device.get_default_device()


def accuracy(pred, target):
    return int(np.sum(np.argmax(pred, axis=1) == target))


def batch_of(images, labels, sel, flat, train, rng):
    """Standardised float32 batch of the uint8 images at ``sel``; training
    batches are mirrored left-right / up-down at random (a blood smear has
    no preferred orientation)."""
    x = images[sel]
    if train:
        lr, ud = rng.rand(len(sel)) < 0.5, rng.rand(len(sel)) < 0.5
        x = np.where(lr[:, None, None, None], x[..., ::-1], x)
        x = np.where(ud[:, None, None, None], x[..., ::-1, :], x)
    x = malaria.normalize(x)
    return (x.reshape(len(sel), -1) if flat else x), labels[sel]


def run(args):
    dev = create_device(args.device_id)
    dev.SetRandSeed(0)
    rng = np.random.RandomState(0)
    size = (args.image_size, args.image_size)

    root = args.dir
    if args.synthetic:
        root = malaria.write_synthetic(
            tempfile.mkdtemp(prefix="malaria_"), args.synthetic_train,
            args.synthetic_train // 4)
    train_x, train_y = malaria.load_train_data(root, size)
    val_x, val_y = malaria.load_test_data(root, size)
    print("malaria: %d training and %d test images of %dx%d" %
          (len(train_x), len(val_x), size[0], size[1]), flush=True)

    flat = args.model == "mlp"
    if flat:
        net = mlp.create_model(data_size=3 * size[0] * size[1], num_classes=2)
        tx = tensor.Tensor((args.batch_size, 3 * size[0] * size[1]), dev)
    else:
        net = cnn.create_model(num_channels=3, num_classes=2,
                               input_size=size[0])
        tx = tensor.Tensor((args.batch_size, 3) + size, dev)
    ty = tensor.Tensor((args.batch_size,), dev, tensor.int32)

    net.set_optimizer(opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5))
    net.compile([tx], is_train=True, use_graph=args.graph, sequential=False)
    dev.SetVerbosity(args.verbosity)

    bs = args.batch_size
    n_train, n_val = len(train_x) // bs, len(val_x) // bs
    if args.max_batches:
        n_train, n_val = min(n_train, args.max_batches), \
            min(n_val, args.max_batches)
    history = []
    for epoch in range(args.max_epoch):
        start = time.time()
        order = rng.permutation(len(train_x))
        net.train()
        correct, loss_sum, pending = 0, 0.0, None
        for b in range(n_train):
            x, y = batch_of(train_x, train_y, np.sort(order[b * bs:(b + 1) * bs]),
                            flat, True, rng)
            # reading the previous step's results is the only point where
            # the host waits for the device; the next batch is ready by then
            if pending is not None:
                correct += accuracy(tensor.to_numpy(pending[0]), pending[2])
                loss_sum += float(tensor.to_numpy(pending[1]).reshape(-1)[0])
            tx.copy_from_numpy(x)
            ty.copy_from_numpy(y)
            out, loss = net(tx, ty)
            pending = (out, loss, y)
        if pending is not None:
            correct += accuracy(tensor.to_numpy(pending[0]), pending[2])
            loss_sum += float(tensor.to_numpy(pending[1]).reshape(-1)[0])

        net.eval()
        val_correct = 0
        for b in range(n_val):
            sel = np.arange(b * bs, (b + 1) * bs)
            x, y = batch_of(val_x, val_y, sel, flat, False, rng)
            tx.copy_from_numpy(x)
            val_correct += accuracy(tensor.to_numpy(net(tx)), y)
        stats = (loss_sum / max(n_train, 1), correct / max(n_train * bs, 1),
                 val_correct / max(n_val * bs, 1))
        history.append(stats)
        print("Epoch %d: training loss = %f, training accuracy = %f, "
              "evaluation accuracy = %f, elapsed %.2fs" %
              ((epoch,) + stats + (time.time() - start,)), flush=True)
    dev.PrintTimeProfiling()
    return history


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("model", nargs="?", default="cnn", choices=["cnn", "mlp"])
    p.add_argument("data", nargs="?", default="malaria", choices=["malaria"])
    p.add_argument("-dir", "--dir-path", dest="dir", default="/tmp/malaria")
    p.add_argument("--synthetic", action="store_true",
                   help="draw a stand-in dataset instead of reading -dir")
    p.add_argument("--synthetic-train", type=int, default=256)
    p.add_argument("--image-size", type=int, default=128)
    p.add_argument("-m", "--max-epoch", default=10, type=int, dest="max_epoch")
    p.add_argument("-b", "--batch-size", default=64, type=int,
                   dest="batch_size")
    p.add_argument("-l", "--learning-rate", default=0.005, type=float,
                   dest="lr")
    p.add_argument("-i", "--device-id", default=0, type=int, dest="device_id")
    p.add_argument("-g", "--disable-graph", default=True,
                   action="store_false", dest="graph")
    p.add_argument("-v", "--log-verbosity", default=0, type=int,
                   dest="verbosity")
    p.add_argument("--max-batches", default=None, type=int)
    return p.parse_args(argv)


if __name__ == "__main__":
    run(parse())
