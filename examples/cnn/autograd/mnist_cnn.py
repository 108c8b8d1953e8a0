"""MNIST CNN written against raw autograd operators and Layer objects, without
the Model class or graph buffering; ``--dist`` wraps the optimizer in DistOpt
and ``--dist-option`` selects plain / half / sparseTopK / sparseThreshold /
partialUpdate gradient exchange.  Counterpart of the reference's
examples/cnn/autograd/{mnist_cnn,mnist_dist,sparsification_mnist}.py.

    python mnist_cnn.py -m 2
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \\
        mnist_cnn.py --dist --dist-option sparseTopK
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", ".."),
                os.path.join(_HERE, "..", "..", "..")]

from singa_amd import autograd, device, layer, opt, tensor  # noqa: E402
import data as datasets  # noqa: E402


class CNN(object):

    def __init__(self):
        self.conv1 = layer.Conv2d(32, 3, padding=1)
        self.conv2 = layer.Conv2d(32, 3, padding=1)
        self.linear1 = layer.Linear(512)
        self.linear2 = layer.Linear(10)
        self.pooling1 = layer.MaxPool2d(2, 2, padding=0)
        self.pooling2 = layer.MaxPool2d(2, 2, padding=0)
        self.relu = layer.ReLU()
        self.flatten = layer.Flatten()

    def forward(self, x):
        y = self.pooling1(self.relu(self.conv1(x)))
        y = self.pooling2(self.relu(self.conv2(y)))
        y = self.relu(self.linear1(self.flatten(y)))
        return self.linear2(y)


def update(sgd, loss, dist_option, spars):
    if not hasattr(sgd, "communicator") or dist_option == "plain":
        sgd(loss)
    elif dist_option == "half":
        sgd.backward_and_update_half(loss)
    elif dist_option == "partialUpdate":
        sgd.backward_and_partial_update(loss)
    elif dist_option == "sparseTopK":
        sgd.backward_and_sparse_update(loss, topK=True, spars=spars)
    elif dist_option == "sparseThreshold":
        sgd.backward_and_sparse_update(loss, topK=False, spars=spars)


def main(argv=None, nccl_id=None, local_rank=None, world_size=None):
    """``nccl_id`` / ``local_rank`` / ``world_size`` are given by a parent that
    starts the ranks itself (mnist_multiprocess.py); under a launcher they
    come from the environment."""
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--max-epoch", type=int, default=2)
    ap.add_argument("-b", "--batch-size", type=int, default=64)
    ap.add_argument("-l", "--lr", type=float, default=0.01)
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--dist-option", default="plain",
                    choices=["plain", "half", "partialUpdate", "sparseTopK",
                             "sparseThreshold"])
    ap.add_argument("-s", "--spars", type=float, default=0.05)
    ap.add_argument("--max-batches", type=int, default=None)
    args = ap.parse_args(argv)

    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5)
    rank, world, local = 0, 1, 0
    if args.dist or nccl_id is not None:
        sgd = opt.DistOpt(sgd, nccl_id=nccl_id, local_rank=local_rank,
                          world_size=world_size)
        rank, world, local = sgd.global_rank, sgd.world_size, sgd.local_rank
    dev = device.create_cuda_gpu_on(local) if device.get_num_gpus() else \
        device.get_default_device()
    train_x, train_y, val_x, val_y = datasets.load("mnist")
    per = len(train_x) // world
    train_x, train_y = train_x[rank * per:(rank + 1) * per], \
        train_y[rank * per:(rank + 1) * per]
    bs = args.batch_size
    nb, nvb = len(train_x) // bs, len(val_x) // bs
    if args.max_batches:
        nb, nvb = min(nb, args.max_batches), min(nvb, args.max_batches)
    model = CNN()
    tx = tensor.Tensor((bs, 1, 28, 28), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    acc = 0.0
    for epoch in range(args.max_epoch):
        start = time.time()
        autograd.training = True
        idx = np.random.permutation(len(train_x))
        loss_sum = 0.0
        for b in range(nb):
            sel = idx[b * bs:(b + 1) * bs]
            tx.copy_from_numpy(train_x[sel])
            ty.copy_from_numpy(train_y[sel])
            loss = autograd.softmax_cross_entropy(model.forward(tx), ty)
            update(sgd, loss, args.dist_option, args.spars)
            loss_sum += float(tensor.to_numpy(loss).reshape(-1)[0])
        autograd.training = False
        correct = 0
        for b in range(nvb):
            tx.copy_from_numpy(val_x[b * bs:(b + 1) * bs])
            pred = np.argmax(tensor.to_numpy(model.forward(tx)), 1)
            correct += int((pred == val_y[b * bs:(b + 1) * bs]).sum())
        acc = correct / max(nvb * bs, 1)
        if rank == 0:
            print("epoch %d: loss %.4f val acc %.3f (%.1fs)" %
                  (epoch, loss_sum / max(nb, 1), acc, time.time() - start),
                  flush=True)
    return acc


if __name__ == "__main__":
    main()
