"""ResNet-50 fine-sized for CIFAR-10, trained with raw autograd calls (no Model
compile, no graph buffering): every batch is augmented (reflect pad + random
crop + flip), enlarged to ``--size`` pixels and pushed through
``autograd.backward`` by the optimizer.  ``--dist`` trains data-parallel; the
per-epoch loss/accuracy are then summed over ranks with the optimizer's own
all-reduce.  Counterpart of the reference's
examples/cnn/autograd/resnet_cifar10.py.

    python resnet_cifar10.py -m 1 --max-batches 20

``--size`` must be at least 224: the network ends in a 7x7 average pool after
five halvings.
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", ".."),
                os.path.join(_HERE, "..", "..", "..")]

from singa_amd import autograd, device, opt, tensor  # noqa: E402
from singa_amd.models import resnet  # noqa: E402
import data as datasets  # noqa: E402


def enlarge(x, size):
    """Bilinear resize of an NCHW batch (sample centres aligned)."""
    n, c, h, w = x.shape
    if (h, w) == (size, size):
        return np.ascontiguousarray(x, dtype=np.float32)

    def taps(src, dst):
        pos = np.clip((np.arange(dst) + 0.5) * src / dst - 0.5, 0, src - 1)
        lo = np.floor(pos).astype(np.int64)
        hi = np.minimum(lo + 1, src - 1)
        return lo, hi, (pos - lo).astype(np.float32)

    r0, r1, rf = taps(h, size)
    c0, c1, cf = taps(w, size)
    rows = x[:, :, r0, :] * (1 - rf)[None, None, :, None] + \
        x[:, :, r1, :] * rf[None, None, :, None]
    out = rows[:, :, :, c0] * (1 - cf) + rows[:, :, :, c1] * cf
    return np.ascontiguousarray(out, dtype=np.float32)


def augment(x, rng):
    n, _, h, w = x.shape
    padded = np.pad(x, [[0, 0], [0, 0], [4, 4], [4, 4]], "symmetric")
    out = np.empty_like(x)
    for i in range(n):
        dy, dx = rng.randint(8, size=2)
        crop = padded[i, :, dy:dy + h, dx:dx + w]
        out[i] = crop[:, :, ::-1] if rng.randint(2) else crop
    return out


def all_reduce_scalar(value, sgd, holder):
    """Sum a host scalar over the ranks through a one-element device tensor."""
    holder.copy_from_numpy(np.asarray([value], np.float32))
    sgd.all_reduce(holder.data)
    sgd.wait()
    return float(tensor.to_numpy(holder)[0])


def average_over_ranks(param, sgd):
    sgd.all_reduce(param.data)
    sgd.wait()
    param /= sgd.world_size


def train_cifar10(argv=None, nccl_id=None, local_rank=None, world_size=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--max-epoch", type=int, default=5)
    ap.add_argument("-b", "--batch-size", type=int, default=32)
    ap.add_argument("-l", "--lr", type=float, default=0.005)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--partial-update", action="store_true")
    ap.add_argument("--max-batches", type=int, default=None)
    args = ap.parse_args(argv)

    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5)
    dist = args.dist or nccl_id is not None
    rank, world, local = 0, 1, 0
    if dist:
        sgd = opt.DistOpt(sgd, nccl_id=nccl_id, local_rank=local_rank,
                          world_size=world_size)
        rank, world, local = sgd.global_rank, sgd.world_size, sgd.local_rank
    dev = device.create_cuda_gpu_on(local) if device.get_num_gpus() else \
        device.get_default_device()

    train_x, train_y, test_x, test_y = datasets.load("cifar10")
    per, per_t = len(train_x) // world, len(test_x) // world
    train_x, train_y = train_x[rank * per:(rank + 1) * per], \
        train_y[rank * per:(rank + 1) * per]
    test_x, test_y = test_x[rank * per_t:(rank + 1) * per_t], \
        test_y[rank * per_t:(rank + 1) * per_t]

    bs, size = args.batch_size, args.size
    model = resnet.resnet50(num_classes=10)
    tx = tensor.Tensor((bs, 3, size, size), dev, tensor.float32)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    holder = tensor.Tensor((1,), dev, tensor.float32)
    nb, ntb = len(train_x) // bs, len(test_x) // bs
    if args.max_batches:
        nb, ntb = min(nb, args.max_batches), min(ntb, args.max_batches)
    rng = np.random.RandomState(rank)

    params = []
    if dist:
        # ranks draw different initial weights: average them once, up front
        autograd.training = True
        tx.gaussian(0.0, 1.0)
        ty.set_value(0)
        loss = autograd.softmax_cross_entropy(model.forward(tx), ty)
        for p, _ in autograd.backward(loss):
            average_over_ranks(p, sgd)
            params.append(p)

    acc = 0.0
    for epoch in range(args.max_epoch):
        start = time.time()
        order = rng.permutation(len(train_x))
        autograd.training = True
        hits, loss_sum = 0.0, 0.0
        for b in range(nb):
            sel = order[b * bs:(b + 1) * bs]
            y = train_y[sel]
            tx.copy_from_numpy(enlarge(augment(train_x[sel], rng), size))
            ty.copy_from_numpy(y)
            out = model.forward(tx)
            loss = autograd.softmax_cross_entropy(out, ty)
            hits += float((np.argmax(tensor.to_numpy(out), 1) == y).sum())
            loss_sum += float(tensor.to_numpy(loss).reshape(-1)[0])
            if not dist:
                sgd(loss)
            elif args.partial_update:
                sgd.backward_and_partial_update(loss)
            else:
                sgd.backward_and_update(loss)
        if dist:
            hits = all_reduce_scalar(hits, sgd, holder)
            loss_sum = all_reduce_scalar(loss_sum, sgd, holder)
        if rank == 0:
            print("epoch %d: training loss %.4f accuracy %.4f" %
                  (epoch, loss_sum / max(nb * world, 1),
                   hits / max(nb * bs * world, 1)), flush=True)

        if dist and args.partial_update:
            # each rank updated a different slice of the weights
            for p in params:
                average_over_ranks(p, sgd)

        autograd.training = False
        hits = 0.0
        for b in range(ntb):
            y = test_y[b * bs:(b + 1) * bs]
            tx.copy_from_numpy(enlarge(test_x[b * bs:(b + 1) * bs], size))
            hits += float((np.argmax(tensor.to_numpy(model.forward(tx)), 1)
                           == y).sum())
        if dist:
            hits = all_reduce_scalar(hits, sgd, holder)
        acc = hits / max(ntb * bs * world, 1)
        if rank == 0:
            print("epoch %d: evaluation accuracy %.4f, %.1fs" %
                  (epoch, acc, time.time() - start), flush=True)
    return acc


if __name__ == "__main__":
    train_cifar10()
