"""Train an image classifier from the model zoo on MNIST / CIFAR, on one GPU or
as one rank of a data-parallel job.

    python train_cnn.py cnn mnist -m 2
    python train_cnn.py resnet cifar10 -b 64 -l 0.05

Same command line as the reference's examples/cnn/train_cnn.py:263-329.  The
epoch loop differs where the host would otherwise starve an MI355X: crops and
flips of a batch are computed with array indexing rather than per-image Python
loops, and resizing goes through one batched bilinear interpolation.
``train_mpi.py`` and ``train_multiprocess.py`` call :func:`run` once per rank.
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# examples/ (shared data loaders) and the source tree when not installed
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]

from singa_amd import device, models, opt, tensor  # noqa: E402
import data as datasets  # noqa: E402

np_dtype = {"float16": np.float16, "float32": np.float32}
singa_dtype = {"float16": tensor.float16, "float32": tensor.float32}


def augmentation(x, batch_size=None, rng=np.random):
    """Random 4-pixel shift (symmetric padding) and horizontal flip, for the
    whole batch at once."""
    n, _, h, w = x.shape
    xpad = np.pad(x, [[0, 0], [0, 0], [4, 4], [4, 4]], "symmetric")
    oy = rng.randint(0, 8, n)
    ox = rng.randint(0, 8, n)
    rows = oy[:, None] + np.arange(h)[None, :]              # (n, h)
    cols = ox[:, None] + np.arange(w)[None, :]              # (n, w)
    flip = rng.randint(0, 2, n).astype(bool)
    cols[flip] = cols[flip, ::-1]
    return xpad[np.arange(n)[:, None, None, None],
                np.arange(x.shape[1])[None, :, None, None],
                rows[:, None, :, None], cols[:, None, None, :]]


def accuracy(pred, target):
    return int((np.argmax(pred, axis=1) == target).sum())


def partition(global_rank, world_size, train_x, train_y, val_x, val_y):
    """Contiguous shard of the training and validation sets for this rank."""
    def shard(a):
        per = a.shape[0] // world_size
        return a[global_rank * per:(global_rank + 1) * per]

    return shard(train_x), shard(train_y), shard(val_x), shard(val_y)


def reduce_variable(variable, dist_opt, reducer):
    """Sum a host scalar over all ranks through the optimizer's communicator."""
    reducer.copy_from_numpy(np.asarray(variable, np.float32).reshape(1))
    dist_opt.all_reduce(reducer.data)
    dist_opt.wait()
    return tensor.to_numpy(reducer)


def resize_dataset(x, image_size):
    """Bilinear resize of an (N,C,H,W) array to image_size x image_size."""
    n, c, h, w = x.shape
    if h == image_size and w == image_size:
        return x
    ys = (np.arange(image_size) + 0.5) * h / image_size - 0.5
    xs = (np.arange(image_size) + 0.5) * w / image_size - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    wy = np.clip(ys - y0, 0, 1).astype(np.float32)[None, None, :, None]
    wx = np.clip(xs - x0, 0, 1).astype(np.float32)[None, None, None, :]
    top = x[:, :, y0][:, :, :, x0] * (1 - wx) + x[:, :, y0][:, :, :, x1] * wx
    bot = x[:, :, y1][:, :, :, x0] * (1 - wx) + x[:, :, y1][:, :, :, x1] * wx
    return (top * (1 - wy) + bot * wy).astype(np.float32)


def create_device(local_rank=0):
    gpus = device.get_num_gpus()
    if gpus > 0:
        # more ranks than GPUs only happens in a rehearsal over gloo, where
        # the ranks may share a card
        return device.create_cuda_gpu_on(local_rank % gpus)
    return device.get_default_device()


def build_model(name, num_channels, num_classes, data_size):
    if name == "resnet":
        return models.resnet.resnet50(num_channels=num_channels,
                                      num_classes=num_classes)
    if name == "mlp":
        return models.mlp.create_model(data_size=data_size,
                                       num_classes=num_classes)
    return models.create(name, num_channels=num_channels,
                         num_classes=num_classes)


def run(global_rank, world_size, local_rank, max_epoch, batch_size, model,
        data, sgd, graph, verbosity, dist_option="plain", spars=None,
        precision="float32", max_batches=None, checkpoint_dir=None,
        checkpoint_every=1):
    """``checkpoint_dir``: write a checkpoint there every
    ``checkpoint_every`` epochs (model, optimizer state, epoch) and, when one
    is already present, continue from the newest instead of starting over --
    what a job restarted by tools/launch_restart.py needs."""
    dev = create_device(local_rank)
    dev.SetRandSeed(0)
    np.random.seed(0)

    train_x, train_y, val_x, val_y = datasets.load(data)
    num_channels = train_x.shape[1]
    image_size = train_x.shape[2]
    data_size = int(np.prod(train_x.shape[1:]))
    num_classes = int(np.max(train_y)) + 1
    model = build_model(model, num_channels, num_classes, data_size)

    DIST = hasattr(sgd, "communicator")
    # a distributed step interleaves collectives with backward: replay the
    # graph in recorded order
    sequential = DIST
    if DIST:
        train_x, train_y, val_x, val_y = partition(global_rank, world_size,
                                                   train_x, train_y, val_x,
                                                   val_y)

    if model.dimension == 4:
        tx = tensor.Tensor(
            (batch_size, num_channels, model.input_size, model.input_size),
            dev, singa_dtype[precision])
    else:
        tx = tensor.Tensor((batch_size, data_size), dev,
                           singa_dtype[precision])
        train_x = train_x.reshape(train_x.shape[0], -1)
        val_x = val_x.reshape(val_x.shape[0], -1)
    ty = tensor.Tensor((batch_size,), dev, tensor.int32)

    num_train_batch = train_x.shape[0] // batch_size
    num_val_batch = val_x.shape[0] // batch_size
    if max_batches:
        num_train_batch = min(num_train_batch, max_batches)
        num_val_batch = min(num_val_batch, max_batches)
    idx = np.arange(train_x.shape[0], dtype=np.int32)

    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=graph, sequential=sequential)
    dev.SetVerbosity(verbosity)
    reducer = tensor.Tensor((1,), dev, tensor.float32) if DIST else None

    history = []
    first_epoch, manager = 0, None
    if checkpoint_dir:
        from singa_amd.utils.checkpoint import CheckpointManager
        manager = CheckpointManager(checkpoint_dir, keep=2, rank=global_rank)
        resumed = manager.restore(model, sgd)
        if resumed:
            first_epoch = int(resumed.get("epoch", -1)) + 1
            if global_rank == 0:
                print("Resuming after epoch %d (checkpoint of step %d)" %
                      (first_epoch - 1, resumed["step"]), flush=True)
    for epoch in range(first_epoch, max_epoch):
        start_time = time.time()
        if manager is not None:
            np.random.seed(epoch)    # the same order whether resumed or not
        np.random.shuffle(idx)
        if global_rank == 0:
            print("Starting Epoch %d:" % epoch)

        train_correct, train_loss, test_correct = 0.0, 0.0, 0.0
        model.train()
        for b in range(num_train_batch):
            sel = idx[b * batch_size:(b + 1) * batch_size]
            x = train_x[sel]
            if model.dimension == 4:
                x = augmentation(x)
                if image_size != model.input_size:
                    x = resize_dataset(x, model.input_size)
            y = train_y[sel]
            tx.copy_from_numpy(x.astype(np_dtype[precision]))
            ty.copy_from_numpy(y)
            out, loss = model(tx, ty, dist_option, spars)
            train_correct += accuracy(tensor.to_numpy(out), y)
            train_loss += float(tensor.to_numpy(loss).reshape(-1)[0])

        if DIST:
            train_correct = reduce_variable(train_correct, sgd, reducer)[0]
            train_loss = reduce_variable(train_loss, sgd, reducer)[0]
        seen = num_train_batch * batch_size * world_size
        if global_rank == 0:
            print("Training loss = %f, training accuracy = %f" %
                  (train_loss, train_correct / seen), flush=True)

        model.eval()
        for b in range(num_val_batch):
            x = val_x[b * batch_size:(b + 1) * batch_size]
            if model.dimension == 4 and image_size != model.input_size:
                x = resize_dataset(x, model.input_size)
            y = val_y[b * batch_size:(b + 1) * batch_size]
            tx.copy_from_numpy(x.astype(np_dtype[precision]))
            ty.copy_from_numpy(y)
            out_test = model(tx)
            test_correct += accuracy(tensor.to_numpy(out_test), y)
        if DIST:
            test_correct = reduce_variable(test_correct, sgd, reducer)[0]
        val_seen = max(num_val_batch * batch_size * world_size, 1)
        if global_rank == 0:
            print("Evaluation accuracy = %f, Elapsed Time = %fs" %
                  (test_correct / val_seen, time.time() - start_time),
                  flush=True)
        history.append((float(train_loss), float(train_correct / seen),
                        float(test_correct / val_seen)))
        if manager is not None and ((epoch + 1) % checkpoint_every == 0 or
                                    epoch + 1 == max_epoch):
            manager.save(model, (epoch + 1) * num_train_batch, sgd,
                         {"epoch": epoch,
                          "val_accuracy": float(test_correct / val_seen)})

    dev.PrintTimeProfiling()
    return history


def add_common_args(parser):
    parser.add_argument("model", nargs="?", default="cnn",
                        choices=["cnn", "resnet", "xceptionnet", "mlp",
                                 "alexnet"])
    parser.add_argument("data", nargs="?", default="mnist",
                        choices=["mnist", "cifar10", "cifar100"])
    parser.add_argument("-p", choices=["float32", "float16"],
                        default="float32", dest="precision")
    parser.add_argument("-m", "--max-epoch", default=100, type=int,
                        dest="max_epoch")
    parser.add_argument("-b", "--batch-size", default=64, type=int,
                        dest="batch_size")
    parser.add_argument("-l", "--learning-rate", default=0.005, type=float,
                        dest="lr")
    parser.add_argument("-g", "--disable-graph", default=True,
                        action="store_false", dest="graph")
    parser.add_argument("-v", "--log-verbosity", default=0, type=int,
                        dest="verbosity")
    parser.add_argument("--max-batches", default=None, type=int,
                        help="cap on batches per epoch (smoke runs)")
    parser.add_argument("--checkpoint-dir", default=None, dest="checkpoint_dir",
                        help="checkpoint every --checkpoint-every epochs into "
                             "this directory; continue from the newest one "
                             "found there")
    parser.add_argument("--checkpoint-every", default=1, type=int,
                        dest="checkpoint_every")
    return parser


if __name__ == "__main__":
    parser = add_common_args(argparse.ArgumentParser(
        description="Training using the autograd and graph."))
    parser.add_argument("-i", "--device-id", default=0, type=int,
                        dest="device_id")
    args = parser.parse_args()
    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                  dtype=singa_dtype[args.precision])
    run(0, 1, args.device_id, args.max_epoch, args.batch_size, args.model,
        args.data, sgd, args.graph, args.verbosity, precision=args.precision,
        max_batches=args.max_batches, checkpoint_dir=args.checkpoint_dir,
        checkpoint_every=args.checkpoint_every)
