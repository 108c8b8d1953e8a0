"""Distributed-training throughput harness: one augmented batch is uploaded
once and trained on for ``--iters`` iterations; rank 0 prints the aggregate
images/s.  Same role and output as the reference's
examples/cifar_distributed_cnn/train_cnn.py:190-250 (the configuration its
run-rtx.sh launches with ``-l 0.015 -b 32`` on 8 ranks).

    python train_cnn.py resnet cifar10 -b 32
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        train_mpi.py resnet cifar10 -l 0.015 -b 32
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# this directory first: children started by multiprocessing inherit the path
# and must find THIS train_cnn.py, not the one in examples/cnn
sys.path[:0] = [_HERE, os.path.join(_HERE, ".."),
                os.path.join(_HERE, "..", "..")]

import importlib.util  # noqa: E402


def _load_base():
    """examples/cnn/train_cnn.py under a name of its own: this file has the
    same module name, so a plain import would find this file again when it
    is imported by train_mpi.py / train_multiprocess.py."""
    spec = importlib.util.spec_from_file_location(
        "cnn_train_cnn", os.path.join(_HERE, "..", "cnn", "train_cnn.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cnn_train_cnn"] = mod
    spec.loader.exec_module(mod)
    return mod


base = _load_base()
from singa_amd import opt, tensor  # noqa: E402
import data as datasets  # noqa: E402


def run(global_rank, world_size, local_rank, max_epoch, batch_size, model,
        data, sgd, graph, verbosity, dist_option="plain", spars=None,
        precision="float32", niters=100, warmup=5):
    dev = base.create_device(local_rank)
    dev.SetRandSeed(0)
    np.random.seed(0)
    train_x, train_y, val_x, val_y = datasets.load(data)
    num_channels, image_size = train_x.shape[1], train_x.shape[2]
    data_size = int(np.prod(train_x.shape[1:]))
    num_classes = int(np.max(train_y)) + 1
    model = base.build_model(model, num_channels, num_classes, data_size)

    DIST = hasattr(sgd, "communicator")
    if DIST:
        train_x, train_y, val_x, val_y = base.partition(
            global_rank, world_size, train_x, train_y, val_x, val_y)

    x, y = train_x[:batch_size], train_y[:batch_size]
    if model.dimension == 4:
        x = base.resize_dataset(base.augmentation(x), model.input_size)
        tx = tensor.Tensor((batch_size, num_channels, model.input_size,
                            model.input_size), dev,
                           base.singa_dtype[precision])
    else:
        x = x.reshape(batch_size, -1)
        tx = tensor.Tensor((batch_size, data_size), dev,
                           base.singa_dtype[precision])
    ty = tensor.Tensor((batch_size,), dev, tensor.int32)
    tx.copy_from_numpy(x.astype(base.np_dtype[precision]))
    ty.copy_from_numpy(y)

    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=graph, sequential=DIST)
    dev.SetVerbosity(verbosity)
    model.train()
    if global_rank == 0:
        print("Check the shape of dataset:")
        print(tx.shape)
        print(ty.shape)

    for _ in range(warmup):
        model(tx, ty, dist_option, spars)
    dev.Sync()
    start = time.time()
    for _ in range(niters):
        model(tx, ty, dist_option, spars)
    dev.Sync()
    end = time.time()
    throughput = float(niters * batch_size * world_size) / (end - start)
    if global_rank == 0:
        print("Throughput = {} per second".format(throughput), flush=True)
        print("TotalTime={}".format(end - start), flush=True)
        print("Total={}".format((end - start) / float(niters)), flush=True)
        print("world_size={}".format(world_size), flush=True)
        print("batch_size={}".format(batch_size), flush=True)
        print("model.input_size={}".format(model.input_size), flush=True)
        print("num_channels={}".format(num_channels), flush=True)
        print("num_classes={}".format(num_classes), flush=True)
        print("data_size={}".format(data_size), flush=True)
        print("image_size={}".format(image_size), flush=True)
    dev.PrintTimeProfiling()
    return throughput


def parser_with_common_args(description):
    parser = base.add_common_args(argparse.ArgumentParser(
        description=description))
    parser.add_argument("--iters", default=100, type=int)
    return parser


if __name__ == "__main__":
    parser = parser_with_common_args("Throughput harness, single device.")
    parser.add_argument("-i", "--device-id", default=0, type=int,
                        dest="device_id")
    args = parser.parse_args()
    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                  dtype=base.singa_dtype[args.precision])
    run(0, 1, args.device_id, args.max_epoch, args.batch_size, args.model,
        args.data, sgd, args.graph, args.verbosity, precision=args.precision,
        niters=args.iters)
