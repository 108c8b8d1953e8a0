"""ResNet-50 training throughput on synthetic ImageNet-shaped data (batch 32,
224x224, fp32, SGD momentum 0.9): the reference's headline benchmark
(examples/cnn/benchmark.py:33-90).

    python benchmark.py                      # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        benchmark.py --dist
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]

from singa_amd import device, models, opt, tensor  # noqa: E402


def train_resnet(DIST=True, graph=True, sequential=False, verbosity=0,
                 niters=100, batch_size=32, img_size=224, depth=50):
    sgd = opt.SGD(lr=0.1, momentum=0.9, weight_decay=1e-5)
    if DIST:
        sgd = opt.DistOpt(sgd)
        # 32 MiB buckets: a ring over xGMI is latency-bound per step
        sgd.threshold = 8 * 1024 * 1024
        world_size, local_rank, global_rank = (sgd.world_size, sgd.local_rank,
                                               sgd.global_rank)
        sequential = True
    else:
        world_size, local_rank, global_rank = 1, 0, 0

    dev = device.create_cuda_gpu_on(local_rank) if device.get_num_gpus() \
        else device.get_default_device()
    tx = tensor.Tensor((batch_size, 3, img_size, img_size), dev)
    ty = tensor.Tensor((batch_size,), dev, tensor.int32)
    tx.copy_from_numpy(np.random.randn(batch_size, 3, img_size,
                                       img_size).astype(np.float32))
    ty.copy_from_numpy(np.random.randint(0, 1000, batch_size, dtype=np.int32))

    dev.SetVerbosity(verbosity)
    dev.SetSkipIteration(5)

    model = getattr(models.resnet, "resnet%d" % depth)(num_channels=3,
                                                       num_classes=1000)
    model.train()
    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=graph, sequential=sequential)

    for _ in range(3):  # graph construction and first replays
        model(tx, ty, dist_option="fp32", spars=None)
    dev.Sync()
    start = time.time()
    for _ in range(niters):
        model(tx, ty, dist_option="fp32", spars=None)
    dev.Sync()
    end = time.time()
    throughput = float(niters * batch_size * world_size) / (end - start)
    if global_rank == 0:
        print("\nThroughput = {} per second".format(throughput), flush=True)
        print("TotalTime={}".format(end - start), flush=True)
        print("Total={}".format((end - start) / float(niters)), flush=True)
        dev.PrintTimeProfiling()
    return throughput


if __name__ == "__main__":
    parser = argparse.ArgumentParser(
        description="Throughput test using Resnet 50")
    parser.add_argument("--dist", "--enable-dist", default=False,
                        action="store_true", dest="DIST")
    parser.add_argument("--no-graph", "--disable-graph", default=True,
                        action="store_false", dest="graph")
    parser.add_argument("--verbosity", "--log-verbosity", default=0, type=int,
                        dest="verbosity")
    parser.add_argument("--iters", default=100, type=int)
    parser.add_argument("--batch-size", default=32, type=int)
    parser.add_argument("--img-size", default=224, type=int)
    parser.add_argument("--depth", default=50, type=int)
    args = parser.parse_args()
    train_resnet(DIST=args.DIST, graph=args.graph, sequential=False,
                 verbosity=args.verbosity, niters=args.iters,
                 batch_size=args.batch_size, img_size=args.img_size,
                 depth=args.depth)
