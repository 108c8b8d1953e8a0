"""Throughput of data-parallel ResNet-50 training on one synthetic batch with
raw autograd calls, split into forward / loss / backward+update time.
Counterpart of the reference's examples/cnn/autograd/resnet_dist.py.

    python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 resnet_dist.py
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", "..", "..")]

from singa_amd import autograd, device, opt, tensor  # noqa: E402
from singa_amd.models import resnet  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--niters", type=int, default=100)
    ap.add_argument("-b", "--batch-size", type=int, default=32)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--depth", type=int, default=50,
                    choices=[18, 34, 50, 101, 152])
    args = ap.parse_args(argv)

    sgd = opt.DistOpt(opt.SGD(lr=0.1, momentum=0.9, weight_decay=1e-5))
    dev = device.create_cuda_gpu_on(sgd.local_rank) \
        if device.get_num_gpus() else device.get_default_device()
    model = getattr(resnet, "resnet%d" % args.depth)()
    bs, size = args.batch_size, args.size
    tx = tensor.Tensor((bs, 3, size, size), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    tx.copy_from_numpy(np.random.randn(bs, 3, size, size).astype(np.float32))
    ty.copy_from_numpy(np.random.randint(0, 1000, bs).astype(np.int32))
    autograd.training = True

    def step(clock):
        dev.Sync()
        t0 = time.time()
        out = model.forward(tx)
        dev.Sync()
        t1 = time.time()
        loss = autograd.softmax_cross_entropy(out, ty)
        dev.Sync()
        t2 = time.time()
        sgd.backward_and_update(loss)
        clock[0] += t1 - t0
        clock[1] += t2 - t1

    step([0.0, 0.0])  # parameters and optimizer state come to life here
    clock = [0.0, 0.0]
    dev.Sync()
    start = time.time()
    for _ in range(args.niters):
        step(clock)
    dev.Sync()
    total = time.time() - start
    if sgd.global_rank == 0:
        it = total / args.niters
        fwd, sm = clock[0] / args.niters, clock[1] / args.niters
        print("Throughput = %.1f per second" %
              (sgd.world_size * args.niters * bs / total), flush=True)
        print("Total=%.5f, forward=%.5f, softmax=%.5f, backward=%.5f" %
              (it, fwd, sm, it - fwd - sm), flush=True)


if __name__ == "__main__":
    main()
