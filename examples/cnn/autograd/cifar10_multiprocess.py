"""ResNet-50 on CIFAR-10, data-parallel, with the ranks started from this
script (``python cifar10_multiprocess.py 2 [options of resnet_cifar10.py]``).
Counterpart of the reference's examples/cnn/autograd/cifar10_multiprocess.py.
"""
import multiprocessing
import sys

import resnet_cifar10
from singa_amd import singa_wrap as singa

if __name__ == "__main__":
    world_size = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nccl_id = singa.NcclIdHolder()
    # spawn: a forked child must not inherit an initialised HIP runtime
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=resnet_cifar10.train_cifar10,
                         args=(sys.argv[2:], nccl_id, rank, world_size))
             for rank in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    raise SystemExit(max(p.exitcode or 0 for p in procs))
