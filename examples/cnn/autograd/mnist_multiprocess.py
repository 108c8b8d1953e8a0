"""MNIST CNN trained data-parallel with the ranks started from this script:

    python mnist_multiprocess.py 2

The parent makes the rendezvous id and passes it to every child.  Counterpart
of the reference's examples/cnn/autograd/mnist_multiprocess.py.
"""
import multiprocessing
import sys

import mnist_cnn
from singa_amd import singa_wrap as singa

if __name__ == "__main__":
    world_size = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nccl_id = singa.NcclIdHolder()
    # spawn: a forked child must not inherit an initialised HIP runtime
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=mnist_cnn.main,
                         args=(sys.argv[2:], nccl_id, rank, world_size))
             for rank in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    raise SystemExit(max(p.exitcode or 0 for p in procs))
