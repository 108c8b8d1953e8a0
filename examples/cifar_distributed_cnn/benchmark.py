"""ResNet-50 throughput over the ranks of a data-parallel job: the reference's
examples/cifar_distributed_cnn/benchmark.py:34-92 (same program as
examples/cnn/benchmark.py; rank 0 prints images/s of the whole job and, with
``--verbosity``, the time per operator including the ``Dist_*`` exchanges).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        benchmark.py --dist
"""
import os
import runpy
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    target = os.path.join(_HERE, "..", "cnn", "benchmark.py")
    sys.argv[0] = target
    runpy.run_path(target, run_name="__main__")
