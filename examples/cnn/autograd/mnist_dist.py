"""MNIST CNN trained data-parallel with plain gradient all-reduce; one rank
per GPU under a launcher.  Counterpart of the reference's
examples/cnn/autograd/mnist_dist.py.

    python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 mnist_dist.py
"""
import sys

import mnist_cnn

if __name__ == "__main__":
    mnist_cnn.main(["--dist"] + sys.argv[1:])
