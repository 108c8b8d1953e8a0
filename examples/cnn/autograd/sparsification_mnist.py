"""MNIST CNN trained data-parallel with sparsified gradient exchange: each
rank sends only the largest ``--spars`` fraction of every gradient (top-K) or,
with ``--no-topk``, the entries above ``--spars`` in magnitude, and keeps the
rest as a local residual.  Counterpart of the reference's
examples/cnn/autograd/sparsification_mnist.py.

    python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 sparsification_mnist.py --spars 0.05
"""
import argparse

import mnist_cnn

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-topk", action="store_true")
    args, rest = ap.parse_known_args()
    option = "sparseThreshold" if args.no_topk else "sparseTopK"
    mnist_cnn.main(["--dist", "--dist-option", option] + rest)
