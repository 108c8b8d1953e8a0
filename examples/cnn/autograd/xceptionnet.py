"""Xception trained on one synthetic 299x299 batch with raw autograd calls.
The network itself lives in ``singa_amd.models.xceptionnet``.  Counterpart of
the reference's examples/cnn/autograd/xceptionnet.py.

    python xceptionnet.py -n 20 -b 16
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", "..", "..")]

from singa_amd import autograd, device, opt, tensor  # noqa: E402
from singa_amd.models.xceptionnet import Xception  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--niters", type=int, default=20)
    ap.add_argument("-b", "--batch-size", type=int, default=16)
    ap.add_argument("--size", type=int, default=299)
    args = ap.parse_args(argv)

    model = Xception(num_classes=1000)
    dev = device.create_cuda_gpu_on(0) if device.get_num_gpus() else \
        device.get_default_device()
    sgd = opt.SGD(lr=0.1, momentum=0.9, weight_decay=1e-5)
    bs, size = args.batch_size, args.size
    tx = tensor.Tensor((bs, 3, size, size), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    tx.copy_from_numpy(np.random.randn(bs, 3, size, size).astype(np.float32))
    ty.copy_from_numpy(np.random.randint(0, 1000, bs).astype(np.int32))
    autograd.training = True
    loss = None
    for it in range(args.niters + 1):
        if it == 1:  # the first step creates parameters and optimizer state
            dev.Sync()
            start = time.time()
        loss = autograd.softmax_cross_entropy(model.forward(tx), ty)
        sgd(loss)
    dev.Sync()
    total = time.time() - start
    print("loss %.4f; %.1f images/s (%.1f ms per step)" %
          (float(tensor.to_numpy(loss).reshape(-1)[0]),
           args.niters * bs / max(total, 1e-9),
           1e3 * total / max(args.niters, 1)), flush=True)


if __name__ == "__main__":
    main()
