"""The same MLP written against raw autograd operators with hand-made
parameter tensors (counterpart of the reference's examples/mlp/native.py)."""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "..")]

from singa_amd import autograd, device, opt, tensor  # noqa: E402
from singa_amd.tensor import Tensor  # noqa: E402
import _toy  # noqa: E402

np_dtype = {"float16": np.float16, "float32": np.float32}
singa_dtype = {"float16": tensor.float16, "float32": tensor.float32}


def _param(dev, precision, shape=None, data=None):
    t = Tensor(shape=shape, data=data, device=dev, dtype=precision,
               requires_grad=True, stores_grad=True)
    if data is None:
        t.set_value(0.0)
    return t


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-p", choices=["float32", "float16"],
                        default="float32", dest="precision")
    parser.add_argument("-m", "--max-epoch", default=1001, type=int,
                        dest="max_epoch")
    args = parser.parse_args(argv)
    np.random.seed(0)
    autograd.training = True

    data, label = _toy.make()
    print("train_data_shape:", data.shape)
    print("train_label_shape:", label.shape)
    precision = singa_dtype[args.precision]
    npp = np_dtype[args.precision]
    dev = device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()

    inputs = Tensor(data=data, device=dev).as_type(precision)
    target = Tensor(data=label, device=dev)
    w0 = _param(dev, precision,
                data=np.random.normal(0, 0.1, (2, 3)).astype(npp))
    b0 = _param(dev, precision, shape=(3,))
    w1 = _param(dev, precision,
                data=np.random.normal(0, 0.1, (3, 2)).astype(npp))
    b1 = _param(dev, precision, shape=(2,))

    sgd = opt.SGD(0.05, 0.8)
    value = None
    for i in range(args.max_epoch):
        x = autograd.matmul(inputs, w0)
        x = autograd.add_bias(x, b0)
        x = autograd.relu(x)
        x = autograd.matmul(x, w1)
        x = autograd.add_bias(x, b1)
        loss = autograd.softmax_cross_entropy(x, target)
        sgd(loss)
        if i % 100 == 0 or i == args.max_epoch - 1:
            value = float(tensor.to_numpy(loss).reshape(-1)[0])
            print("%d, training loss = " % i, value)
    autograd.training = False
    return value


if __name__ == "__main__":
    main()
