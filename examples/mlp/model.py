"""MLP trained through the Model API with graph buffering on a 2-d toy
problem (counterpart of the reference's examples/mlp/model.py:91-147).

    python model.py            # 1001 steps, graph on
    python model.py -g -m 200  # eager
"""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402
from singa_amd.models.mlp import MLP, create_model  # noqa: E402,F401
import _toy  # noqa: E402

np_dtype = {"float16": np.float16, "float32": np.float32}
singa_dtype = {"float16": tensor.float16, "float32": tensor.float32}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-p", choices=["float32", "float16"],
                        default="float32", dest="precision")
    parser.add_argument("-g", "--disable-graph", default=True,
                        action="store_false", dest="graph")
    parser.add_argument("-m", "--max-epoch", default=1001, type=int,
                        dest="max_epoch")
    parser.add_argument("--seed", default=None, type=int,
                        help="seed of the device's generator (the initial "
                             "weights); unset: whatever state it is in")
    args = parser.parse_args(argv)

    data, label = _toy.make()
    dev = device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()
    if args.seed is not None:
        dev.SetRandSeed(args.seed)
    sgd = opt.SGD(0.1, 0.9, 1e-5, dtype=singa_dtype[args.precision])
    tx = tensor.Tensor((400, 2), dev, singa_dtype[args.precision])
    ty = tensor.Tensor((400,), dev, tensor.int32)
    model = MLP(data_size=2, perceptron_size=3, num_classes=2)
    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=args.graph, sequential=True)
    model.train()
    loss_value = None
    for i in range(args.max_epoch):
        tx.copy_from_numpy(data.astype(np_dtype[args.precision]))
        ty.copy_from_numpy(label)
        out, loss = model(tx, ty, "plain", spars=None)
        if i % 100 == 0 or i == args.max_epoch - 1:
            loss_value = float(tensor.to_numpy(loss).reshape(-1)[0])
            print("training loss = ", loss_value, flush=True)
    return loss_value


if __name__ == "__main__":
    main()
