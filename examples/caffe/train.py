"""Train a network written for Caffe: the layers come from a .prototxt, the
optimizer from the solver file, and the parameters can start from (and are
written back to) a .caffemodel.

    python train.py                               # LeNet on synthetic digits
    python train.py --net my.prototxt --solver my_solver.prototxt \
                    --weights my.caffemodel --shape 3 224 224 --classes 1000
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                "..", ".."))
from singa_amd import converter, device, tensor  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic(n, shape, classes, seed=0):
    """Images whose class is readable from their content: a bright band whose
    position depends on the label, plus noise."""
    rng = np.random.RandomState(seed)
    y = rng.randint(0, classes, n).astype(np.int32)
    x = rng.randn(n, *shape).astype(np.float32) * 0.3
    band = max(shape[1] // classes, 1)
    for i, label in enumerate(y):
        lo = (label * band) % shape[1]
        x[i, :, lo:lo + band, :] += 1.5
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default=os.path.join(HERE, "lenet.prototxt"))
    ap.add_argument("--solver",
                    default=os.path.join(HERE, "lenet_solver.prototxt"))
    ap.add_argument("--weights", default=None, help=".caffemodel to start from")
    ap.add_argument("--save", default=None, help=".caffemodel to write")
    ap.add_argument("--shape", type=int, nargs=3, default=None)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("-g", "--no-graph", dest="graph", action="store_false")
    args = ap.parse_args()

    dev = device.get_default_device() if args.cpu else \
        device.create_cuda_gpu()
    conv = converter.CaffeConverter(args.net, args.solver,
                                    input_sample_shape=args.shape,
                                    param_path=args.weights)
    net = conv.create_net("train")
    net.set_optimizer(conv.create_optimizer())
    shape = tuple(net.input_sample_shape)
    data, labels = synthetic(args.batch * 16, shape, args.classes)
    tx = tensor.Tensor((args.batch,) + shape, dev)
    ty = tensor.Tensor((args.batch,), dev, tensor.int32)
    tx.copy_from_numpy(data[:args.batch])
    net.compile([tx], is_train=True, use_graph=args.graph, sequential=False)
    if args.weights:
        print("restored", conv.convert_params(net))
    net.train()
    start = time.time()
    for it in range(args.iters):
        lo = (it * args.batch) % (len(data) - args.batch + 1)
        tx.copy_from_numpy(data[lo:lo + args.batch])
        ty.copy_from_numpy(labels[lo:lo + args.batch])
        out, loss = net(tx, ty)
        if it % 50 == 0 or it == args.iters - 1:
            pred = np.argmax(tensor.to_numpy(out), 1)
            print("iteration %4d  loss %.4f  accuracy %.3f" % (
                it, float(tensor.to_numpy(loss).ravel()[0]),
                float((pred == labels[lo:lo + args.batch]).mean())),
                flush=True)
    dev.Sync()
    print("%.1f iterations/s" % (args.iters / (time.time() - start)))
    if args.save:
        converter.export_caffemodel(net, args.save)
        print("wrote", args.save)


if __name__ == "__main__":
    main()
