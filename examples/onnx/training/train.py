"""Fine-tune an imported ONNX backbone on CIFAR with a new classifier head,
single GPU or data parallel (counterpart of the reference's
examples/onnx/training/train.py:105-330).

    python train.py --model /data/resnet18v1.onnx --last-layers -1 --in-dim 1000
    python train.py --self-export          # backbone exported from this
                                           # framework's own ResNet-18
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        train.py --self-export --dist
"""
import argparse
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..", "cnn"),
                os.path.join(_HERE, "..", ".."),
                os.path.join(_HERE, "..", "..", "..")]

from singa_amd import autograd, device, layer, models, opt, sonnx, tensor  # noqa: E402
from singa_amd import onnx_proto as onnx  # noqa: E402
from singa_amd.models.common import apply_update  # noqa: E402
import train_cnn as base  # noqa: E402
import data as datasets  # noqa: E402


class MyModel(sonnx.SONNXModel):
    """Imported graph cut at ``last_layers`` + a trainable linear head."""

    def __init__(self, onnx_model, num_classes=10, num_channels=3,
                 last_layers=-1, in_dim=1000, input_size=224):
        super(MyModel, self).__init__(onnx_model)
        self.num_classes = num_classes
        self.input_size = input_size
        self.dimension = 4
        self.num_channels = num_channels
        self.last_layers = last_layers
        self.linear = layer.Linear(in_dim, num_classes)

    def forward(self, *x):
        y = super(MyModel, self).forward(*x, last_layers=self.last_layers)[0]
        if len(y.shape) > 2:
            y = autograd.flatten(y)
        return self.linear(y)

    def train_one_batch(self, x, y, dist_option="plain", spars=None):
        out = self.forward(x)
        loss = autograd.softmax_cross_entropy(out, y)
        apply_update(self.optimizer, loss,
                     {"fp32": "plain", "fp16": "half"}.get(dist_option,
                                                          dist_option), spars)
        return out, loss

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer


def self_exported_backbone(dev, input_size):
    net = models.resnet.resnet18(num_channels=3, num_classes=1000)
    x = tensor.Tensor((1, 3, input_size, input_size), dev)
    x.gaussian(0, 1)
    net.compile([x], is_train=False, use_graph=False, sequential=True)
    net.eval()
    return sonnx.to_onnx([x], [net.forward(x)])


def run(args, sgd, global_rank=0, world_size=1, local_rank=0):
    dev = base.create_device(local_rank)
    dev.SetRandSeed(0)
    np.random.seed(0)
    train_x, train_y, val_x, val_y = datasets.load(args.data)
    num_classes = int(np.max(train_y)) + 1
    if args.self_export:
        onnx_model = self_exported_backbone(dev, args.input_size)
    else:
        onnx_model = onnx.load(args.model)
    model = MyModel(onnx_model, num_classes=num_classes,
                    last_layers=args.last_layers, in_dim=args.in_dim,
                    input_size=args.input_size)
    DIST = hasattr(sgd, "communicator")
    if DIST:
        train_x, train_y, val_x, val_y = base.partition(
            global_rank, world_size, train_x, train_y, val_x, val_y)
    bs = args.batch_size
    tx = tensor.Tensor((bs, 3, model.input_size, model.input_size), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=args.graph, sequential=DIST)
    nb = train_x.shape[0] // bs
    nvb = val_x.shape[0] // bs
    if args.max_batches:
        nb, nvb = min(nb, args.max_batches), min(nvb, args.max_batches)
    idx = np.arange(train_x.shape[0])
    result = None
    for epoch in range(args.max_epoch):
        start = time.time()
        np.random.shuffle(idx)
        model.train()
        correct, total_loss = 0, 0.0
        for b in range(nb):
            sel = idx[b * bs:(b + 1) * bs]
            x = base.resize_dataset(base.augmentation(train_x[sel]),
                                    model.input_size)
            tx.copy_from_numpy(x)
            ty.copy_from_numpy(train_y[sel])
            out, loss = model(tx, ty, args.dist_option, args.spars)
            correct += base.accuracy(tensor.to_numpy(out), train_y[sel])
            total_loss += float(tensor.to_numpy(loss).reshape(-1)[0])
        model.eval()
        vcorrect = 0
        for b in range(nvb):
            x = base.resize_dataset(val_x[b * bs:(b + 1) * bs],
                                    model.input_size)
            tx.copy_from_numpy(x)
            vcorrect += base.accuracy(tensor.to_numpy(model(tx)),
                                      val_y[b * bs:(b + 1) * bs])
        result = (total_loss, correct / max(nb * bs, 1),
                  vcorrect / max(nvb * bs, 1))
        if global_rank == 0:
            print("epoch %d: loss %f train acc %f val acc %f (%.1fs)" %
                  ((epoch,) + result + (time.time() - start,)), flush=True)
    return result


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None, help="backbone .onnx file")
    ap.add_argument("--self-export", action="store_true")
    ap.add_argument("--last-layers", type=int, default=-1)
    ap.add_argument("--in-dim", type=int, default=1000)
    ap.add_argument("--input-size", type=int, default=224)
    ap.add_argument("--data", default="cifar10",
                    choices=["cifar10", "cifar100"])
    ap.add_argument("-m", "--max-epoch", type=int, default=10)
    ap.add_argument("-b", "--batch-size", type=int, default=32)
    ap.add_argument("-l", "--learning-rate", type=float, default=0.005,
                    dest="lr")
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("-d", "--dist-option", default="plain",
                    choices=["plain", "fp32", "fp16", "half", "partialUpdate",
                             "sparseTopK", "sparseThreshold"])
    ap.add_argument("-s", "--sparsification", type=float, default=0.05,
                    dest="spars")
    ap.add_argument("-g", "--disable-graph", default=True,
                    action="store_false", dest="graph")
    ap.add_argument("--max-batches", type=int, default=None)
    args = ap.parse_args(argv)
    assert args.model or args.self_export, "give --model or --self-export"
    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5)
    if args.dist:
        sgd = opt.DistOpt(sgd)
        return run(args, sgd, sgd.global_rank, sgd.world_size,
                   sgd.local_rank)
    return run(args, sgd)


if __name__ == "__main__":
    main()
