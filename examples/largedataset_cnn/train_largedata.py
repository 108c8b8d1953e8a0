"""Train a zoo model on an image-folder dataset that does not fit in memory:
a background process decodes and augments batches (``ImageBatchIter`` +
``ImageTool``) and ``DevicePrefetcher`` uploads batch i+1 through pinned memory
on a copy stream while the GPU trains on batch i.  Counterpart of the
reference's examples/largedataset_cnn/train_largedata.py.

    python train_largedata.py cnn /data/my_images -m 5 -b 32
    python train_largedata.py cnn --synthetic            # generated dataset
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "cnn"),
                os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]

from singa_amd import data as sdata, image_tool, opt, tensor  # noqa: E402
import process_data  # noqa: E402
import train_cnn as base  # noqa: E402

SIZE = 32


def train_transform(path):
    return image_tool.ImageTool().load(path).resize_by_list([SIZE + 4]) \
        .random_crop((SIZE, SIZE)).flip().get()


def eval_transform(path):
    return image_tool.ImageTool().load(path).resize_by_list([SIZE + 4]) \
        .crop5((SIZE, SIZE), 5).get()[4:]        # centre crop


class Normalised(object):
    """Scale uint8-range images to zero mean / unit range on the fly."""

    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return self

    def __next__(self):
        x, y = next(self.it)
        return (x / 255.0 - 0.5).astype(np.float32), y


def run(args):
    root = args.root
    if args.synthetic or root is None:
        root = process_data.make_synthetic(
            os.path.join(tempfile.gettempdir(), "singa_folder_dataset"))
    prefix = os.path.join(tempfile.gettempdir(), "singa_folder_index")
    classes, counts = process_data.index_folder(root, prefix)
    print("classes", classes, counts)
    dev = base.create_device(0)
    bs = args.batch_size
    model = base.build_model(args.model, 3, len(classes), 3 * SIZE * SIZE)
    model.set_optimizer(opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5))
    # a recorded graph always reads these two tensors: every batch is copied
    # into them (device to device) after its asynchronous upload
    tx = tensor.Tensor((bs, 3, model.input_size, model.input_size), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    model.compile([tx], is_train=True, use_graph=args.graph, sequential=False)

    train_it = sdata.ImageBatchIter(prefix + "_train.txt", bs, train_transform,
                                    shuffle=True, image_folder=root,
                                    capacity=8, num_workers=args.workers)
    val_it = sdata.ImageBatchIter(prefix + "_val.txt", bs, eval_transform,
                                  shuffle=False, image_folder=root,
                                  capacity=4, num_workers=args.workers)
    train_it.start()
    val_it.start()
    result = None
    try:
        feed = sdata.DevicePrefetcher(Normalised(train_it), dev)
        val_feed = Normalised(val_it)
        nb = max(counts["train"] // bs, 1)
        nvb = max(counts["val"] // bs, 1)
        for epoch in range(args.max_epoch):
            start = time.time()
            model.train()
            correct, total_loss = 0, 0.0
            for _ in range(nb):
                x, y = next(feed)
                if model.input_size != SIZE:
                    x = tensor.from_numpy(base.resize_dataset(
                        tensor.to_numpy(x), model.input_size), dev)
                tx.copy_data(x)
                ty.copy_data(y)
                out, loss = model(tx, ty, "plain", None)
                correct += base.accuracy(tensor.to_numpy(out),
                                         tensor.to_numpy(ty))
                total_loss += float(tensor.to_numpy(loss).reshape(-1)[0])
            model.eval()
            vcorrect = 0
            for _ in range(nvb):
                xv, yv = next(val_feed)
                xv = base.resize_dataset(xv, model.input_size)
                tx.copy_from_numpy(xv)
                vcorrect += base.accuracy(tensor.to_numpy(model(tx)), yv)
            result = (total_loss / nb, correct / (nb * bs),
                      vcorrect / (nvb * bs))
            print("epoch %d: loss %.4f train acc %.3f val acc %.3f (%.1fs)" %
                  ((epoch,) + result + (time.time() - start,)), flush=True)
    finally:
        train_it.end()
        val_it.end()
    return result


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="cnn",
                    choices=["cnn", "alexnet", "resnet", "xceptionnet"])
    ap.add_argument("root", nargs="?", default=None,
                    help="image folder: root/<class>/*.png")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("-m", "--max-epoch", type=int, default=5)
    ap.add_argument("-b", "--batch-size", type=int, default=32)
    ap.add_argument("-l", "--learning-rate", type=float, default=0.01,
                    dest="lr")
    ap.add_argument("-g", "--disable-graph", default=True,
                    action="store_false", dest="graph")
    ap.add_argument("--workers", type=int, default=2)
    return run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
