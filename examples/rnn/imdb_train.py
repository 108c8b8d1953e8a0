"""Train the IMDB sentiment model (counterpart of the reference's
examples/rnn/imdb_train.py).

    python imdb_data.py /path/to/aclImdb      # once, optional
    python imdb_train.py -m 5 -b 128 --mode lstm
"""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402
import imdb_data  # noqa: E402
from imdb_model import IMDBModel  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--max-epoch", default=5, type=int,
                        dest="max_epoch")
    parser.add_argument("-b", "--batch-size", default=128, type=int, dest="bs")
    parser.add_argument("-l", "--learning-rate", default=0.01, type=float,
                        dest="lr")
    parser.add_argument("-i", "--device-id", default=0, type=int,
                        dest="device_id")
    parser.add_argument("--mode", default="lstm",
                        help="relu, tanh, lstm, gru")
    parser.add_argument("-s", "--return-sequences", default=False,
                        action="store_true", dest="return_sequences")
    parser.add_argument("-d", "--bidirectional", default=False,
                        action="store_true", dest="bidirectional")
    parser.add_argument("-n", "--num-layers", default=2, type=int,
                        dest="num_layers")
    parser.add_argument("--seq-limit", default=50, type=int)
    parser.add_argument("--hidden", default=32, type=int)
    args = parser.parse_args(argv)

    imdb = imdb_data.load()
    embed = imdb["embed_weights"]
    seq_limit, hid = args.seq_limit, args.hidden
    dev = device.create_cuda_gpu_on(args.device_id) if device.get_num_gpus() \
        else device.get_default_device()

    tx = tensor.Tensor((args.bs, seq_limit, embed.shape[1]), dev,
                       tensor.float32)
    ty = tensor.Tensor((args.bs, 2), dev, tensor.float32)
    tx.gaussian(0, 1)
    ty.gaussian(0, 1)

    m = IMDBModel(hid, mode=args.mode, return_sequences=args.return_sequences,
                  bidirectional=args.bidirectional,
                  num_layers=args.num_layers)
    m.set_opt(opt.SGD(args.lr, 0.9))
    m.compile([tx], is_train=True, use_graph=True, sequential=False)

    x_train, y1h_train, y_train = imdb_data.pad_batch_2vec(
        imdb["train"], seq_limit, embed)
    x_test, y1h_test, y_test = imdb_data.pad_batch_2vec(
        imdb["val"], seq_limit, embed)

    def batches(x, y1h, y):
        for i in range(len(x) // args.bs):
            s = slice(i * args.bs, (i + 1) * args.bs)
            yield x[s], y1h[s], y[s]

    acc = 0.0
    m.train()
    for epoch in range(args.max_epoch):
        total, correct, trials, nb = 0.0, 0, 0, 0
        for xb, y1hb, yb in batches(x_train, y1h_train, y_train):
            tx.copy_from_numpy(xb)
            ty.copy_from_numpy(y1hb)
            out, loss = m(tx, ty)
            total += float(tensor.to_numpy(loss).reshape(-1)[0])
            correct += int((yb == np.argmax(tensor.to_numpy(out), 1)).sum())
            trials += len(yb)
            nb += 1
        print("epoch %d loss %s; acc %.3f" % (epoch, total / max(nb, 1),
                                              correct / max(trials, 1)))
    m.eval()
    correct, trials = 0, 0
    for xb, y1hb, yb in batches(x_test, y1h_test, y_test):
        tx.copy_from_numpy(xb)
        out = m(tx)
        correct += int((yb == np.argmax(tensor.to_numpy(out), 1)).sum())
        trials += len(yb)
    acc = correct / max(trials, 1)
    print("eval acc %.3f" % acc)
    return acc


if __name__ == "__main__":
    main()
