"""Restricted Boltzmann machine trained with one step of contrastive
divergence, written with raw tensor operations only (no autograd).
Counterpart of the reference's examples/rbm/train.py:48-126.

    python train.py --use_gpu -m 10
"""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402
import data as datasets  # noqa: E402


def load_train_data(file_path=None):
    """(train, valid) images as (N, 784) float32 in [0, 1]."""
    if file_path and file_path.endswith(".npz") and os.path.exists(file_path):
        z = np.load(file_path, allow_pickle=False)
        return z["train"].astype(np.float32), z["valid"].astype(np.float32)
    tx, _, vx, _ = datasets.load("mnist")

    def flat(a):
        a = a.reshape(a.shape[0], -1)
        return ((a - a.min()) / max(a.max() - a.min(), 1e-6)).astype(
            np.float32)

    return flat(tx), flat(vx)


def hidden_prob(v, w, hb):
    return tensor.sigmoid(tensor.mult(v, w) + hb)


def reconstruct(v, w, vb, hb, dev):
    """Sample the hidden units from v, return (p(h|v), p(v|h_sample))."""
    ph = hidden_prob(v, w, hb)
    rnd = tensor.Tensor(ph.shape, dev)
    rnd.uniform(0.0, 1.0)
    hsample = tensor.gt(ph, rnd)
    pv = tensor.sigmoid(tensor.mult(hsample, w.T()) + vb)
    return ph, pv


def train(data_file, use_gpu, num_epoch=10, batch_size=100, hdim=1000):
    print("Start intialization............")
    lr, weight_decay = 0.0005, 0.0002
    print("Loading data ..................")
    train_x, valid_x = load_train_data(data_file)
    vdim = train_x.shape[1]
    dev = device.create_cuda_gpu() if use_gpu and device.get_num_gpus() \
        else device.get_default_device()
    tweight = tensor.Tensor((vdim, hdim), dev)
    tweight.gaussian(0.0, 0.1)
    tvbias = tensor.from_numpy(np.zeros(vdim, dtype=np.float32), dev)
    thbias = tensor.from_numpy(np.zeros(hdim, dtype=np.float32), dev)
    sgd = opt.SGD(lr=lr, momentum=0.9, weight_decay=weight_decay)

    num_train_batch = train_x.shape[0] // batch_size
    print("num_train_batch = %d " % num_train_batch)
    errors = []
    for epoch in range(num_epoch):
        err_sum = 0.0
        print("Epoch %d" % epoch)
        for b in range(num_train_batch):
            tdata = tensor.from_numpy(
                train_x[b * batch_size:(b + 1) * batch_size], dev)
            # positive phase, one Gibbs step, negative phase
            tposhidprob, tnegdata = reconstruct(tdata, tweight, tvbias,
                                                thbias, dev)
            tneghidprob = hidden_prob(tnegdata, tweight, thbias)
            err_sum = tensor.sum(tensor.square(tdata - tnegdata)) + err_sum

            tgweight = tensor.mult(tnegdata.T(), tneghidprob) \
                - tensor.mult(tdata.T(), tposhidprob)
            tgvbias = tensor.sum(tnegdata, 0) - tensor.sum(tdata, 0)
            tghbias = tensor.sum(tneghidprob, 0) - tensor.sum(tposhidprob, 0)
            sgd.apply("w", tweight, tgweight)
            sgd.apply("vb", tvbias, tgvbias)
            sgd.apply("hb", thbias, tghbias)
            sgd.step()
        train_err = float(np.asarray(
            tensor.to_numpy(err_sum) if isinstance(err_sum, tensor.Tensor)
            else err_sum).reshape(-1)[0]) / train_x.shape[0]
        print("training erroraverage = %f" % train_err)

        tvalid = tensor.from_numpy(valid_x, dev)
        _, tvalidneg = reconstruct(tvalid, tweight, tvbias, thbias, dev)
        v = tensor.sum(tensor.square(tvalid - tvalidneg))
        valid_err = float(np.asarray(
            tensor.to_numpy(v) if isinstance(v, tensor.Tensor) else v)
            .reshape(-1)[0]) / valid_x.shape[0]
        print("valid erroraverage = %f" % valid_err)
        errors.append((train_err, valid_err))
    return errors


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train RBM over MNIST")
    parser.add_argument("file", type=str, nargs="?", default=None,
                        help="optional .npz with arrays train, valid")
    parser.add_argument("--use_gpu", action="store_true")
    parser.add_argument("-m", "--max-epoch", type=int, default=10)
    parser.add_argument("--hdim", type=int, default=1000)
    args = parser.parse_args()
    train(args.file, args.use_gpu, args.max_epoch, hdim=args.hdim)
