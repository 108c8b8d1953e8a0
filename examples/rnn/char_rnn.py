"""Character-level language model: an LSTM unrolled over per-timestep one-hot
inputs, trained with graph buffering (counterpart of the reference's
examples/rnn/char_rnn.py:39-258).

    python char_rnn.py corpus.txt -b 32 -l 64 -d 128 -m 50
    python char_rnn.py                    # trains on this file's own text

One-hot encoding of a batch is a single fancy-index assignment; sampling is
driven by ``--sample N --seed TEXT`` instead of prompting on stdin.
"""
import argparse
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, "..", "..")]

from singa_amd import autograd, device, layer, model, opt, tensor  # noqa: E402


class CharRNN(model.Model):
    """LSTM over one-hot characters followed by a linear read-out of every
    time step.  The recurrent state lives in ``self.state`` (name -> tensor of
    shape (batch, hidden)) and is carried from one call to the next; the fused
    sequence operator behind ``layer.LSTM`` runs the whole unrolled recurrence
    as one node of the recorded graph."""

    STATE_NAMES = ("h", "c")

    def __init__(self, vocab_size, hidden_size=32, lr=0.01):
        super(CharRNN, self).__init__()
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.rnn = layer.LSTM(vocab_size, hidden_size)
        self.readout = layer.Linear(hidden_size, vocab_size)
        self.stack = layer.Cat()
        self.fold_steps = layer.Reshape()
        self.fold_labels = layer.Reshape()
        self.loss_fn = layer.SoftMaxCrossEntropy()
        self.set_optimizer(opt.SGD(lr))
        self.state = {}

    # the pair handed to layer.LSTM
    @property
    def hx(self):
        return self.state["h"]

    @property
    def cx(self):
        return self.state["c"]

    def new_state(self, batch, dev):
        """Zero state for sequences of ``batch`` rows on ``dev``."""
        self.state = {}
        for name in self.STATE_NAMES:
            t = tensor.Tensor((batch, self.hidden_size), dev)
            t.set_value(0.0)
            t.name = "state." + name
            self.state[name] = t

    def reset_states(self, dev):
        """Start a new sequence: zero the carried state in place (the
        recorded step keeps reading the same tensors)."""
        for t in self.state.values():
            t.to_device(dev)
            t.set_value(0.0)

    def initialize(self, inputs):
        self.new_state(inputs[0].shape[0], inputs[0].device)

    def forward(self, inputs):
        steps, h, c = self.rnn(inputs, (self.hx, self.cx))
        for t, new in zip((self.hx, self.cx), (h, c)):
            t.copy_data(new)
        rows = self.fold_steps(self.stack(steps), (-1, self.hidden_size))
        return self.readout(rows)

    def train_one_batch(self, x, y):
        logits = self.forward(x)
        loss = self.loss_fn(logits, self.fold_labels(y, (-1, 1)))
        self.optimizer(loss)
        return logits, loss

    def get_states(self):
        states = super().get_states()
        states.update((t.name, t) for t in self.state.values())
        return states

    def set_states(self, states):
        for t in self.state.values():
            t.copy_from(states[t.name])
        super().set_states(states)


class Data(object):
    """Plain text file cut into (seq_length + 1)-character samples."""

    def __init__(self, fpath, batch_size=32, seq_length=100, train_ratio=0.8):
        self.raw_data = open(fpath, "r", encoding="iso-8859-1").read()
        chars = sorted(set(self.raw_data))
        self.vocab_size = len(chars)
        self.char_to_idx = {ch: i for i, ch in enumerate(chars)}
        self.idx_to_char = {i: ch for i, ch in enumerate(chars)}
        data = np.fromiter((self.char_to_idx[c] for c in self.raw_data),
                           dtype=np.int32, count=len(self.raw_data))
        nsamples = len(data) // (1 + seq_length)
        data = data[:nsamples * (1 + seq_length)].reshape(-1, seq_length + 1)
        np.random.shuffle(data)
        ntrain = int(data.shape[0] * train_ratio)
        self.train_dat, self.val_dat = data[:ntrain], data[ntrain:]
        self.num_train_batch = self.train_dat.shape[0] // batch_size
        self.num_test_batch = self.val_dat.shape[0] // batch_size
        print("train dat", self.train_dat.shape)
        print("val dat", self.val_dat.shape)


def one_hot(idx, vocab_size):
    out = np.zeros(idx.shape + (vocab_size,), dtype=np.float32)
    np.put_along_axis(out, idx[..., None].astype(np.int64), 1.0, axis=-1)
    return out


def convert(batch, batch_size, seq_length, vocab_size, dev, inputs=None,
            labels=None):
    """(batch, seq+1) char ids -> list of seq one-hot tensors (batch, vocab)
    and a (seq*batch, 1) label tensor, reusing ``inputs`` / ``labels``."""
    x = one_hot(batch[:, :seq_length], vocab_size).swapaxes(0, 1)
    y = np.ascontiguousarray(batch[:, 1:].swapaxes(0, 1)).reshape(-1, 1)
    if labels is None:
        labels = tensor.from_numpy(y.astype(np.int32), dev)
    else:
        labels.copy_from_numpy(y.astype(np.int32))
    if inputs is None:
        inputs = [tensor.from_numpy(np.ascontiguousarray(x[t]), dev)
                  for t in range(seq_length)]
    else:
        for t in range(seq_length):
            inputs[t].copy_from_numpy(np.ascontiguousarray(x[t]))
    return inputs, labels


def sample(model, data, dev, seed, nsamples=100, use_max=False):
    """Continue ``seed`` for ``nsamples`` characters; returns the text."""
    model.eval()
    model.graph(False, False)
    seed = [c for c in seed if c in data.char_to_idx] or \
        [data.idx_to_char[0]]
    ids = np.array([[data.char_to_idx[c]] for c in seed], np.int32)
    inputs = [tensor.from_numpy(one_hot(i, data.vocab_size), dev)
              for i in ids]
    # generation runs with batch 1: fresh state tensors of that shape
    model.new_state(1, dev)
    outputs = model(inputs)
    text = list(seed)
    prob = tensor.to_numpy(tensor.softmax(outputs))[-1]
    for _ in range(nsamples):
        prob = prob.astype(np.float64)
        prob /= prob.sum()
        cur = int(np.argmax(prob)) if use_max else \
            int(np.random.choice(data.vocab_size, 1, p=prob)[0])
        text.append(data.idx_to_char[cur])
        tx = tensor.from_numpy(one_hot(np.array([cur]), data.vocab_size), dev)
        prob = tensor.to_numpy(tensor.softmax(model([tx])))[-1]
    return "".join(text)


def evaluate(model, data, batch_size, seq_length, dev, inputs, labels):
    model.eval()
    val_loss = 0.0
    for b in range(data.num_test_batch):
        batch = data.val_dat[b * batch_size:(b + 1) * batch_size]
        inputs, labels = convert(batch, batch_size, seq_length,
                                 data.vocab_size, dev, inputs, labels)
        model.reset_states(dev)
        y = model(inputs)
        loss = autograd.softmax_cross_entropy(y, labels)
        val_loss += float(tensor.to_numpy(loss).reshape(-1)[0])
    denom = max(data.num_test_batch, 1)
    print("            validation loss is %f" % (val_loss / denom))
    return val_loss / denom


def train(data, max_epoch, hidden_size=100, seq_length=100, batch_size=16,
          lr=0.01, use_graph=True, max_batches=None):
    dev = device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()
    model = CharRNN(data.vocab_size, hidden_size, lr)
    model.graph(use_graph, False)
    inputs, labels = None, None
    losses = []
    nb = data.num_train_batch if max_batches is None else \
        min(max_batches, data.num_train_batch)
    for epoch in range(max_epoch):
        model.train()
        train_loss = 0.0
        for b in range(nb):
            batch = data.train_dat[b * batch_size:(b + 1) * batch_size]
            inputs, labels = convert(batch, batch_size, seq_length,
                                     data.vocab_size, dev, inputs, labels)
            out, loss = model(inputs, labels)
            model.reset_states(dev)
            train_loss += float(tensor.to_numpy(loss).reshape(-1)[0])
        losses.append(train_loss / max(nb, 1))
        print("\nEpoch %d, train loss is %f" % (epoch, losses[-1]))
        evaluate(model, data, batch_size, seq_length, dev, inputs, labels)
    return model, dev, losses


if __name__ == "__main__":
    parser = argparse.ArgumentParser(
        description="Train an LSTM character model on a plain text file")
    parser.add_argument("data", type=str, nargs="?",
                        default=os.path.abspath(__file__),
                        help="training text file")
    parser.add_argument("-b", type=int, default=32, help="batch_size")
    parser.add_argument("-l", type=int, default=64, help="sequence length")
    parser.add_argument("-d", type=int, default=128, help="hidden size")
    parser.add_argument("-m", type=int, default=50, help="max num of epoch")
    parser.add_argument("--lr", type=float, default=0.01)
    parser.add_argument("--sample", type=int, default=0,
                        help="characters to generate after training")
    parser.add_argument("--seed", type=str, default="def ")
    args = parser.parse_args()
    data = Data(args.data, batch_size=args.b, seq_length=args.l)
    m, dev, _ = train(data, args.m, hidden_size=args.d, seq_length=args.l,
                      batch_size=args.b, lr=args.lr)
    if args.sample:
        print(sample(m, data, dev, args.seed, args.sample))
