"""Federated-learning client: trains an MLP locally for one epoch per round,
pushes its weights, pulls the average (counterpart of the reference's
examples/hfl/src/client.py)."""
import os
import sys

import numpy as np

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402
from singa_amd.models import mlp  # noqa: E402

from . import bank, wire  # noqa: E402
from .server import parseargs  # noqa: E402


class Client(object):

    def __init__(self, global_rank=0, host="127.0.0.1", port=1234):
        self.global_rank = global_rank
        self.host, self.port = host, int(port)
        self.sock = None
        self.weights = {}

    def start(self):
        self.sock = wire.connect(self.host, self.port)
        wire.send_int(self.sock, self.global_rank)
        print("[Client %d] Connect to %s:%s" % (self.global_rank, self.host,
                                                self.port), flush=True)

    def close(self):
        self.sock.close()

    def pull(self):
        op, self.weights = wire.receive_weights(self.sock)
        assert op == wire.SCATTER

    def push(self):
        wire.send_weights(self.sock, wire.GATHER, self.weights)


def get_weights(model):
    return {k: tensor.to_numpy(v) for k, v in model.get_states().items()}


def set_weights(model, weights, dev):
    model.set_states({k: tensor.from_numpy(v, dev)
                      for k, v in weights.items()})


def accuracy(pred, target):
    return int((np.argmax(pred, 1) == target).sum())


def run(args):
    dev = device.get_default_device()
    tx_, ty_, vx_, vy_ = bank.load(args.rank, args.num_clients)
    bs = args.batch_size
    model = mlp.create_model(data_size=tx_.shape[1], perceptron_size=32,
                             num_classes=2)
    tx = tensor.Tensor((bs, tx_.shape[1]), dev)
    ty = tensor.Tensor((bs,), dev, tensor.int32)
    model.set_optimizer(opt.SGD(lr=args.lr, momentum=0.9))
    model.compile([tx], is_train=True, use_graph=False, sequential=False)

    client = Client(args.rank, args.host, args.port)
    client.start()
    acc = 0.0
    for epoch in range(args.max_epoch):
        if epoch > 0:
            client.pull()
            set_weights(model, client.weights, dev)
        model.train()
        idx = np.random.permutation(len(tx_))
        for b in range(len(tx_) // bs):
            sel = idx[b * bs:(b + 1) * bs]
            tx.copy_from_numpy(tx_[sel])
            ty.copy_from_numpy(ty_[sel])
            model(tx, ty)
        model.eval()
        correct = 0
        nvb = len(vx_) // bs
        for b in range(nvb):
            tx.copy_from_numpy(vx_[b * bs:(b + 1) * bs])
            correct += accuracy(tensor.to_numpy(model(tx)),
                                vy_[b * bs:(b + 1) * bs])
        acc = correct / max(nvb * bs, 1)
        print("[Client %d] epoch %d eval acc %.3f" % (args.rank, epoch, acc),
              flush=True)
        client.weights = get_weights(model)
        client.push()
    client.close()
    return acc


if __name__ == "__main__":
    run(parseargs())
