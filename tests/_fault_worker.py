"""One rank of the failure-detection tests: a few data-parallel steps of a
small MLP on the host over gloo.  Usage: _fault_worker.py RANK WORLD PORT
STEPS GRAPH; failure detection and the injected fault come from the
environment (SINGA_HEARTBEAT, SINGA_FAULT_INJECT, ...)."""
import os
import sys

os.environ["CUDA_VISIBLE_DEVICES"] = ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from singa_amd import device, layer, model, opt, tensor  # noqa: E402
from singa_amd import singa_wrap as singa  # noqa: E402
from singa_amd.parallel import health  # noqa: E402


class Net(model.Model):

    def __init__(self):
        super(Net, self).__init__()
        self.fc1, self.act, self.fc2 = layer.Linear(8), layer.ReLU(), \
            layer.Linear(2)
        self.loss = layer.SoftMaxCrossEntropy()

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))

    def train_one_batch(self, x, y):
        out = self.forward(x)
        loss = self.loss(out, y)
        self.optimizer(loss)
        return out, loss

    def set_optimizer(self, o):
        self.optimizer = o


def main():
    rank, world, port, steps, graph = (int(v) for v in sys.argv[1:6])
    holder = singa.NcclIdHolder(port=port)
    sgd = opt.DistOpt(opt.SGD(lr=0.1), nccl_id=holder, local_rank=rank,
                      world_size=world)
    dev = device.get_default_device()
    rng = np.random.RandomState(rank)
    tx = tensor.Tensor((4, 6), dev)
    ty = tensor.Tensor((4,), dev, tensor.int32)
    m = Net()
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=bool(graph), sequential=True)
    try:
        for i in range(steps):
            tx.copy_from_numpy(rng.randn(4, 6).astype(np.float32))
            ty.copy_from_numpy(rng.randint(0, 2, 4).astype(np.int32))
            m(tx, ty)
            print("rank %d step %d done (communicator counts %d)" % (
                rank, i, sgd.communicator.steps), flush=True)
    except health.RankFailure as e:
        print("rank %d caught RankFailure: %s" % (rank, e), flush=True)
        os._exit(3)
    sgd.communicator.close()
    print("rank %d finished" % rank, flush=True)


if __name__ == "__main__":
    main()
