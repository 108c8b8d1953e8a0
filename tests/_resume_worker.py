"""One rank of the checkpoint / restart tests: data-parallel training of a
small MLP on the host, a checkpoint every other step, resuming from the
newest one.  Usage (ranks from the launcher's environment):
_resume_worker.py CKPT_DIR STEPS OUT_FILE"""
import json
import os
import sys

os.environ["CUDA_VISIBLE_DEVICES"] = ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from singa_amd import device, layer, model, opt, tensor  # noqa: E402
from singa_amd.utils.checkpoint import CheckpointManager  # noqa: E402


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
    ckpt_dir, steps, out_file = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank = int(os.environ.get("RANK", "0"))
    sgd = opt.DistOpt(opt.SGD(lr=0.1, momentum=0.9))
    dev = device.get_default_device()
    dev.SetRandSeed(7)                  # same initial weights on every attempt
    tx = tensor.Tensor((4, 6), dev)
    ty = tensor.Tensor((4,), dev, tensor.int32)
    m = Net()
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=False, sequential=True)
    manager = CheckpointManager(ckpt_dir, keep=2, rank=rank)
    resumed = manager.restore(m, sgd)
    start = resumed["step"] if resumed else 0
    print("rank %d attempt %s starts at step %d" % (
        rank, os.environ.get("SINGA_RESTART_COUNT", "0"), start), flush=True)
    losses = []
    for step in range(start, steps):
        rng = np.random.RandomState(1000 * step + rank)   # data of this step
        tx.copy_from_numpy(rng.randn(4, 6).astype(np.float32))
        ty.copy_from_numpy(rng.randint(0, 2, 4).astype(np.int32))
        _, loss = m(tx, ty)
        losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
        if (step + 1) % 2 == 0:
            manager.save(m, step + 1, sgd, {"last_loss": losses[-1]})
    if rank == 0:
        states = {k: tensor.to_numpy(v).tolist()
                  for k, v in m.get_states().items()}
        with open(out_file, "w") as f:
            json.dump({"start": start, "losses": losses, "states": states}, f)
    sgd.communicator.close()


if __name__ == "__main__":
    main()
