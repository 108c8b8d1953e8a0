"""Loss trajectories of plain SGD vs DistOpt on a single rank (the all-reduce
is then the identity, so they must agree step by step), eager and graph."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device, models, opt, tensor  # noqa: E402


def run(dist, graph, steps=6, depth=18, bs=8, bucket=None):
    dev = device.create_cuda_gpu() if device.get_num_gpus() else \
        device.get_default_device()
    dev.SetRandSeed(0)
    rs = np.random.RandomState(0)
    x = rs.randn(bs, 3, 224, 224).astype(np.float32)
    y = rs.randint(0, 10, bs).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    sgd = opt.SGD(lr=0.05, momentum=0.9, weight_decay=1e-5)
    if dist:
        sgd = opt.DistOpt(sgd)
        if bucket:
            sgd.threshold = bucket
    m = getattr(models.resnet, "resnet%d" % depth)(num_classes=10)
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=graph, sequential=dist)
    losses = []
    for _ in range(steps):
        out, loss = m(tx, ty, "plain", None)
        losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    dev.EnableGraph(False)
    dev.ResetGraph()
    return losses


if __name__ == "__main__":
    # one configuration per process: "plain|dist" "eager|graph" [depth] [bs]
    os.environ.setdefault("SINGA_DIST_FORCE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist = sys.argv[1] == "dist"
    graph = sys.argv[2] == "graph"
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 18
    bs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    bucket = int(sys.argv[5]) if len(sys.argv) > 5 else None
    got = run(dist, graph, depth=depth, bs=bs, bucket=bucket)
    print("%-5s %-5s resnet%d bs%d" % (sys.argv[1], sys.argv[2], depth, bs),
          ["%.4f" % v for v in got], flush=True)
