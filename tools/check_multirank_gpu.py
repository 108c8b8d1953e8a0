"""Rehearsal of the multi-rank training path on ONE GPU: two ranks share the
device and exchange gradients through gloo (RCCL does not accept two ranks on
one GPU), everything else -- DistOpt buckets, pack / unpack kernels, side
streams, hipGraph segments with the collectives between them -- is the
production path.  Both ranks train on the same batch, so the averaged
gradient equals the single-process one and the loss trajectory must match a
plain single-process run.

    python tools/check_multirank_gpu.py            # launches the ranks itself
"""
import faulthandler
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# plain | half | partialUpdate | sparseTopK | sparseThreshold
OPTION = os.environ.get("MR_OPTION", "plain")
NRANKS = int(os.environ.get("MR_RANKS", "2"))
sys.path.insert(0, ROOT)


def train(dist, steps=6):
    from singa_amd import device, models, opt, tensor
    dev = device.create_cuda_gpu_on(0)
    dev.SetRandSeed(0)
    rs = np.random.RandomState(0)
    x = rs.randn(8, 3, 224, 224).astype(np.float32)
    y = rs.randint(0, 10, 8).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    sgd = opt.SGD(lr=0.05, momentum=0.9, weight_decay=1e-5)
    if dist:
        sgd = opt.DistOpt(sgd)
        sgd.threshold = 1 << 20          # several buckets
    m = models.resnet.resnet18(num_classes=10)
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True,
              use_graph=os.environ.get("MR_GRAPH", "1") == "1",
              sequential=dist)
    losses = []
    for i in range(steps):
        if dist:
            # a step stuck for a minute prints where, then the rank exits
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(60, exit=True)
        _, loss = m(tx, ty, OPTION if dist else "plain",
                    0.05 if OPTION.startswith("sparse") else None)
        losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
        print("  [%s] step %d loss %.5f" % (os.environ.get("RANK", "-"), i,
                                           losses[-1]), flush=True)
    faulthandler.cancel_dump_traceback_later()
    dev.EnableGraph(False)
    dev.ResetGraph()
    return losses


def main():
    if "RANK" in os.environ:             # a rank started by the launcher
        print("  [%s] started" % os.environ["RANK"], flush=True)
        losses = train(True)
        # one write: the ranks share the log file, and print() would hand the
        # text and the line break to it separately
        sys.stdout.write("rank %s %s\n" % (
            os.environ["RANK"], " ".join("%.5f" % v for v in losses)))
        sys.stdout.flush()
        return 0
    env = dict(os.environ, SINGA_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    log = os.path.join(ROOT, "gpurun_out", "multirank_ranks.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as fd:
        rc = subprocess.run(
            [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1",
             "--nproc-per-node=%d" % NRANKS, "--master-addr", "127.0.0.1",
             "--master-port", "29541", os.path.abspath(__file__)],
            env=env, stdout=fd, stderr=subprocess.STDOUT,
            timeout=200).returncode
    text = open(log).read()
    print(text[-3000:])
    if rc != 0:
        return 1
    ranks = [[float(v) for v in m.group(1).split()] for m in re.finditer(
        r"rank \d+((?: -?(?:\d+\.\d+|nan|inf))+)", text)]
    assert len(ranks) == NRANKS, ranks
    ref = train(False)
    print("single", " ".join("%.5f" % v for v in ref))
    for r in ranks:
        if OPTION == "plain":
            np.testing.assert_allclose(r, ref, rtol=2e-3, atol=2e-3)
        elif OPTION == "half":      # fp16 on the wire
            np.testing.assert_allclose(r, ref, rtol=5e-2, atol=5e-3)
        else:                       # different algorithms: must still train
            assert np.all(np.isfinite(r)) and r[-1] < r[0], r
        if OPTION != "partialUpdate":   # (there ranks drift apart by design)
            # split reductions use float atomics: ranks agree to rounding only
            np.testing.assert_allclose(r, ranks[0], rtol=2e-2, atol=1e-3)
    print("MULTIRANK-OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
