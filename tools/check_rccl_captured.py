"""One-GPU check that RCCL collectives are part of the captured training step.

A single-rank RCCL process group is forced (SINGA_DIST_FORCE=1), so every
all-reduce / all-gather of DistOpt really goes through RCCL on its side
stream.  ResNet-18 is trained in graph mode through DistOpt; the step must
replay as ONE hipGraph (collective nodes inside it) and follow the loss
trajectory of a plain eager single-process run (a one-rank sum is the
identity).  With a verbosity the per-node device time stamps are captured too
and the communication profile is printed from the same production replay.

    python tools/check_rccl_captured.py [option] [dtype] [verbosity] [failcapture]
    option: plain | half | partialUpdate | sparseTopK | sparseThreshold
    dtype: fp32 | fp16

With ``failcapture`` the first capture of the step is made to fail: the device
must fall back to hipGraph segments around eagerly issued collectives without
running any collective twice (parameters averaged twice, residuals consumed),
i.e. the loss trajectory must still be the eager one.
"""
import os
import sys

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
os.environ["SINGA_DIST_FORCE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OPTION = sys.argv[1] if len(sys.argv) > 1 else "plain"
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "fp32"
VERBOSITY = int(sys.argv[3]) if len(sys.argv) > 3 else 0
FAIL_CAPTURE = len(sys.argv) > 4 and sys.argv[4] == "failcapture"


def train(dist, graph, steps=7):
    from singa_amd import device, models, opt, tensor
    dev = device.create_cuda_gpu_on(0)
    dev.SetRandSeed(0)
    dev.SetVerbosity(VERBOSITY if dist else 0)
    dev.SetSkipIteration(3)
    rs = np.random.RandomState(0)
    x = rs.randn(8, 3, 224, 224).astype(np.float32)
    y = rs.randint(0, 10, 8).astype(np.int32)
    dt = tensor.float16 if DTYPE == "fp16" else tensor.float32
    tx = tensor.Tensor(device=dev, data=x.astype(np.float16)
                       if DTYPE == "fp16" else x)
    assert tx.dtype == dt
    ty = tensor.Tensor(device=dev, data=y)
    sgd = opt.SGD(lr=0.05, momentum=0.9, weight_decay=1e-5)
    if dist:
        sgd = opt.DistOpt(sgd)
        sgd.threshold = 1 << 20          # several buckets
    if dist and FAIL_CAPTURE:
        from singa_amd.core import device as core_device
        core_device._HipGraphReplay.fail_next = True
    np.random.seed(7)
    m = models.resnet.resnet18(num_classes=10)
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=graph, sequential=dist)
    losses = []
    for i in range(steps):
        _, loss = m(tx, ty, OPTION if dist else "plain",
                    1.0 if OPTION == "sparseTopK" else
                    (0.0 if OPTION == "sparseThreshold" else None))
        losses.append(float(tensor.to_numpy(
            loss.as_type(tensor.float32)).reshape(-1)[0]))
    info = {"segments": dev.hipgraph_segments(),
            "steps": len(dev._replay.steps) if dev._replay else 0}
    if dist and VERBOSITY:
        dev.PrintTimeProfiling()
    dev.EnableGraph(False)
    dev.ResetGraph()
    return losses, info


def main():
    from singa_amd import tensor
    got, info = train(True, True)
    print("captured", " ".join("%.5f" % v for v in got), info, flush=True)
    if FAIL_CAPTURE:
        assert info["segments"] > 1 and info["steps"] > info["segments"], info
    else:
        assert info["segments"] == 1 and info["steps"] == 1, info
    ref, _ = train(False, False)
    print("eager   ", " ".join("%.5f" % v for v in ref), flush=True)
    if OPTION == "half" or DTYPE == "fp16":
        np.testing.assert_allclose(got, ref, rtol=6e-2, atol=2e-2)
    else:
        # split reductions use float atomics: equal up to summation order
        np.testing.assert_allclose(got, ref, rtol=5e-3, atol=5e-3)
    print("RCCL-CAPTURED-OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
