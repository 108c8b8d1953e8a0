"""Run a few replayed ResNet-18 training steps from a fixed seed and print the
losses and a digest of the parameters as hexadecimal bit patterns, so that two
processes can be compared bit for bit.

    SINGA_DETERMINISTIC=1 python tools/check_determinism.py [steps]

With SINGA_DETERMINISTIC=1 the output of two runs must be identical; without
it the split reductions add with float atomics and the last bits drift.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    from singa_amd import device, models, opt, tensor
    dev = device.create_cuda_gpu_on(0)
    dev.SetRandSeed(1234)
    rs = np.random.RandomState(0)
    x = rs.randn(16, 3, 224, 224).astype(np.float32)
    y = rs.randint(0, 10, 16).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    np.random.seed(7)
    m = models.resnet.resnet18(num_classes=10)
    m.set_optimizer(opt.SGD(lr=0.05, momentum=0.9, weight_decay=1e-5))
    m.compile([tx], is_train=True, use_graph=True, sequential=False)
    losses = []
    for _ in range(steps + 2):          # eager, capture, then replays
        _, loss = m(tx, ty, "plain", None)
        losses.append(tensor.to_numpy(loss).astype(np.float32).reshape(-1)[0])
    digest = hashlib.sha256()
    for name, p in sorted(m.get_params().items()):
        digest.update(tensor.to_numpy(p).tobytes())
    print("LOSSES", " ".join("%08x" % v.view(np.uint32) for v in losses))
    print("VALUES", " ".join("%.9g" % v for v in losses))
    print("PARAMS", digest.hexdigest())


if __name__ == "__main__":
    main()
