"""Layer-norm + GELU block of a BERT-base sized activation (32 x 256 x 768)
imported from ONNX: the exported 14-node chain versus the fused operators."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests"))

from singa_amd import device, sonnx, tensor  # noqa: E402
from test_onnx import _ln_gelu_chain_model  # noqa: E402


def main():
    dev = device.create_cuda_gpu()
    m, _, _ = _ln_gelu_chain_model(D=768)
    x = tensor.from_numpy(
        np.random.randn(32, 256, 768).astype(np.float32), dev)
    for fuse in (False, True):
        sonnx.SingaBackend.fuse_patterns = fuse
        rep = sonnx.prepare(m, device=dev)
        rep.is_graph = True          # keep outputs on the device
        for _ in range(5):
            rep.run([x])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            rep.run([x])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print("fuse_patterns=%-5s %2d nodes  %.3f ms per run" % (
            fuse, len(rep.layers), dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
