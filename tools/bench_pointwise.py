"""Times a 1x1 convolution (stride 1, no padding) on the implicit-GEMM
convolution kernels against the same products issued as batched GEMMs
(forward Y[n] = W X[n], backward-data dX[n] = W^T dY[n]).

    python tools/bench_pointwise.py [N C HW K]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from singa_amd import device
from singa_amd.core import math_hip


def main():
    N, C, HW, K = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 \
        else (16, 728, 361, 728)
    H = int(round(HW ** 0.5))
    assert H * H == HW, "HW must be a square"
    dev = device.create_cuda_gpu()
    hm = dev.math if hasattr(dev, "math") else math_hip.HipMath(dev)
    td = dev.torch_device
    x = torch.randn(N, C, H, H, device=td)
    w = torch.randn(K, C, 1, 1, device=td) * 0.05
    y = torch.empty(N, K, H, H, device=td)
    dy = torch.randn(N, K, H, H, device=td)
    dx = torch.empty_like(x)
    y2 = torch.empty_like(y)
    dx2 = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream

    def conv_f():
        hm.conv_fwd(x, w, None, y, (1, 1), (0, 0), 1)

    def conv_b():
        hm.conv_bwd_data(dy, w, dx, (1, 1), (0, 0), 1)

    def gemm_f():
        hm.k.gemm(0, K, HW, C, 1.0, w.data_ptr(), C, 1, 0, x.data_ptr(), HW, 1,
                  C * HW, 0.0, y2.data_ptr(), HW, K * HW, N, 0, 0, s)

    def gemm_b():
        hm.k.gemm(0, C, HW, K, 1.0, w.data_ptr(), 1, C, 0, dy.data_ptr(), HW,
                  1, K * HW, 0.0, dx2.data_ptr(), HW, C * HW, N, 0, 0, s)

    def timeit(fn):
        for _ in range(3):
            fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3

    flop = 2.0 * N * K * C * HW
    for name, fn in (("conv fwd", conv_f), ("gemm fwd", gemm_f),
                     ("conv bwd-data", conv_b), ("gemm bwd-data", gemm_b)):
        us = timeit(fn)
        print("%-14s %8.1f us  %6.1f TF/s" % (name, us, flop / us / 1e6),
              flush=True)
    print("max |y - y2| %.3g, max |dx - dx2| %.3g" %
          ((y - y2).abs().max().item(), (dx - dx2).abs().max().item()))


if __name__ == "__main__":
    main()
