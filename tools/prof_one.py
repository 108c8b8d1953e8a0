"""Run one convolution pass a few times (for rocprofv3 --pmc / --kernel-trace).

Usage: python tools/prof_one.py C K R stride pad H pass [batch] [iters]
  pass: fwd | bwd_data | bwd_weight
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device as sdevice  # noqa: E402


def main():
    C, K, R, st, pd, H = [int(v) for v in sys.argv[1:7]]
    which = sys.argv[7]
    N = int(sys.argv[8]) if len(sys.argv) > 8 else 32
    iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    m = sdev.math
    P = (H + 2 * pd - R) // st + 1
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, R, R, device=dev) * 0.05
    y = torch.empty(N, K, P, P, device=dev)
    dy = torch.randn(N, K, P, P, device=dev)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    for _ in range(iters):
        if which == "fwd":
            m.conv_fwd(x, w, None, y, (st, st), (pd, pd), 1)
        elif which == "bwd_data":
            m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1)
        else:
            m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
