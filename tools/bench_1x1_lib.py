"""1x1 stride-1 convolutions are plain GEMMs: compare the implicit-GEMM conv
kernel with the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) on
the ResNet-50 shapes, to decide whether dispatching them to the library pays.

fwd:      Y[n] (K x HW) = W (K x C)   @ X[n]  (C x HW)
bwd data: dX[n] (C x HW) = W^T (C x K) @ dY[n] (K x HW)
bwd wgt:  dW (K x C) = sum_n dY[n] (K x HW) @ X[n]^T (HW x C)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device as sdevice  # noqa: E402
from tools.bench_conv import resnet50_layers, timeit  # noqa: E402


def main():
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    m = sdev.math
    N = 32
    print("%-26s | %-23s | %-23s | %-23s" % ("C,K,H", "fwd conv/lib ms",
                                              "bwdD conv/lib ms",
                                              "bwdW conv/lib ms"))
    tot = [0.0] * 6
    for (C, K, R, s, p, H, cnt) in resnet50_layers():
        if R != 1 or s != 1:
            continue
        HW = H * H
        x = torch.randn(N, C, H, H, device=dev)
        w = torch.randn(K, C, 1, 1, device=dev)
        dy = torch.randn(N, K, H, H, device=dev)
        y = torch.empty(N, K, H, H, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        w2 = w.view(K, C)
        t = []
        t.append(timeit(lambda: m.conv_fwd(x, w, None, y, (1, 1), (0, 0), 1), 20))
        t.append(timeit(lambda: torch.matmul(w2, x.view(N, C, HW),
                                             out=y.view(N, K, HW)), 20))
        t.append(timeit(lambda: m.conv_bwd_data(dy, w, dx, (1, 1), (0, 0), 1), 20))
        t.append(timeit(lambda: torch.matmul(w2.t(), dy.view(N, K, HW),
                                             out=dx.view(N, C, HW)), 20))
        t.append(timeit(lambda: m.conv_bwd_weight(dy, x, dw, (1, 1), (0, 0), 1), 20))
        # library form: one GEMM over the (n, hw) axis after making it the
        # contraction dim of 2-d operands: dW = dY2 (K x N*HW) @ X2^T with
        # dY2 = dy.transpose(0,1).reshape(K, N*HW) (a copy) -- count the copy
        def lib_w():
            dy2 = dy.view(N, K, HW).transpose(0, 1).reshape(K, N * HW)
            x2 = x.view(N, C, HW).transpose(0, 1).reshape(C, N * HW)
            torch.matmul(dy2, x2.t(), out=dw.view(K, C))
        t.append(timeit(lib_w, 20))
        for i in range(6):
            tot[i] += t[i] * cnt
        print("%-26s | %9.3f / %9.3f | %9.3f / %9.3f | %9.3f / %9.3f" %
              ("%d,%d,%d x%d" % (C, K, H, cnt), *t), flush=True)
    print("totals (ms/step): fwd %.2f / %.2f   bwdD %.2f / %.2f   bwdW %.2f / %.2f"
          % tuple(tot))


if __name__ == "__main__":
    main()
