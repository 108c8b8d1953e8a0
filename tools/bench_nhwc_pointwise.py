"""What a channels-last layout would buy the 16-bit 1x1 convolutions of
ResNet-50: in NHWC every pass of such a layer is ONE plain matrix product
with the channel index contiguous,

    forward          y  [NHW x K] = x  [NHW x C] * w^T [C x K]
    data gradient    dx [NHW x C] = dy [NHW x K] * w   [K x C]
    weight gradient  dw [K x C]   = dy^T [K x NHW] * x [NHW x C]

which the tiled 16-bit GEMM kernel runs as it is.  Timed against the NCHW
kernels the framework uses today, layer by layer (each call replayed from a
hipGraph).

    python tools/bench_nhwc_pointwise.py [--batch 32] [--dtype float16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device as sdevice  # noqa: E402
from bench_conv import resnet50_layers, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="float16")
    args = ap.parse_args()
    sdev = sdevice.create_cuda_gpu()
    dev, m = sdev.torch_device, sdev.math
    dt = getattr(torch, args.dtype)
    N = args.batch
    tot = {"nchw": [0.0, 0.0, 0.0], "nhwc": [0.0, 0.0, 0.0]}
    print("%-22s | %-26s | %-26s" % ("C,K,H x count", "NCHW fwd / dgrad / wgrad ms",
                                    "NHWC GEMM fwd / dgrad / wgrad ms"))
    for (C, K, R, st, pd, H, cnt) in resnet50_layers():
        if R != 1 or st != 1:
            continue
        x = torch.randn(N, C, H, H, device=dev, dtype=dt)
        w = (torch.randn(K, C, 1, 1, device=dev) * 0.05).to(dt)
        y = torch.empty(N, K, H, H, device=dev, dtype=dt)
        dy = torch.randn(N, K, H, H, device=dev, dtype=dt)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        a = [timeit(lambda: m.conv_fwd(x, w, None, y, (1, 1), (0, 0), 1), 10),
             timeit(lambda: m.conv_bwd_data(dy, w, dx, (1, 1), (0, 0), 1), 10),
             timeit(lambda: m.conv_bwd_weight(dy, x, dw, (1, 1), (0, 0), 1),
                    10)]
        P = N * H * H
        xl = torch.randn(P, C, device=dev, dtype=dt)
        dyl = torch.randn(P, K, device=dev, dtype=dt)
        yl = torch.empty(P, K, device=dev, dtype=dt)
        dxl = torch.empty(P, C, device=dev, dtype=dt)
        w2, dw2 = w.view(K, C), torch.empty(K, C, device=dev, dtype=dt)
        b = [timeit(lambda: m.gemm(1.0, xl, w2.t(), 0.0, yl), 10),
             timeit(lambda: m.gemm(1.0, dyl, w2, 0.0, dxl), 10),
             timeit(lambda: m.gemm(1.0, dyl.t(), xl, 0.0, dw2), 10)]
        print("%-22s | %7.3f %7.3f %7.3f    | %7.3f %7.3f %7.3f" % (
            "%d,%d,%d x%d" % (C, K, H, cnt), a[0], a[1], a[2], b[0], b[1],
            b[2]), flush=True)
        for i in range(3):
            tot["nchw"][i] += a[i] * cnt
            tot["nhwc"][i] += b[i] * cnt
    print("sum over the 1x1 stride-1 layers of a step: NCHW %.3f + %.3f + %.3f"
          " = %.3f ms, NHWC %.3f + %.3f + %.3f = %.3f ms" % (
              tot["nchw"][0], tot["nchw"][1], tot["nchw"][2],
              sum(tot["nchw"]), tot["nhwc"][0], tot["nhwc"][1],
              tot["nhwc"][2], sum(tot["nhwc"])))


if __name__ == "__main__":
    main()
