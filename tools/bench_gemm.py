"""GEMM kernel throughput on square and transformer-shaped problems.

Usage: python tools/bench_gemm.py [--dtype float32|float16|bfloat16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device as sdevice  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 1024),
          (1024, 1024, 1024), (1024, 1024, 4096), (2048, 1024, 2048), (32, 1000, 2048),
          # BERT-base, 32 x 256 tokens: QKV / out projection, FFN
          (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072)]
# weight gradients of the same layers: x^T (K x M tokens, read transposed) @ dy
WGRAD_SHAPES = [(768, 768, 8192), (768, 3072, 8192), (3072, 768, 8192)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--vendor", action="store_true",
                    help="also time the vendor BLAS through torch.matmul")
    args = ap.parse_args()
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    dt = getattr(torch, args.dtype)
    m = sdev.math
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(K, N, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: m.gemm(1.0, a, b, 0.0, c), args.iters)
        line = "gemm %s %5dx%5dx%5d: %8.3f ms %7.1f TF/s" % (
            args.dtype, M, N, K, t, 2.0 * M * N * K / t / 1e9)
        if args.vendor:
            tv = timeit(lambda: torch.matmul(a, b, out=c), args.iters)
            line += "   vendor %8.3f ms %7.1f TF/s" % (
                tv, 2.0 * M * N * K / tv / 1e9)
        print(line, flush=True)
    wgrad(args, sdev)


def wgrad(args, sdev):
    dev, m = sdev.torch_device, sdev.math
    dt = getattr(torch, args.dtype)
    for (M, N, K) in WGRAD_SHAPES:
        x = torch.randn(K, M, device=dev, dtype=dt)     # tokens x features
        dy = torch.randn(K, N, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: m.gemm(1.0, x.t(), dy, 0.0, c), args.iters)
        line = "wgrad %s %5dx%5dx%5d: %8.3f ms %7.1f TF/s" % (
            args.dtype, M, N, K, t, 2.0 * M * N * K / t / 1e9)
        if args.vendor:
            tv = timeit(lambda: torch.matmul(x.t(), dy, out=c), args.iters)
            line += "   vendor %8.3f ms %7.1f TF/s" % (
                tv, 2.0 * M * N * K / tv / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
