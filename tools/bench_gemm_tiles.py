"""float32 GEMM: every tile shape of the two-stage kernel family on every
benchmark shape, next to the single-stage kernel, the automatic choice and the
vendor library.

Usage: python tools/bench_gemm_tiles.py [--iters 10] [--no-vendor]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device as sdevice  # noqa: E402
from bench_conv import timeit  # noqa: E402
from bench_gemm import SHAPES, WGRAD_SHAPES  # noqa: E402

from singa_amd.core import math_hip  # noqa: E402

TILES_F32 = ["64x64", "128x64", "64x128", "128x128", "256x128", "256x96",
             "128x256", "1st128"]
TILES_H16 = ["128x128", "256x256", "256x128", "128x256", "64x128"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--dtype", default="float32")
    args = ap.parse_args()
    dt = getattr(torch, args.dtype)
    f32 = dt == torch.float32
    TILES = TILES_F32 if f32 else TILES_H16
    sdev = sdevice.create_cuda_gpu()
    dev, m = sdev.torch_device, sdev.math
    k = m.k
    set_tile = k.set_gemm_f32_tile if f32 else k.set_gemm_h16_tile
    cases = [("nn", s) for s in SHAPES] + [("tn", s) for s in WGRAD_SHAPES]
    print("%-22s %8s |" % ("TF/s", "single") +
          "".join(" %9s" % t for t in TILES) + " | %8s %8s %8s" % (
              "rule", "tuned", "vendor"))
    for kind, (M, N, K) in cases:
        if kind == "nn":
            a = torch.randn(M, K, device=dev, dtype=dt)
        else:
            a = torch.randn(K, M, device=dev, dtype=dt).t()
        b = torch.randn(K, N, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        ref = None
        row = []
        for tile in [-2 if f32 else -1] + list(range(len(TILES))) + \
                [-1, "tuned"]:
            # forced shapes and the launcher's rule run without measured
            # plans; the last column is what a training run gets
            math_hip._GEMM_AUTOTUNE = tile == "tuned"
            set_tile(-1 if tile == "tuned" else tile)
            t = timeit(lambda: m.gemm(1.0, a, b, 0.0, c), args.iters)
            row.append(2.0 * M * N * K / t / 1e9)
            if ref is None:
                ref = c.clone()
            else:
                err = (c.float() - ref.float()).abs().max().item()
                assert err <= (1e-3 if f32 else 2e-2) * \
                    ref.float().abs().max().item(), (tile, err)
        set_tile(-1)
        tv = 0.0
        if not args.no_vendor:
            tv = 2.0 * M * N * K / timeit(
                lambda: torch.matmul(a, b, out=c), args.iters) / 1e9
        print("%s %5dx%5dx%5d %8.1f |" % (kind, M, N, K, row[0]) +
              "".join(" %9.1f" % v for v in row[1:-2]) +
              " | %8.1f %8.1f %8.1f  %s" % (
                  row[-2], row[-1], tv,
                  [v for kk, v in m._gemm_plans.items()
                   if kk[2:5] == (M, N, K)]), flush=True)


if __name__ == "__main__":
    main()
