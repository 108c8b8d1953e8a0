"""Per-layer timing of the batch-norm kernels on ResNet-50 shapes (batch 32):
time and effective bandwidth against the bytes each pass has to move
(forward: read x [+ residual], write y; backward: read dy, x or y, write dx
[+ residual gradient]).

    python tools/bench_bn.py [--batch 32] [--dtype float32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device as sdevice  # noqa: E402

# (channels, H, act, count): act 0 none (downsample), 1 ReLU, 2 add + ReLU
def resnet50_bn():
    out = [(64, 112, 1, 1)]
    for cin, mid, cout, h, blocks in ((64, 64, 256, 56, 3),
                                      (256, 128, 512, 28, 4),
                                      (512, 256, 1024, 14, 6),
                                      (1024, 512, 2048, 7, 3)):
        hin = h if cin == 64 else 2 * h
        out.append((mid, hin, 1, 1))          # first block, conv1 (input size)
        out.append((mid, h, 1, 2 * blocks - 1))
        out.append((cout, h, 2, blocks))
        out.append((cout, h, 0, 1))           # downsample branch
    return out


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="float32")
    args = ap.parse_args()
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    m = sdev.math
    dt = getattr(torch, args.dtype)
    N = args.batch
    sz = torch.tensor([], dtype=dt).element_size()
    tot_f = tot_b = floor_f = floor_b = 0.0
    print("%-22s %9s %8s | %9s %8s" % ("C,H,act xcount", "fwd ms", "GB/s",
                                       "bwd ms", "GB/s"))
    for C, H, act, cnt in resnet50_bn():
        x = torch.randn(N, C, H, H, device=dev, dtype=dt)
        y = torch.empty_like(x)
        dy = torch.randn_like(x)
        dx = torch.empty_like(x)
        res = torch.randn_like(x) if act == 2 else None
        dres = torch.empty_like(x) if act == 2 else None
        scale = torch.ones(C, device=dev)
        bias = torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, inv = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ds, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        t_f = timeit(lambda: m.bn_fwd_train(x, scale, bias, rm, rv, 0.1, 1e-5,
                                            y, mean, inv, act, res))
        t_b = timeit(lambda: m.bn_bwd(dy, x, scale, mean, inv, dx, ds, db,
                                      act, bias, y if act == 2 else None,
                                      dres))
        n = x.numel() * sz
        bytes_f = n * (3 if act == 2 else 2)
        bytes_b = n * (4 if act == 2 else 3)
        print("%-22s %9.4f %8.0f | %9.4f %8.0f" % (
            "%d,%d,%d x%d" % (C, H, act, cnt), t_f, bytes_f / t_f / 1e6, t_b,
            bytes_b / t_b / 1e6), flush=True)
        tot_f += t_f * cnt
        tot_b += t_b * cnt
        floor_f += bytes_f * cnt
        floor_b += bytes_b * cnt
    print("total ms/step: fwd %.3f bwd %.3f sum %.3f; bytes %.2f GB -> %.3f ms "
          "at 5.5 TB/s" % (tot_f, tot_b, tot_f + tot_b,
                           (floor_f + floor_b) / 1e9,
                           (floor_f + floor_b) / 5.5e9))


if __name__ == "__main__":
    main()
