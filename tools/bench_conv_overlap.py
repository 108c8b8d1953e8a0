"""Backward of one convolution layer: data gradient and weight gradient issued
back to back on one stream versus the weight gradient forked onto a second
stream (joined afterwards), over the ResNet-50 layer shapes.

Usage: python tools/bench_conv_overlap.py [--batch 32] [--dtype float32]
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
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    dt = getattr(torch, args.dtype)
    m = sdev.math
    N = args.batch
    side = torch.cuda.Stream(dev)
    tot_seq = tot_fork = 0.0
    print("%-30s %10s %10s" % ("layer C,K,R,s,p,H xcount", "serial ms",
                               "forked ms"))
    for li, (C, K, R, st, pd, H, cnt) in enumerate(resnet50_layers()):
        if li == 0:
            continue
        P = (H + 2 * pd - R) // st + 1
        x = torch.randn(N, C, H, H, device=dev, dtype=dt)
        w = torch.randn(K, C, R, R, device=dev, dtype=dt) * 0.05
        dy = torch.randn(N, K, P, P, device=dev, dtype=dt)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)

        def serial():
            m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1)
            m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1)

        def forked():
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1)
            m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1)
            cur.wait_stream(side)

        serial()  # autotune outside the timed region
        t_s = timeit(serial, args.iters)
        t_f = timeit(forked, args.iters)
        print("%-30s %10.3f %10.3f" % (
            "%d,%d,%d,%d,%d,%d x%d" % (C, K, R, st, pd, H, cnt), t_s, t_f),
            flush=True)
        tot_seq += t_s * cnt
        tot_fork += t_f * cnt
    print("total ms/step: serial %.2f forked %.2f" % (tot_seq, tot_fork))


if __name__ == "__main__":
    main()
