"""Backward chain of ResNet-50 (batch-norm backward -> data gradient per
layer, deepest layer first) with the weight gradients (a) in line, as the
training step issues them, (b) on a second stream that forks after the batch
norm backward of its layer and joins at the end of the chain: do the
compute-bound weight gradients hide under the memory-bound normalisation
kernels?

    python tools/bench_conv_overlap2.py [--batch 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device as sdevice  # noqa: E402
from bench_conv import resnet50_layers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="fp32",
                    choices=["fp32", "fp16", "bf16"])
    args = ap.parse_args()
    dt = {"fp32": torch.float32, "fp16": torch.float16,
          "bf16": torch.bfloat16}[args.dtype]
    sdev = sdevice.create_cuda_gpu()
    dev, m = sdev.torch_device, sdev.math
    N = args.batch
    layers = []
    for (C, K, R, st, pd, H, cnt) in resnet50_layers()[1:]:
        P = (H + 2 * pd - R) // st + 1
        x = torch.randn(N, C, H, H, device=dev).to(dt)
        w = (torch.randn(K, C, R, R, device=dev) * 0.05).to(dt)
        y = torch.randn(N, K, P, P, device=dev).to(dt)   # conv output = BN input
        dz = torch.randn(N, K, P, P, device=dev).to(dt)  # gradient after the BN
        dy = torch.empty_like(dz)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        scale, bias = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        mean, inv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
        ds, db = torch.empty(K, device=dev), torch.empty(K, device=dev)
        for _ in range(cnt):
            layers.append((x, w, y, dz, dy, dx, dw, scale, bias, mean, inv,
                           ds, db, (st, st), (pd, pd)))
    layers.reverse()
    side = torch.cuda.Stream()

    def chain(fork):
        cur = torch.cuda.current_stream()
        for (x, w, y, dz, dy, dx, dw, scale, bias, mean, inv, ds, db, st,
             pd) in layers:
            m.bn_bwd(dz, y, scale, mean, inv, dy, ds, db, 1, bias, None, None)
            if fork:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    m.conv_bwd_weight(dy, x, dw, st, pd, 1)
                m.conv_bwd_data(dy, w, dx, st, pd, 1)
            else:
                m.conv_bwd_data(dy, w, dx, st, pd, 1)
                m.conv_bwd_weight(dy, x, dw, st, pd, 1)
        if fork:
            cur.wait_stream(side)

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5

    t_s = timeit(lambda: chain(False))
    t_f = timeit(lambda: chain(True))
    print(args.dtype, end=" ")
    print("backward chain of %d layers: in line %.3f ms, weight gradients on "
          "a second stream %.3f ms (%.2fx)" % (len(layers), t_s, t_f,
                                              t_s / t_f))


if __name__ == "__main__":
    main()
