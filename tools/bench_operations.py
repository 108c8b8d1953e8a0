"""Operator micro-benchmarks of the reference's test/singa/
test_operation_benchmark.cc:36-146 through the flat native API: cross entropy
on 64 x 10 logits, products 64 x {1595, 1600, 1605} x 2048 and the forward
pass of 1x1 / stride 2 convolutions (batch 64, 28 x 28, 1024 / 256 / 64 -> 64
channels), each in float16 and float32.  The reference times 1000 calls with a
stream synchronisation after every call (launch + completion latency); the
same is printed here, next to the time of back-to-back launches.

    python tools/bench_operations.py [--iters 1000]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device  # noqa: E402
from singa_amd import singa_wrap as singa  # noqa: E402


def measure(dev, fn, iters):
    for _ in range(10):
        fn()
    dev.SyncNow()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
        dev.SyncNow()
    synced = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    dev.SyncNow()
    return synced * 1e6, (time.perf_counter() - t0) / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    args = ap.parse_args()
    dev = device.create_cuda_gpu()
    types = (("float16", singa.kFloat16), ("float32", singa.kFloat32))

    def emit(**row):
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v)
                          for k, v in row.items()}), flush=True)

    t = singa.Tensor([64], dev, singa.kInt)
    t.SetFloatValue(0.0)
    for name, dt in types:
        p = singa.Tensor([64, 10], dev, singa.kFloat32)
        singa.Uniform(0.0, 1.0, p)
        p = p.AsType(dt)
        a, b = measure(dev, lambda: singa.CrossEntropyFwd(p, t), args.iters)
        emit(op="CrossEntropyFwd", shape="64x10", dtype=name, us_synced=a,
             us_back_to_back=b)
    for k in (1595, 1600, 1605):
        for name, dt in types[::-1]:
            x = singa.Tensor([64, k], dev, singa.kFloat32)
            w = singa.Tensor([k, 2048], dev, singa.kFloat32)
            singa.Gaussian(0.0, 1.0, x)
            singa.Gaussian(0.0, 1.0, w)
            x, w = x.AsType(dt), w.AsType(dt)
            a, b = measure(dev, lambda: singa.Mult(x, w), args.iters)
            emit(op="Mult", shape="64x%dx2048" % k, dtype=name, us_synced=a,
                 us_back_to_back=b,
                 tflops_back_to_back=2.0 * 64 * k * 2048 / b / 1e6)
    for cin in (1024, 256, 64):
        for name, dt in types:
            x = singa.Tensor([64, cin, 28, 28], dev, singa.kFloat32)
            w = singa.Tensor([64, cin, 1, 1], dev, singa.kFloat32)
            bias = singa.Tensor([64], dev, singa.kFloat32)
            for tt in (x, w, bias):
                singa.Gaussian(0.0, 1.0, tt)
            x, w, bias = x.AsType(dt), w.AsType(dt), bias.AsType(dt)
            h = singa.CudnnConvHandle(x, [1, 1], [2, 2], [0, 0], cin, 64,
                                      False)
            a, b = measure(dev, lambda: singa.GpuConvForward(x, w, bias, h),
                           args.iters)
            emit(op="Conv 1x1 stride 2", shape="64x%dx28x28 -> 64" % cin,
                 dtype=name, us_synced=a, us_back_to_back=b,
                 tflops_back_to_back=2.0 * 64 * 14 * 14 * cin * 64 / b / 1e6)


if __name__ == "__main__":
    main()
