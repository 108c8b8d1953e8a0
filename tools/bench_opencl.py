"""Throughput of the OpenCL device of the C++ library next to the HIP device,
through the same C interface (singa_amd.native): a square float32 product and
an element-wise add.  The OpenCL device is the compatibility path; the numbers
say how far it is from the HIP kernels.

    python tools/bench_opencl.py [--n 2048]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))

from singa_amd import native  # noqa: E402


def timed(dev, fn, reps):
    fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    dev.sync()
    del out
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    n = args.n
    rs = np.random.RandomState(0)
    a = rs.randn(n, n).astype(np.float32)
    b = rs.randn(n, n).astype(np.float32)
    devices = []
    if native.num_gpus() > 0:
        devices.append(("hip", native.Device.gpu(0)))
    if native.num_opencl_devices() > 0:
        devices.append(("opencl", native.Device.opencl(0)))
    ref = None
    for name, dev in devices:
        ta, tb = dev.tensor(a), dev.tensor(b)
        got = (ta @ tb).numpy()
        if ref is None:
            ref = got
        err = float(np.abs(got - ref).max())
        t_mm = timed(dev, lambda: ta @ tb, args.reps)
        t_add = timed(dev, lambda: ta + tb, args.reps)
        print(json.dumps({
            "device": name, "n": n,
            "matmul_ms": round(t_mm * 1e3, 3),
            "matmul_tflops": round(2 * n ** 3 / t_mm / 1e12, 2),
            "add_ms": round(t_add * 1e3, 3),
            "add_gb_per_s": round(3 * 4 * n * n / t_add / 1e9, 1),
            "max_abs_diff_vs_first": err}), flush=True)


if __name__ == "__main__":
    main()
