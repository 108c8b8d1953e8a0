"""Direct float32 3x3 kernels against the Winograd routes (F(2x2, 3x3),
F(4x4, 3x3): transforms of csrc/hip/winograd.hip around a batched GEMM) on the
stride-1 3x3 layers of ResNet-50 at batch 32; each variant replayed from a
hipGraph burst.  Prints ms per call for forward and data gradient and the
maximal deviation of each transform from the direct kernel's result."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device  # noqa: E402

LAYERS = [(64, 64, 56, 3), (128, 128, 28, 3), (256, 256, 14, 5),
          (512, 512, 7, 2)]


def timed(fn, dev):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(8):
            fn()
    g.replay()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 32


def main():
    batch = int(os.environ.get("BATCH", "32"))
    dt = {"fp32": torch.float32, "fp16": torch.float16,
          "bf16": torch.bfloat16}[os.environ.get("DTYPE", "fp32")]
    ms = (2, 4) if dt == torch.float32 else (2,)
    modes = 3
    dev = device.create_cuda_gpu_on(0)
    hm = dev.math
    td = dev.torch_device
    total = {"direct": 0.0, "best": 0.0}
    print("layer C,K,HW xcount |   fwd direct   F2     F4  |  bwdD direct   F2     F4  |  bwdW direct   F2     F4  | max rel dev F2 / F4 per pass")
    for C, K, HW, count in LAYERS:
        x = torch.randn(batch, C, HW, HW, device=td).to(dt)
        w = (torch.randn(K, C, 3, 3, device=td) * 0.05).to(dt)
        dy = torch.randn(batch, K, HW, HW, device=td).to(dt)
        y = torch.empty(batch, K, HW, HW, device=td, dtype=dt)
        dx = torch.empty(batch, C, HW, HW, device=td, dtype=dt)
        row, devs = [], []
        dw = torch.empty(K, C, 3, 3, device=td, dtype=dt)
        for mode, act, out in ((0, x, y), (1, dy, dx), (2, dy, dw))[:modes]:
            if mode == 0:
                d = lambda: hm._conv_fwd_main(x, w, None, y, (1, 1), (1, 1), 1, None, None)
            elif mode == 1:
                d = lambda: hm._conv_bwd_data_main(dy, w, dx, (1, 1), (1, 1), 1, None, None)
            else:
                d = lambda: (hm._wgrad16(dy, x, dw, (1, 1), (1, 1), 1) or
                             hm._conv_bwd_weight_direct(dy, x, dw, (1, 1), (1, 1), 1, False))
            t0 = timed(d, dev)
            ref = out.clone()
            ts = [t0]
            for m in ms:
                f = lambda: hm._winograd_any(m, mode, act, x if mode == 2 else w, None, None, out)
                if not f():
                    ts.append(float("nan"))
                    devs.append(float("nan"))
                    continue
                ts.append(timed(f, dev))
                devs.append(float((out.float() - ref.float()).abs().max() / ref.float().abs().max()))
            row.append(ts)
            total["direct"] += count * t0
            total["best"] += count * min(ts)
        print("%d,%d,%d x%d | %s | %s" % (
            C, K, HW, count, " | ".join(" ".join("%7.3f" % t for t in r) for r in row),
            " ".join("%.1e" % v for v in devs)), flush=True)
    print("sum over the step's stride-1 3x3 layers, all three passes: direct %.3f ms, best route per layer %.3f ms"
          % (total["direct"], total["best"]))


if __name__ == "__main__":
    main()
