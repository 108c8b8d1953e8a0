"""Per-layer timing of the HIP convolution / GEMM kernels on ResNet-50 shapes.

Usage: python tools/bench_conv.py [--batch 32] [--dtype float32] [--check]

Prints one line per (layer, pass) with the time and the achieved TFLOP/s and
a FLOP-weighted total, which is the convolution part of one ResNet-50
training step.  ``--check`` also compares each result with torch's own
convolution (fp32) and prints the max abs error.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device as sdevice  # noqa: E402

# (C, K, R, stride, pad, H_in, count) over the ResNet-50 v1 graph of
# singa_amd/models/resnet.py (stride on the 3x3 convolution of a stage's first
# block)
LAYERS = []


def _stage(cin, mid, cout, h, stride, blocks):
    out = []
    ho = h // stride
    # first block
    out.append((cin, mid, 1, 1, 0, h, 1))
    out.append((mid, mid, 3, stride, 1, h, 1))
    out.append((mid, cout, 1, 1, 0, ho, 1))
    out.append((cin, cout, 1, stride, 0, h, 1))  # downsample
    out.append((cout, mid, 1, 1, 0, ho, blocks - 1))
    out.append((mid, mid, 3, 1, 1, ho, blocks - 1))
    out.append((mid, cout, 1, 1, 0, ho, blocks - 1))
    return out


def resnet50_layers():
    ls = [(3, 64, 7, 2, 3, 224, 1)]
    ls += _stage(64, 64, 256, 56, 1, 3)
    ls += _stage(256, 128, 512, 56, 2, 4)
    ls += _stage(512, 256, 1024, 28, 2, 6)
    ls += _stage(1024, 512, 2048, 14, 2, 3)
    return ls


USE_GRAPH = True


def timeit(fn, iters):
    """ms per call.  The calls are captured into one hipGraph and replayed,
    so the host's launch cost (tens of microseconds per call from Python)
    does not hide kernels shorter than that."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    if not USE_GRAPH:
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters)


PROF_NAMES = ("prologue", "tile0", "plan+issue", "mfma", "wait+store",
              "barrier", "epilogue")


def prof_shares(m, fn):
    """One call with the diagnostic phase counters on (needs a build with
    -DSA_CONV_PROF): share of wave 0's cycles per phase, mean cycles per
    workgroup, and the span between the first workgroup's start and the
    last one's, in cycles."""
    m.conv_prof = torch.zeros(8 + 8 * 16384, dtype=torch.int64, device="cuda")
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    n = int(m.conv_prof[0].item())
    if n == 0:
        m.conv_prof = None
        return "(kernel without phase counters)"
    rec = m.conv_prof[8:8 + 8 * n].reshape(n, 8).cpu().double()
    m.conv_prof = None
    c = rec[:, :7].sum(0).tolist()
    tot = float(sum(c)) or 1.0
    starts = rec[:, 7]
    return " ".join("%s %.0f%%" % (nm, 100.0 * v / tot)
                    for nm, v in zip(PROF_NAMES, c)) + \
        " | cyc/wg %.0f wgs %d start-span %.0f call %.3f ms" % (
            tot / max(n, 1), n, float(starts.max() - starts.min()),
            e0.elapsed_time(e1))



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--prof", action="store_true",
                    help="print phase shares (diagnostic build only)")
    ap.add_argument("--pair", action="store_true",
                    help="data and weight gradient of a layer one after the "
                         "other vs. on two streams at once")
    ap.add_argument("--layers", default="",
                    help="comma separated layer indices (default: all)")
    args = ap.parse_args()
    global USE_GRAPH
    USE_GRAPH = not args.no_graph
    only = set(int(v) for v in args.layers.split(",")) if args.layers else None
    sdev = sdevice.create_cuda_gpu()
    dev = sdev.torch_device
    dt = getattr(torch, args.dtype)
    m = sdev.math
    N = args.batch
    tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
    flops_tot = 0.0
    print("%-34s %10s %8s | %10s %8s | %10s %8s" %
          ("layer C,K,R,s,p,H xcount", "fwd ms", "TF/s", "bwdD ms", "TF/s",
           "bwdW ms", "TF/s"))
    for li, (C, K, R, st, pd, H, cnt) in enumerate(resnet50_layers()):
        if only is not None and li not in only:
            continue
        P = (H + 2 * pd - R) // st + 1
        x = torch.randn(N, C, H, H, device=dev, dtype=dt)
        w = torch.randn(K, C, R, R, device=dev, dtype=dt) * 0.05
        y = torch.empty(N, K, P, P, device=dev, dtype=dt)
        dy = torch.randn(N, K, P, P, device=dev, dtype=dt)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        fl = 2.0 * N * K * P * P * C * R * R
        t_f = timeit(lambda: m.conv_fwd(x, w, None, y, (st, st), (pd, pd), 1),
                     args.iters)
        if os.environ.get("BENCH_CONV_STATS"):
            # forward with batch-norm statistics taken in the same launch
            sums = torch.zeros((2 * m.STAT_COPIES + 1) * K, device=dev)
            shift = torch.zeros(K, device=dev)
            t_d = timeit(lambda: m.conv_fwd(x, w, None, y, (st, st), (pd, pd),
                                            1, sums, shift), args.iters)
            print("%-34s %10.3f plain  %10.3f with statistics" %
                  ("%d,%d,%d,%d,%d,%d x%d" % (C, K, R, st, pd, H, cnt), t_f,
                   t_d), flush=True)
            continue
        if args.pair:
            if li == 0:
                continue
            side = torch.cuda.Stream()

            def serial():
                m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1)
                m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1)

            def forked():
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1)
                m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1)
                cur.wait_stream(side)
            t_s = timeit(serial, args.iters)
            t_p = timeit(forked, args.iters)
            print("%-34s serial %8.3f  two streams %8.3f  (%.2fx)" %
                  ("%d,%d,%d,%d,%d,%d x%d" % (C, K, R, st, pd, H, cnt), t_s,
                   t_p, t_s / t_p), flush=True)
            tot["bwd_data"] += t_s * cnt
            tot["bwd_weight"] += t_p * cnt
            continue
        t_d = 0.0
        if li > 0:
            t_d = timeit(
                lambda: m.conv_bwd_data(dy, w, dx, (st, st), (pd, pd), 1),
                args.iters)
        t_w = timeit(
            lambda: m.conv_bwd_weight(dy, x, dw, (st, st), (pd, pd), 1),
            args.iters)
        err = ""
        if args.check:
            xr = x.float().requires_grad_(True)
            wr = w.float().requires_grad_(True)
            yr = torch.nn.functional.conv2d(xr, wr, None, st, pd)
            yr.backward(dy.float())
            err = "  err f/d/w %.2e %.2e %.2e" % (
                (y.float() - yr).abs().max().item(),
                (dx.float() - xr.grad).abs().max().item() if li > 0 else 0.0,
                (dw.float() - wr.grad).abs().max().item())
        tf = lambda t: fl / t / 1e9 if t > 0 else 0.0
        print("%-34s %10.3f %8.1f | %10.3f %8.1f | %10.3f %8.1f%s" %
              ("%d,%d,%d,%d,%d,%d x%d" % (C, K, R, st, pd, H, cnt), t_f,
               tf(t_f), t_d, tf(t_d), t_w, tf(t_w), err), flush=True)
        if args.prof:
            print("   fwd :", prof_shares(m, lambda: m.conv_fwd(
                x, w, None, y, (st, st), (pd, pd), 1)))
            if li > 0:
                print("   bwdD:", prof_shares(m, lambda: m.conv_bwd_data(
                    dy, w, dx, (st, st), (pd, pd), 1)))
            print("   bwdW:", prof_shares(m, lambda: m.conv_bwd_weight(
                dy, x, dw, (st, st), (pd, pd), 1)))
        tot["fwd"] += t_f * cnt
        tot["bwd_data"] += t_d * cnt
        tot["bwd_weight"] += t_w * cnt
        flops_tot += fl * cnt
    if args.pair:
        print("total ms/step: serial %.2f  two streams %.2f" %
              (tot["bwd_data"], tot["bwd_weight"]))
        return
    print("total ms/step: fwd %.2f  bwd_data %.2f  bwd_weight %.2f  sum %.2f; "
          "fwd FLOPs %.1f G" % (tot["fwd"], tot["bwd_data"],
                                tot["bwd_weight"], sum(tot.values()),
                                flops_tot / 1e9))
    if args.gemm:
        for (M, Nn, Kk) in [(4096, 4096, 4096), (8192, 8192, 1024),
                            (32, 1000, 2048), (1024, 1024, 1024)]:
            a = torch.randn(M, Kk, device=dev, dtype=dt)
            b = torch.randn(Kk, Nn, device=dev, dtype=dt)
            c = torch.empty(M, Nn, device=dev, dtype=dt)
            t = timeit(lambda: m.gemm(1.0, a, b, 0.0, c), args.iters)
            print("gemm %dx%dx%d: %.3f ms %.1f TF/s" %
                  (M, Nn, Kk, t, 2.0 * M * Nn * Kk / t / 1e9))
    for k, v in sorted(type(m)._conv_plans.items(), key=str):
        print("plan", k[0], k[1], k[2][2:], "->", v)



if __name__ == "__main__":
    main()
