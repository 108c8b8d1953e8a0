"""Forward-only (serving) latency and throughput of a model-zoo classifier.

    python tools/bench_inference.py --model resnet50 --batch 1 32 --dtype fp16

The model runs in eval mode; with --graph (default) the forward pass is
recorded once and replayed as a hipGraph (Model.graph_inference).  One JSON
line per (batch, mode).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 32])
    ap.add_argument("--dtype", default="fp32",
                    choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--eager", action="store_true",
                    help="also time the unrecorded forward pass")
    args = ap.parse_args()

    import torch
    from singa_amd import device, tensor
    from singa_amd.models import resnet

    dev = device.create_cuda_gpu_on(0)
    dt = {"fp32": tensor.float32, "fp16": tensor.float16,
          "bf16": tensor.bfloat16}[args.dtype]
    for B in args.batch:
        x = np.random.RandomState(0).randn(B, 3, args.size,
                                           args.size).astype(np.float32)
        tx = tensor.Tensor((B, 3, args.size, args.size), dev, tensor.float32)
        tx.copy_from_numpy(x)
        if dt != tensor.float32:
            tx = tx.as_type(dt)
        for graph in ([True, False] if args.eager else [True]):
            model = getattr(resnet, args.model)(num_channels=3,
                                                num_classes=1000)
            model.compile([tx], is_train=False, use_graph=graph,
                          sequential=False)
            model.eval()
            model.graph_inference = graph
            for _ in range(5):
                out = model(tx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model(tx)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            top = int(np.argmax(tensor.to_numpy(
                out.as_type(tensor.float32))[0]))
            print(json.dumps({
                "model": args.model, "batch": B, "dtype": args.dtype,
                "recorded": graph, "ms": round(ms, 3),
                "images_per_s": round(B * 1e3 / ms, 1), "top1_of_first": top,
                "data": "synthetic"}), flush=True)


if __name__ == "__main__":
    main()
