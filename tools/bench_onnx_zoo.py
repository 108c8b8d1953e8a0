"""Batch-1 inference latency of the twelve model-zoo vision networks at full
size (synthetic graphs with random weights, examples/onnx/<model>.py), eager
dispatch against the recorded forward pass (one hipGraph per inference).

    python tools/bench_onnx_zoo.py [--repeat 30] [--models resnet18,vgg16]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "onnx"))

import zoo  # noqa: E402
from singa_amd import device, tensor  # noqa: E402


def latency(model, x, dev, graph, repeat):
    m = zoo.ZooModel(model)
    m.compile([x], is_train=False, use_graph=graph, sequential=True)
    m.graph_inference = graph
    m.eval()
    for _ in range(4):
        out = m(x)
    y = tensor.to_numpy(out)
    dev.SyncNow()
    t0 = time.perf_counter()
    for _ in range(repeat):
        out = m(x)
    dev.SyncNow()
    return (time.perf_counter() - t0) * 1e3 / repeat, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=30)
    ap.add_argument("--models", default=",".join(zoo.MODELS))
    args = ap.parse_args()
    dev = device.create_cuda_gpu()
    for name in args.models.split(","):
        mod = importlib.import_module(name)
        model = mod.build()
        x_np = mod.preprocess(zoo.synthetic_image(mod.INPUT)).astype(
            np.float32)
        x = tensor.from_numpy(np.ascontiguousarray(x_np), dev)
        eager, y0 = latency(model, x, dev, False, args.repeat)
        graph, y1 = latency(model, x, dev, True, args.repeat)
        diff = float(np.abs(y0 - y1).max() / max(1.0, np.abs(y0).max()))
        print(json.dumps({"model": name, "nodes": len(model.graph.node),
                          "eager_ms": round(eager, 3),
                          "graph_ms": round(graph, 3),
                          "max_rel_diff": diff, "out_scale": float(np.abs(y0).max())}), flush=True)


if __name__ == "__main__":
    main()
