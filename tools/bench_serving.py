"""Throughput and latency of the prediction service's batching front end
(examples/singa_easy/serve.py) without the HTTP layer: C client threads each
send single images back to back; the predictor stacks what arrives within
``--max-wait-ms`` and replays the recorded forward pass.

    python tools/bench_serving.py --model resnet50 --dtype fp16 --clients 1 8 32 64

One JSON line per client count (synthetic images, random weights).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "examples", "singa_easy")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--dtype", default="fp16", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--clients", type=int, nargs="+", default=[1, 8, 32, 64])
    ap.add_argument("--max-batch", type=int, default=32)
    ap.add_argument("--max-wait-ms", type=float, default=1.0)
    ap.add_argument("--seconds", type=float, default=4.0)
    args = ap.parse_args()

    import serve
    from singa_amd import device, models, tensor
    dev = device.create_cuda_gpu()
    dt = {"fp32": tensor.float32, "fp16": tensor.float16,
          "bf16": tensor.bfloat16}[args.dtype]

    def build():
        return getattr(models.resnet, args.model)(num_channels=3,
                                                  num_classes=1000)

    predictor = serve.BatchingPredictor(build, (3, 224, 224), args.max_batch,
                                        args.max_wait_ms / 1e3, dev, dt, True)
    assert predictor.ready.wait(300) and predictor.failed is None, \
        predictor.failed
    image = np.random.RandomState(0).randn(3, 224, 224).astype(np.float32)
    for _ in range(5):
        predictor.predict(image)
    for clients in args.clients:
        stop = time.monotonic() + args.seconds
        latencies = [[] for _ in range(clients)]

        def client(k):
            while time.monotonic() < stop:
                t0 = time.perf_counter()
                predictor.predict(image)
                latencies[k].append(time.perf_counter() - t0)

        before = dict(predictor.stats)
        threads = [threading.Thread(target=client, args=(k,))
                   for k in range(clients)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        wall = time.perf_counter() - t0
        lat = np.sort(np.concatenate([np.asarray(v) for v in latencies]))
        batches = predictor.stats["batches"] - before["batches"]
        print(json.dumps({
            "model": args.model, "dtype": args.dtype, "clients": clients,
            "max_batch": args.max_batch, "max_wait_ms": args.max_wait_ms,
            "requests_per_s": round(len(lat) / wall, 1),
            "mean_batch": round(len(lat) / max(batches, 1), 2),
            "latency_ms_p50": round(float(lat[len(lat) // 2]) * 1e3, 2),
            "latency_ms_p99": round(float(lat[int(len(lat) * 0.99)]) * 1e3, 2),
            "data": "synthetic"}), flush=True)
    predictor.close()


if __name__ == "__main__":
    main()
