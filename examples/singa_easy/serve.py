"""A prediction service around a recorded forward pass: requests that arrive
within a few milliseconds of each other are stacked into one batch, the batch
runs as one replayed hipGraph on the MI355X (``Model.graph_inference``), and
every caller gets its own rows back.  The role of the application that the
reference's examples/singa_easy/scripts/start.sh brings up.

    python serve.py --model resnet18 --classes 10 --port 8000
    python serve.py --onnx model.onnx --shape 3 224 224 --dtype fp16
    curl -X POST localhost:8000/predict -H 'content-type: application/json' \\
         -d '{"inputs": [[[[0.1, ...]]]]}'

Endpoints: ``POST /predict`` (JSON ``{"inputs": nested list}`` of one sample or
a list of samples, or an ``.npy`` body with content type
``application/x-npy``), ``POST /explain`` (LIME weights of one image's grid
segments, see lime.py), ``GET /healthz``, ``GET /metrics`` (Prometheus text).
"""
import argparse
import io
import os
import queue
import sys
import threading
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, ".."),
                os.path.join(_HERE, "..", "..")]

from singa_amd import device, tensor  # noqa: E402


class BatchingPredictor(object):
    """Owns the model and the only thread that touches the device.

    ``submit(samples)`` (an array of shape (n,) + sample_shape) returns an
    object with ``result(timeout)``.  The worker takes what is queued, up to
    ``max_batch`` samples or until ``max_wait`` seconds have passed since the
    first one, copies it into the input tensor of the smallest recorded batch
    size that holds it and replays that pass."""

    def __init__(self, build_model, sample_shape, max_batch=8,
                 max_wait=0.002, dev=None, dtype=tensor.float32, graph=True):
        self.sample_shape = tuple(sample_shape)
        self.max_batch, self.max_wait = int(max_batch), float(max_wait)
        self.dev = dev or device.get_default_device()
        self.dtype = dtype
        self.graph = graph
        self.build_model = build_model
        self.requests = queue.Queue()
        self.stats = {"requests": 0, "samples": 0, "batches": 0,
                      "device_seconds": 0.0, "errors": 0}
        self.ready = threading.Event()
        self.failed = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    # ---- client side -------------------------------------------------
    class _Pending(object):
        def __init__(self, samples):
            self.samples = samples
            self.done = threading.Event()
            self.value, self.error = None, None

        def result(self, timeout=None):
            if not self.done.wait(timeout):
                raise TimeoutError("prediction timed out")
            if self.error is not None:
                raise self.error
            return self.value

    def submit(self, samples):
        samples = np.asarray(samples, dtype=np.float32)
        if samples.shape == self.sample_shape:
            samples = samples[None]
        if samples.shape[1:] != self.sample_shape or samples.shape[0] == 0:
            raise ValueError("expected samples of shape %s, got %s" % (
                self.sample_shape, samples.shape))
        if self.failed is not None:
            raise RuntimeError("the model could not be set up: %s" %
                               (self.failed,))
        pending = self._Pending(samples)
        self.requests.put(pending)
        return pending

    def predict(self, samples, timeout=30.0):
        return self.submit(samples).result(timeout)

    def close(self):
        self.requests.put(None)
        self.thread.join(10)

    # ---- device side -------------------------------------------------
    def _compile(self):
        """One input tensor and one recorded pass per batch size 1, 2, 4, ...
        max_batch: a lone request neither uploads nor computes a full batch
        (``Model`` keeps a recorded pass per input shape)."""
        self.buckets = sorted({min(1 << k, self.max_batch)
                               for k in range(self.max_batch.bit_length() + 1)})
        self.slots = {}
        for rows in self.buckets:
            host = self._host_buffer((rows,) + self.sample_shape)
            staging = tensor.Tensor(host.shape, self.dev, tensor.float32)
            staging.copy_from_numpy(host)
            self.slots[rows] = (host, staging)
        self.model = self.build_model()
        self.model.compile([self._input(self.buckets[-1])], is_train=False,
                           use_graph=self.graph, sequential=False)
        self.model.eval()
        self.model.graph_inference = self.graph
        for rows in self.buckets:
            self._forward(rows)             # records the pass of this size

    def _host_buffer(self, shape):
        """Zeroed float32 staging array; page-locked when the model runs on a
        GPU, so that the upload is one DMA transfer from this buffer."""
        import torch
        if self.graph and torch.cuda.is_available():
            pinned = torch.zeros(shape, dtype=torch.float32).pin_memory()
            self._pinned = getattr(self, "_pinned", []) + [pinned]
            return pinned.numpy()
        return np.zeros(shape, np.float32)

    def _input(self, rows):
        staging = self.slots[rows][1]
        if self.dtype == tensor.float32:
            return staging
        return staging.as_type(self.dtype)

    def _forward(self, rows):
        out = self.model(self._input(rows))
        return tensor.to_numpy(out.as_type(tensor.float32)
                               if self.dtype != tensor.float32 else out)

    def _run(self):
        try:
            self._compile()
        except Exception as e:              # surfaced by /healthz and submit
            self.failed = e
            self.ready.set()
            self._drain(e)
            return
        self.ready.set()
        carry, stop = None, False
        while not stop:
            first, carry = (carry, None) if carry is not None \
                else (self.requests.get(), None)
            if first is None:
                return
            group, rows = [first], first.samples.shape[0]
            deadline = time.monotonic() + self.max_wait
            while rows < self.max_batch:
                left = deadline - time.monotonic()
                try:
                    nxt = self.requests.get(timeout=left) if left > 0 \
                        else self.requests.get_nowait()
                except queue.Empty:
                    break
                if nxt is None:
                    stop = True
                    break
                if rows + nxt.samples.shape[0] > self.max_batch:
                    carry = nxt             # opens the next batch
                    break
                group.append(nxt)
                rows += nxt.samples.shape[0]
            self._serve(group)

    def _drain(self, error):
        while True:
            try:
                p = self.requests.get_nowait()
            except queue.Empty:
                return
            if p is not None:
                p.error = error
                p.done.set()

    def _serve(self, group):
        """Run the samples of ``group`` in passes of at most ``max_batch``
        rows; each sample is copied once, from the request straight into the
        staging buffer of the pass it belongs to."""
        try:
            total = sum(p.samples.shape[0] for p in group)
            outs = []
            flat = [(p, i) for p in group for i in range(p.samples.shape[0])]
            for lo in range(0, total, self.max_batch):
                part = flat[lo:lo + self.max_batch]
                rows = next(b for b in self.buckets if b >= len(part))
                host, staging = self.slots[rows]
                # rows beyond the request keep what an earlier batch left:
                # every row of the output depends on its own sample only
                at = 0
                while at < len(part):       # runs of rows of one request
                    p, i = part[at]
                    run = 1
                    while at + run < len(part) and part[at + run][0] is p:
                        run += 1
                    host[at:at + run] = p.samples[i:i + run]
                    at += run
                t0 = time.perf_counter()
                staging.copy_from_numpy(host)
                y = self._forward(rows)
                self.stats["device_seconds"] += time.perf_counter() - t0
                self.stats["batches"] += 1
                outs.append(y[:len(part)])
            y = outs[0] if len(outs) == 1 else np.concatenate(outs, 0)
            at = 0
            for p in group:
                n = p.samples.shape[0]
                p.value = np.array(y[at:at + n])
                at += n
                self.stats["requests"] += 1
                self.stats["samples"] += n
                p.done.set()
        except Exception as e:
            self.stats["errors"] += 1
            for p in group:
                p.error = e
                p.done.set()


def metrics_text(stats):
    names = {"requests": "prediction requests answered",
             "samples": "samples predicted",
             "batches": "forward passes replayed",
             "device_seconds": "seconds spent in upload + forward + download",
             "errors": "batches that failed"}
    lines = []
    for key, doc in names.items():
        lines.append("# HELP singa_serve_%s %s" % (key, doc))
        lines.append("# TYPE singa_serve_%s counter" % key)
        lines.append("singa_serve_%s %s" % (key, stats[key]))
    return "\n".join(lines) + "\n"


def create_app(predictor, explain_cells=4, explain_samples=128):
    """The ASGI application (starlette)."""
    from starlette.applications import Starlette
    from starlette.concurrency import run_in_threadpool
    from starlette.responses import JSONResponse, PlainTextResponse
    from starlette.routing import Route

    async def _samples(request):
        body = await request.body()
        kind = request.headers.get("content-type", "")
        if kind.startswith("application/x-npy"):
            return np.load(io.BytesIO(body), allow_pickle=False)
        import json
        return np.asarray(json.loads(body)["inputs"], dtype=np.float32)

    async def predict(request):
        try:
            samples = await _samples(request)
            pending = predictor.submit(samples)
        except (ValueError, KeyError, TypeError) as e:
            return JSONResponse({"error": str(e)}, status_code=400)
        try:
            y = await run_in_threadpool(pending.result, 60.0)
        except Exception as e:
            return JSONResponse({"error": str(e)}, status_code=500)
        return JSONResponse({"outputs": y.tolist(),
                             "classes": np.argmax(y, -1).tolist()})

    async def explain(request):
        import lime
        try:
            image = await _samples(request)
            if image.shape != predictor.sample_shape:
                raise ValueError("one sample of shape %s" % (
                    predictor.sample_shape,))
        except (ValueError, KeyError, TypeError) as e:
            return JSONResponse({"error": str(e)}, status_code=400)

        def work():
            def scores(batch):
                y = predictor.predict(batch)
                e = np.exp(y - y.max(1, keepdims=True))
                return e / e.sum(1, keepdims=True)
            label = int(np.argmax(scores(image[None])[0]))
            w = lime.explain(scores, image, label, cells=explain_cells,
                             samples=explain_samples,
                             rng=np.random.RandomState(0))
            return label, w
        label, w = await run_in_threadpool(work)
        return JSONResponse({"class": label, "cells": explain_cells,
                             "segment_weights": np.asarray(w).tolist()})

    async def healthz(request):
        if predictor.failed is not None:
            return JSONResponse({"status": "failed",
                                 "error": str(predictor.failed)},
                                status_code=500)
        ok = predictor.ready.is_set()
        return JSONResponse({"status": "ok" if ok else "loading"},
                            status_code=200 if ok else 503)

    async def metrics(request):
        return PlainTextResponse(metrics_text(predictor.stats))

    return Starlette(routes=[
        Route("/predict", predict, methods=["POST"]),
        Route("/explain", explain, methods=["POST"]),
        Route("/healthz", healthz), Route("/metrics", metrics)])


def model_builder(args):
    """() -> model, sample shape."""
    if args.onnx:
        from singa_amd import sonnx
        shape = tuple(args.shape)

        def build():
            proto = sonnx.onnx.load(args.onnx)
            return sonnx.SONNXModel(proto)
        return build, shape
    from singa_amd import models
    shape = (3, args.size, args.size)

    def build():
        if args.model.startswith("resnet"):
            return getattr(models.resnet, args.model)(
                num_channels=3, num_classes=args.classes)
        return getattr(models, args.model).create_model(
            num_channels=3, num_classes=args.classes)
    return build, shape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet18",
                    help="a model-zoo name: resnet18..152, cnn, alexnet, "
                         "xceptionnet")
    ap.add_argument("--onnx", default=None, help="serve this ONNX file")
    ap.add_argument("--shape", type=int, nargs="+", default=[3, 224, 224],
                    help="sample shape of the ONNX model")
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--max-wait-ms", type=float, default=2.0)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    import uvicorn
    dev = device.get_default_device() if args.cpu else \
        device.create_cuda_gpu()
    build, shape = model_builder(args)
    dt = {"fp32": tensor.float32, "fp16": tensor.float16,
          "bf16": tensor.bfloat16}[args.dtype]
    predictor = BatchingPredictor(build, shape, args.max_batch,
                                  args.max_wait_ms / 1e3, dev, dt,
                                  graph=not args.cpu)
    uvicorn.run(create_app(predictor), host=args.host, port=args.port,
                log_level="warning")


if __name__ == "__main__":
    main()
