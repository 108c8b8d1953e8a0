"""BERT through the ONNX importer on one MI355X: inference and fine-tuning
throughput (sequences / second), with and without the importer's pattern
fusion (fused attention / layer-norm / GELU kernels).

Workload of the reference: examples/onnx/bert/bert-squad.py:102-165 (BERT-base,
batch 1, sequence 256, eager) and a fine-tuning step in the style of
examples/onnx/training/train.py:105-142 (Linear head, SGD, graph mode).  The
graph is generated (random weights of the BERT-base architecture), so no model
file is needed.

    python tools/bench_bert.py [--layers 12] [--hidden 768] [--heads 12]
        [--seq 256] [--train-batches 8,32] [--steps 10] [--no-unfused]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "onnx"))

import torch  # noqa: E402

import text_models as tm  # noqa: E402
from singa_amd import device, opt, sonnx, tensor  # noqa: E402


def inputs(dev, B, S, vocab, rs):
    ids = rs.randint(0, vocab, (B, S))
    mask = np.ones((B, S), np.int64)
    mask[:, S - S // 8:] = 0            # padded tail
    seg = (np.arange(S)[None, :] >= S // 2).astype(np.int64).repeat(B, 0)
    return [tensor.from_numpy(a.astype(np.int32), dev)
            for a in (ids, mask, seg)]


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--ffn", type=int, default=3072)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--vocab", type=int, default=30522)
    ap.add_argument("--train-batches", default="8,32")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-unfused", action="store_true")
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--dtype", choices=["fp32", "fp16", "bf16"],
                    default="fp32",
                    help="16-bit: the float32 graph is imported with its "
                         "weights and activations in that type (float32 "
                         "master weights in the optimizer)")
    ap.add_argument("--check", action="store_true",
                    help="compare fused and unfused outputs with float64")
    args = ap.parse_args()
    dev = device.create_cuda_gpu()
    rs = np.random.RandomState(0)
    cfg = dict(seq=args.seq, hidden=args.hidden, heads=args.heads,
               layers=args.layers, ffn=args.ffn, vocab=args.vocab)
    out = {"config": cfg, "dtype": args.dtype, "data": "synthetic",
           "results": []}
    dt = {"fp32": None, "fp16": tensor.float16,
          "bf16": tensor.bfloat16}[args.dtype]
    modes = [True] if args.no_unfused else [True, False]

    # ---- inference, batch 1, eager (the reference's script) and graph mode
    model1, W = tm.bert_encoder(batch=1, **cfg)
    xs = inputs(dev, 1, args.seq, args.vocab, rs)
    outs = {}
    for fuse in ([] if args.no_inference else modes):
        sonnx.SingaBackend.fuse_patterns = fuse
        for graph in (False, True):
            m = tm.TextModel(model1, dtype=dt)
            m.compile(xs, is_train=False, use_graph=graph, sequential=True)
            m.graph_inference = graph   # recorded forward pass -> hipGraph
            m.eval()
            t = timed(lambda: m(*xs), args.steps, 3)
            outs[fuse] = tensor.to_numpy(
                m(*xs)[0].as_type(tensor.float32))
            out["results"].append({"task": "inference", "batch": 1,
                                   "fused": fuse, "graph": graph,
                                   "ms": round(t * 1e3, 3),
                                   "seq_per_s": round(1.0 / t, 2)})
            print(json.dumps(out["results"][-1]), flush=True)
            dev.ResetGraph()
    if args.check:
        want = tm.bert_encoder_numpy(
            W, *[tensor.to_numpy(x).astype(np.int64) for x in xs],
            args.heads, args.layers)
        for fuse, got in outs.items():
            print("fused=%s max |out - float64 reference| = %.3g" % (
                fuse, float(np.abs(got - want).max())), flush=True)
        if len(outs) == 2:
            print("fused vs unfused max abs diff %.3g" % float(
                np.abs(outs[True] - outs[False]).max()), flush=True)
    del model1, W

    # ---- fine-tuning step (Linear head on the first token, SGD, graph mode)
    for B in [int(b) for b in args.train_batches.split(",") if b]:
        modelB, _ = tm.bert_encoder(batch=B, squad_head=False, **cfg)
        xs = inputs(dev, B, args.seq, args.vocab, rs)
        ty = tensor.from_numpy(rs.randint(0, 2, B).astype(np.int32), dev)
        for fuse in modes:
            sonnx.SingaBackend.fuse_patterns = fuse
            m = tm.BertClassifier(modelB, num_classes=2, dtype=dt)
            m.set_optimizer(opt.SGD(lr=1e-3, momentum=0.9))
            m.compile(xs, is_train=True, use_graph=True, sequential=False)
            m.train()
            def step():
                return m(*xs, ty)[1]

            def loss_value():
                return float(tensor.to_numpy(
                    step().as_type(tensor.float32)).reshape(-1)[0])

            first = loss_value()
            t = timed(step, args.steps, 4)
            lv = [first, loss_value()]
            out["results"].append({"task": "fine-tune", "batch": B,
                                   "fused": fuse, "graph": True,
                                   "ms": round(t * 1e3, 3),
                                   "seq_per_s": round(B / t, 2),
                                   "loss_first_last": lv})
            print(json.dumps(out["results"][-1]), flush=True)
            dev.ResetGraph()
            del m
        del modelB
    sonnx.SingaBackend.fuse_patterns = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
