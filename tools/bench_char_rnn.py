"""Training throughput of the character-level LSTM example
(examples/rnn/char_rnn.py: batch 32, 64 steps, hidden 128 by default) on
synthetic text, in graph mode.

    python tools/bench_char_rnn.py [--steps 30] [--variant all]

Variants: ``sequence`` (the whole sequence is one operator; on the GPU the
recurrence is a single persistent launch), ``per-step-kernels`` (one operator,
but one GEMM + one cell kernel per time step) and ``unrolled`` (a chain of
per-step operators, the structure of the reference's python LSTM).  Each
variant runs in a child process because the switches are read from the
environment.  Prints one JSON line per variant: tokens/s and ms per
iteration.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "rnn"))

VARIANTS = {
    "sequence": {},
    "per-step-kernels": {"SINGA_LSTM_SEQ": "0"},
    "unrolled": {"SINGA_LSTM_UNROLLED": "1"},
}


def run(args):
    import char_rnn
    from singa_amd import device, tensor
    dev = device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()
    rs = np.random.RandomState(0)
    vocab, B, T = args.vocab, args.batch, args.seq
    model = char_rnn.CharRNN(vocab, args.hidden, 0.01)
    model.graph(not args.eager, False)
    model.train()
    batch = rs.randint(0, vocab, (B, T + 1)).astype(np.int32)
    inputs, labels = char_rnn.convert(batch, B, T, vocab, dev)
    losses = []
    for _ in range(args.warmup):
        out, loss = model(inputs, labels)
        model.reset_states(dev)
    losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    dev.Sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, loss = model(inputs, labels)
        model.reset_states(dev)
    dev.Sync()
    dt = (time.perf_counter() - t0) / args.steps
    losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    print(json.dumps({
        "variant": args.variant, "tokens_per_s": round(B * T / dt, 1),
        "ms_per_iteration": round(dt * 1e3, 3), "batch": B, "seq": T,
        "hidden": args.hidden, "vocab": vocab, "graph": not args.eager,
        "loss_first_last": losses, "data": "synthetic"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="all",
                    choices=["all"] + list(VARIANTS))
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--vocab", type=int, default=101)
    ap.add_argument("--eager", action="store_true")
    args = ap.parse_args()
    if args.variant != "all":
        return run(args)
    for name, env in VARIANTS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--variant", name]
        for k in ("steps", "warmup", "batch", "seq", "hidden", "vocab"):
            cmd += ["--" + k, str(getattr(args, k))]
        if args.eager:
            cmd.append("--eager")
        code = subprocess.call(cmd, env=dict(os.environ, **env))
        if code != 0:
            sys.exit(code)


if __name__ == "__main__":
    main()
