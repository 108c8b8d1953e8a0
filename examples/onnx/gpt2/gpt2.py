"""GPT-2 text generation through the ONNX importer.

    python gpt2.py --model gpt2-lm-head-10.onnx --vocab /data/gpt2 \
        --text "Here is some text to encode : Hello World" --length 20
    python gpt2.py                          # synthetic decoder, byte vocabulary

The graph is re-run on the growing sequence (no key / value cache), as the
reference's script does; ``--top-k`` samples instead of taking the arg-max.
Without ``--model`` a GPT-style decoder is built as an ONNX graph with random
weights (text_models.tiny_gpt) over a byte vocabulary.

Workload: reference examples/onnx/gpt2/gpt2.py:42-110.
"""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))
import bpe  # noqa: E402
import text_models  # noqa: E402
from singa_amd import onnx_proto as onnx  # noqa: E402


def next_token(logits, top_k, rng):
    """Arg-max, or a draw from the soft-max over the ``top_k`` best."""
    if top_k <= 1:
        return int(np.argmax(logits))
    best = np.argsort(-logits)[:top_k]
    p = np.exp(logits[best] - logits[best].max())
    return int(rng.choice(best, p=p / p.sum()))


def generate(step, ids, length, top_k=1, seed=0):
    """``step(ids) -> logits of the last position``; returns the new ids."""
    rng = np.random.RandomState(seed)
    ids, new = list(ids), []
    for i in range(length):
        tok = next_token(step(ids), top_k, rng)
        logging.info("token %d: id %d", i, tok)
        ids.append(tok)
        new.append(tok)
    return new


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", help="gpt2-lm-head .onnx file")
    ap.add_argument("--vocab", help="directory with vocab.json / merges.txt")
    ap.add_argument("--text", default="Here is some text to encode : "
                                      "Hello World")
    ap.add_argument("--length", type=int, default=20)
    ap.add_argument("--top-k", type=int, default=1)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    dev = text_models.create_device()
    if args.model:
        tok = bpe.BPETokenizer(args.vocab)
        model = onnx.load(args.model)

        def step(ids):      # zoo layout: (batch, 1, sequence) -> logits
            x = np.asarray(ids, np.int64).reshape(1, 1, -1)
            logits = text_models.run(model, [x], dev)[0]
            return logits.reshape(-1, logits.shape[-1])[-1]
    else:
        tok = bpe.ByteTokenizer()
        seq = len(tok.encode(args.text)) + args.length
        logging.info("no model file: decoder with random weights, byte "
                     "vocabulary, context %d", seq)
        model, _ = text_models.tiny_gpt(vocab=256, seq=seq, dim=128, heads=2,
                                        layers=4)

        def step(ids):      # fixed context: pad, read the last real position
            x = np.zeros((1, seq), np.int64)
            x[0, :len(ids)] = ids
            return text_models.run(model, [x], dev)[0][0, len(ids) - 1]
    ids = tok.encode(args.text)
    new = generate(step, ids, args.length, args.top_k)
    text = tok.decode(ids + new)
    logging.info("output: %r", text)
    return text


if __name__ == "__main__":
    main()
