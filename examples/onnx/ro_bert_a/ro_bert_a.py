"""RoBERTa sequence classification through the ONNX importer.

    python ro_bert_a.py --model roberta-sequence-classification-9.onnx \
        --vocab /data/roberta-base --text "This film is so good"
    python ro_bert_a.py                     # synthetic encoder, byte vocabulary

Without ``--model`` a small BERT-style encoder is built as an ONNX graph with
random weights (text_models.bert_encoder) and a two-class head reads its first
token.

Workload: reference examples/onnx/ro_bert_a/ro_bert_a.py:36-100.
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

LABELS = ("negative", "positive")
BOS, EOS = 0, 2     # <s>, </s> of the RoBERTa vocabulary


def classify(logits):
    p = np.exp(logits - logits.max())
    p = p / p.sum()
    k = int(np.argmax(p))
    logging.info("prediction: %s (p = %.3f)", LABELS[k], p[k])
    return k, p


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", help="roberta-sequence-classification .onnx")
    ap.add_argument("--vocab", help="directory with vocab.json / merges.txt")
    ap.add_argument("--text", default="This film is so good")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    dev = text_models.create_device()
    if args.model:
        tok = bpe.BPETokenizer(args.vocab)
        ids = np.asarray([BOS] + tok.encode(args.text) + [EOS], np.int64)[None]
        logits = text_models.run(onnx.load(args.model), [ids], dev)[0]
        return classify(logits.reshape(-1))
    tok = bpe.ByteTokenizer()
    ids = [BOS] + [3 + b for b in tok.encode(args.text)] + [EOS]
    seq = len(ids)
    logging.info("no model file: encoder with random weights, byte "
                 "vocabulary, sequence %d", seq)
    model, W = text_models.bert_encoder(batch=1, seq=seq, hidden=128, heads=2,
                                        layers=2, ffn=512, vocab=3 + 256,
                                        squad_head=False)
    x = np.asarray(ids, np.int64)[None]
    seq_out = text_models.run(model, [x, np.ones_like(x), np.zeros_like(x)],
                              dev)[0]
    head = np.random.RandomState(1).randn(seq_out.shape[-1], 2) * 0.1
    return classify(seq_out[0, 0] @ head)


if __name__ == "__main__":
    main()
