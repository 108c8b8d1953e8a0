"""BERT question answering (SQuAD) through the ONNX importer.

    python bert-squad.py --model bertsquad-10.onnx --vocab vocab.txt \
        --question "..." --context "..."
    python bert-squad.py                    # synthetic BERT-base, no files

Without ``--model`` a BERT-base encoder with a span head is built as an ONNX
graph with random weights (text_models.bert_encoder: 12 layers, hidden 768,
12 heads, sequence 256) and the vocabulary is made from the words of the
question and context: the whole path -- WordPiece windows, import with the
fused attention / layer-norm / GELU operators, inference, span extraction --
runs; the answer of random weights is of course arbitrary.

Workload: reference examples/onnx/bert/bert-squad.py:102-165.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))
import squad_text  # noqa: E402
import text_models  # noqa: E402
from singa_amd import onnx_proto as onnx  # noqa: E402
from singa_amd import tensor  # noqa: E402

CONTEXT = ("In its early years, the new convention center failed to meet "
           "attendance and revenue expectations. By 2002, many Silicon Valley "
           "businesses were choosing the much larger Moscone Center in San "
           "Francisco over the San Jose Convention Center due to the latter's "
           "limited space. A ballot measure to finance an expansion via a "
           "hotel tax failed to reach the required two-thirds majority to "
           "pass.")
QUESTION = "Where is the Moscone Center located?"


def answer(model, wins, words, dev, zoo_layout):
    """Run every window through the imported graph and rank the spans."""
    starts, ends = [], []
    m = None
    for i, w in enumerate(wins):
        feeds = [w.input_ids[None], w.input_mask[None], w.segment_ids[None]]
        if zoo_layout:
            # bertsquad-10 inputs: unique_ids, segment_ids, input_mask, ids
            feeds = [np.array([i], np.int64), w.segment_ids[None],
                     w.input_mask[None], w.input_ids[None]]
        ts = [tensor.from_numpy(np.ascontiguousarray(a.astype(np.int32)), dev)
              for a in feeds]
        if m is None:
            m = text_models.TextModel(model)
            m.compile(ts, is_train=False, use_graph=False, sequential=True)
        t0 = time.perf_counter()
        outs = [tensor.to_numpy(y) for y in m.forward(*ts)]
        logging.info("window %d: %.1f ms", i,
                     (time.perf_counter() - t0) * 1e3)
        if zoo_layout:          # unstack_end_logits, unstack_start_logits, ids
            ends.append(outs[0].reshape(-1))
            starts.append(outs[1].reshape(-1))
        else:                   # sequence_output, logits (1, S, 2)
            logits = outs[-1].reshape(-1, 2)
            starts.append(logits[:, 0])
            ends.append(logits[:, 1])
    return squad_text.best_spans(wins, words, starts, ends)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", help="bertsquad .onnx file")
    ap.add_argument("--vocab", help="vocab.txt of the checkpoint")
    ap.add_argument("--question", default=QUESTION)
    ap.add_argument("--context", default=CONTEXT)
    ap.add_argument("--max-seq", type=int, default=256)
    ap.add_argument("--layers", type=int, default=12,
                    help="depth of the synthetic encoder")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    dev = text_models.create_device()
    if args.model:
        vocab = squad_text.load_vocab(args.vocab)
        model = onnx.load(args.model)
    else:
        vocab = squad_text.build_vocab([args.question, args.context])
        logging.info("no model file: BERT-base with random weights, "
                     "%d-word vocabulary", len(vocab))
        model, _ = text_models.bert_encoder(batch=1, seq=args.max_seq,
                                            layers=args.layers,
                                            vocab=max(len(vocab), 64))
    wins, words = squad_text.windows(args.question, args.context, vocab,
                                     max_seq=args.max_seq)
    spans = answer(model, wins, words, dev, zoo_layout=bool(args.model))
    for text, score in spans:
        logging.info("answer (score %.3f): %s", score, text)
    return spans


if __name__ == "__main__":
    main()
