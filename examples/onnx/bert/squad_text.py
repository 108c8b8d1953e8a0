"""Text side of extractive question answering with BERT: WordPiece
tokenisation from a ``vocab.txt``, packing of (question, context) pairs into
fixed-length windows, and extraction of the best answer spans from the start /
end logits.  Plain Python, no tokenizer library.

Role: reference examples/onnx/bert/tokenization.py and run_onnx_squad.py.
"""
import collections
import unicodedata

import numpy as np

Window = collections.namedtuple(
    "Window", "input_ids input_mask segment_ids tokens token_to_word "
              "context_start")


def load_vocab(path):
    with open(path, encoding="utf-8") as f:
        return {line.rstrip("\n"): i for i, line in enumerate(f)}


def build_vocab(texts, specials=("[PAD]", "[UNK]", "[CLS]", "[SEP]")):
    """A word-level vocabulary of the given texts (for runs without a
    ``vocab.txt``): every lower-cased word and punctuation mark."""
    vocab = {s: i for i, s in enumerate(specials)}
    for t in texts:
        for w in basic_tokens(t, True):
            vocab.setdefault(w, len(vocab))
    return vocab


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def basic_tokens(text, lower=True):
    """Whitespace split, accents stripped, punctuation as separate tokens."""
    out, word = [], []

    def flush():
        if word:
            out.append("".join(word))
            del word[:]
    if lower:
        text = unicodedata.normalize("NFD", text.lower())
    for ch in text:
        cat = unicodedata.category(ch)
        if cat == "Mn" and lower:
            continue
        if ch.isspace() or cat in ("Cc", "Cf"):
            flush()
        elif _is_punct(ch):
            flush()
            out.append(ch)
        else:
            word.append(ch)
    flush()
    return out


def wordpiece(word, vocab, unk="[UNK]", max_chars=100):
    """Greedy longest-match-first split of one word into vocabulary pieces
    (continuations carry the ``##`` prefix)."""
    if len(word) > max_chars:
        return [unk]
    pieces, start = [], 0
    while start < len(word):
        end = len(word)
        piece = None
        while start < end:
            cand = ("##" if start else "") + word[start:end]
            if cand in vocab:
                piece = cand
                break
            end -= 1
        if piece is None:
            return [unk]
        pieces.append(piece)
        start = end
    return pieces


def windows(question, context, vocab, max_seq=256, max_query=64, stride=128,
            lower=True):
    """Pack ``[CLS] question [SEP] context-span [SEP]`` windows of ``max_seq``
    tokens; long contexts are covered by overlapping spans ``stride`` apart.
    ``token_to_word`` maps a window position to the index of the context word
    it came from (-1 elsewhere)."""
    q = [p for w in basic_tokens(question, lower)
         for p in wordpiece(w, vocab)][:max_query]
    words = context.split()
    pieces, owner = [], []
    for wi, w in enumerate(words):
        for t in basic_tokens(w, lower):
            for p in wordpiece(t, vocab):
                pieces.append(p)
                owner.append(wi)
    room = max_seq - len(q) - 3
    assert room > 0, "question leaves no room for the context"
    out, start = [], 0
    while True:
        span = pieces[start:start + room]
        toks = ["[CLS]"] + q + ["[SEP]"] + span + ["[SEP]"]
        seg = [0] * (len(q) + 2) + [1] * (len(span) + 1)
        t2w = [-1] * (len(q) + 2) + owner[start:start + room] + [-1]
        ids = [vocab.get(t, vocab.get("[UNK]", 0)) for t in toks]
        pad = max_seq - len(ids)
        out.append(Window(np.array(ids + [0] * pad, np.int64),
                          np.array([1] * len(ids) + [0] * pad, np.int64),
                          np.array(seg + [0] * pad, np.int64),
                          toks, t2w + [-1] * pad, len(q) + 2))
        if start + room >= len(pieces):
            break
        start += min(stride, room)
    return out, words


def best_spans(wins, words, start_logits, end_logits, n_best=5, max_answer=30):
    """[(text, score)] of the ``n_best`` highest scoring (start, end) pairs
    over all windows; a span lies inside the context and has at most
    ``max_answer`` tokens."""
    cands = []
    for wi, w in enumerate(wins):
        s_log, e_log = np.asarray(start_logits[wi]), np.asarray(end_logits[wi])
        top_s = np.argsort(-s_log)[:n_best * 2]
        top_e = np.argsort(-e_log)[:n_best * 2]
        for s in top_s:
            for e in top_e:
                if e < s or e - s + 1 > max_answer:
                    continue
                if w.token_to_word[s] < 0 or w.token_to_word[e] < 0:
                    continue
                cands.append((float(s_log[s] + e_log[e]),
                              w.token_to_word[s], w.token_to_word[e]))
    cands.sort(reverse=True)
    seen, out = set(), []
    for score, ws, we in cands:
        if (ws, we) in seen:
            continue
        seen.add((ws, we))
        out.append((" ".join(words[ws:we + 1]), score))
        if len(out) == n_best:
            break
    return out
