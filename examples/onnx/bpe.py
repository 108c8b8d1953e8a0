"""Byte-level byte-pair-encoding tokenizer of GPT-2 / RoBERTa, read from a
``vocab.json`` + ``merges.txt`` pair; plain Python.  Without those files
``ByteTokenizer`` maps text to its UTF-8 bytes (256 ids), which is enough to
drive a randomly initialised model.

Role: the ``transformers`` tokenizers the reference's examples/onnx/gpt2/
gpt2.py:39 and ro_bert_a/ro_bert_a.py:38 import.
"""
import json
import os
import re


def _byte_alphabet():
    """The printable stand-in character of every byte value."""
    keep = list(range(33, 127)) + list(range(161, 173)) + \
        list(range(174, 256))
    chars, n = keep[:], 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(keep, chars)}


# contractions, letter runs, digit runs, other runs, whitespace -- a leading
# space stays attached to the word that follows it
_PIECES = re.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?[^\W\d_]+| ?\d+|"
                     r" ?[^\s\w]+|\s+(?!\S)|\s+")


class BPETokenizer:

    def __init__(self, directory):
        with open(os.path.join(directory, "vocab.json"),
                  encoding="utf-8") as f:
            self.encoder = json.load(f)
        self.decoder = {v: k for k, v in self.encoder.items()}
        with open(os.path.join(directory, "merges.txt"),
                  encoding="utf-8") as f:
            merges = [tuple(l.split()) for l in f.read().split("\n")
                      if l and not l.startswith("#version")]
        self.rank = {m: i for i, m in enumerate(merges)}
        self.b2c = _byte_alphabet()
        self.c2b = {c: b for b, c in self.b2c.items()}
        self.cache = {}

    def _merge(self, token):
        if token in self.cache:
            return self.cache[token]
        word = list(token)
        while len(word) > 1:
            pairs = [(self.rank.get((a, b), 1 << 30), i)
                     for i, (a, b) in enumerate(zip(word, word[1:]))]
            best, at = min(pairs)
            if best == 1 << 30:
                break
            first, second = word[at], word[at + 1]
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == first and \
                        word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self.cache[token] = word
        return word

    def encode(self, text):
        ids = []
        for piece in _PIECES.findall(text):
            mapped = "".join(self.b2c[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._merge(mapped))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.c2b[c] for c in text).decode("utf-8",
                                                           errors="replace")

    @property
    def vocab_size(self):
        return len(self.encoder)


class ByteTokenizer:
    vocab_size = 256

    def encode(self, text):
        return list(text.encode("utf-8"))

    def decode(self, ids):
        return bytes(int(i) & 255 for i in ids).decode("utf-8",
                                                       errors="replace")
