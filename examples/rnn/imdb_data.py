"""IMDB sentiment data for the recurrent examples.

``prepare()`` turns an extracted ``aclImdb`` directory (train|test / pos|neg
/*.txt) into ``imdb_processed.npz``: integer-encoded reviews (0 = <pad>), labels
and an embedding table.  It needs nothing beyond numpy: tokens are lower-cased
alphanumeric runs minus a small stop list, the vocabulary keeps the most
frequent words, and the table comes from a word2vec-format *text* file when
one is given, else from a fixed random projection.  (The reference's
examples/rnn/imdb_data.py does the same job with nltk / gensim / bs4 / pandas
and a pickle.)

When no processed file exists ``load()`` returns a synthetic corpus with the
same structure so the training scripts run offline: two word clusters, one per
sentiment, mixed with neutral words.
"""
import os
import re

import numpy as np

download_dir = os.environ.get("SINGA_DATA", "/tmp")
preprocessed_imdb_data_fp = os.path.join(download_dir, "imdb_processed.npz")

_STOP = set("a an the and or of to in is it this that was for with as on at by "
            "be are i you he she they we his her its their".split())
_TOKEN = re.compile(r"[a-z0-9]+")


def tokenize(text):
    text = re.sub(r"<[^>]+>", " ", text.lower())
    return [w for w in _TOKEN.findall(text) if w not in _STOP]


def _pad(rows, seq_limit):
    out = np.zeros((len(rows), seq_limit), np.int32)
    for i, r in enumerate(rows):
        r = r[:seq_limit]
        out[i, :len(r)] = r
    return out


def pad_batch(b, seq_limit):
    """[(ids, label)] -> (ids (N,seq) int32, one-hot (N,2) f32, label f32)."""
    seq = _pad([list(r[0]) for r in b], seq_limit)
    senti = np.array([r[1] for r in b], np.float32)
    onehot = np.eye(2, dtype=np.float32)[senti.astype(np.int64)]
    return seq, onehot, senti


def pad_batch_2vec(b, seq_limit, embed_weights):
    """Same, with ids replaced by their embedding rows: (N, seq, dim) f32."""
    seq, onehot, senti = pad_batch(b, seq_limit)
    return np.asarray(embed_weights, np.float32)[seq], onehot, senti


def _read_reviews(root):
    data = []
    for split in ("train", "test"):
        for name, label in (("pos", 1), ("neg", 0)):
            d = os.path.join(root, split, name)
            for fn in sorted(os.listdir(d)):
                if fn.endswith(".txt"):
                    with open(os.path.join(d, fn), encoding="utf-8") as f:
                        data.append((tokenize(f.read()), label))
    return data


def _load_word_vectors(path, wanted, dim):
    table = {}
    with open(path, encoding="utf-8", errors="ignore") as f:
        for line in f:
            parts = line.rstrip().split(" ")
            if len(parts) == dim + 1 and parts[0] in wanted:
                table[parts[0]] = np.asarray(parts[1:], np.float32)
    return table


def prepare(acl_dir, vectors_txt=None, vocab_size=28240, dim=300,
            test_size=0.33, seed=42, out=preprocessed_imdb_data_fp):
    reviews = _read_reviews(acl_dir)
    counts = {}
    for toks, _ in reviews:
        for w in toks:
            counts[w] = counts.get(w, 0) + 1
    words = sorted(counts, key=lambda w: (-counts[w], w))[:vocab_size]
    w2i = {"<pad>": 0}
    for w in words:
        w2i[w] = len(w2i)
    rng = np.random.RandomState(seed)
    embed = (rng.randn(len(w2i), dim) / np.sqrt(dim)).astype(np.float32)
    embed[0] = 0
    if vectors_txt:
        for w, v in _load_word_vectors(vectors_txt, w2i, dim).items():
            embed[w2i[w]] = v
    enc = [np.asarray([w2i[w] for w in toks if w in w2i], np.int32)
           for toks, _ in reviews]
    labels = np.asarray([l for _, l in reviews], np.int32)
    order = rng.permutation(len(enc))
    nval = int(len(enc) * test_size)
    lengths = np.asarray([len(e) for e in enc], np.int64)
    np.savez_compressed(
        out, flat=np.concatenate(enc), lengths=lengths, labels=labels,
        val_idx=order[:nval], train_idx=order[nval:], embed_weights=embed,
        vocab=np.asarray(list(w2i.keys())))
    print("saved", out)


def _synthetic(n_train=1024, n_val=256, vocab=200, dim=32, seed=0):
    rng = np.random.RandomState(seed)
    embed = rng.randn(vocab, dim).astype(np.float32)
    embed[0] = 0
    third = vocab // 3

    def make(n):
        rows = []
        for _ in range(n):
            label = int(rng.randint(2))
            ln = int(rng.randint(10, 60))
            senti = rng.randint(1 + label * third, 1 + (label + 1) * third, ln)
            neutral = rng.randint(1 + 2 * third, vocab, ln)
            ids = np.where(rng.rand(ln) < 0.5, senti, neutral)
            rows.append((ids.astype(np.int32), label))
        return rows

    return {"train": make(n_train), "val": make(n_val),
            "embed_weights": embed}


def load(path=preprocessed_imdb_data_fp):
    """-> dict(train=[(ids, label)], val=[...], embed_weights=(V, dim))."""
    if not os.path.isfile(path):
        print("%s not found: using a synthetic corpus" % path)
        return _synthetic()
    z = np.load(path, allow_pickle=False)
    ends = np.cumsum(z["lengths"])
    starts = ends - z["lengths"]
    rows = [(z["flat"][s:e], int(l))
            for s, e, l in zip(starts, ends, z["labels"])]
    return {"train": [rows[i] for i in z["train_idx"]],
            "val": [rows[i] for i in z["val_idx"]],
            "embed_weights": z["embed_weights"],
            "vocab": z["vocab"]}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("acl_dir", help="extracted aclImdb directory")
    ap.add_argument("--vectors", default=None,
                    help="word2vec text-format file (optional)")
    a = ap.parse_args()
    prepare(a.acl_dir, a.vectors)
