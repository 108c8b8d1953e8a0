"""Question / answer-pool data for the answer-selection example.

The reference (examples/qabot/qabot_data.py) reads the InsuranceQA corpus and
GoogleNews word vectors.  ``prepare_data`` here loads that corpus when its
files are under ``$SINGA_DATA/insuranceQA`` together with a word2vec *text*
file; otherwise it synthesises a corpus with the same structure: every
question belongs to a topic, its ground-truth answers are drawn from the
topic's vocabulary, negatives from other topics.

Data model:
  train / test: list of (question word ids, [ground-truth answer labels])
  label2answer: dict label -> answer word ids
  idx2vec:      (vocab, embed) float32 table, row 0 = padding
"""
import os
import random

import numpy as np


def _pad_embed(ids, limit, idx2vec):
    out = np.zeros((limit, idx2vec.shape[1]), np.float32)
    ids = np.asarray(ids[:limit], np.int64)
    out[:len(ids)] = idx2vec[ids]
    return out


def limit_encode_train(train, label2answer, idx2word, q_seq_limit,
                       ans_seq_limit, idx2vec):
    """One (question, positive, random negative) triple per training item."""
    labels = list(label2answer.keys())
    q, pos, neg = [], [], []
    for q_ids, truth in train:
        p = random.choice(truth)
        n = random.choice(labels)
        while n in truth:
            n = random.choice(labels)
        q.append(_pad_embed(q_ids, q_seq_limit, idx2vec))
        pos.append(_pad_embed(label2answer[p], ans_seq_limit, idx2vec))
        neg.append(_pad_embed(label2answer[n], ans_seq_limit, idx2vec))
    return np.stack(q), np.stack(pos), np.stack(neg)


def limit_encode_eval(test, label2answer, idx2word, q_seq_limit,
                      ans_seq_limit, idx2vec, number_of_candidates):
    """Per question: a pool of ``number_of_candidates`` answers whose first
    ``ans_count`` entries are the ground truths."""
    labels = list(label2answer.keys())
    q, pools, counts = [], [], []
    for q_ids, truth in test:
        truth = list(truth)[:number_of_candidates // 2]
        wrong = [l for l in random.sample(labels, min(len(labels),
                                                      number_of_candidates * 2))
                 if l not in truth][:number_of_candidates - len(truth)]
        pool = truth + wrong
        while len(pool) < number_of_candidates:
            pool.append(random.choice(wrong))
        q.append(_pad_embed(q_ids, q_seq_limit, idx2vec))
        pools.append(np.stack([_pad_embed(label2answer[l], ans_seq_limit,
                                          idx2vec) for l in pool]))
        counts.append(len(truth))
    return np.stack(q), np.stack(pools), counts


def _synthetic(num_topics=20, words_per_topic=30, embed=64, n_train=600,
               n_test=100, answers_per_topic=8, seed=0):
    rng = np.random.RandomState(seed)
    vocab = 1 + num_topics * words_per_topic
    centers = rng.randn(num_topics, embed).astype(np.float32)
    idx2vec = np.zeros((vocab, embed), np.float32)
    for t in range(num_topics):
        lo = 1 + t * words_per_topic
        idx2vec[lo:lo + words_per_topic] = centers[t] + \
            0.7 * rng.randn(words_per_topic, embed).astype(np.float32)
    label2answer, topic_answers = {}, {}
    for t in range(num_topics):
        lo = 1 + t * words_per_topic
        for _ in range(answers_per_topic):
            lab = len(label2answer)
            label2answer[lab] = rng.randint(lo, lo + words_per_topic,
                                            rng.randint(8, 30)).tolist()
            topic_answers.setdefault(t, []).append(lab)

    def make(n):
        rows = []
        for _ in range(n):
            t = int(rng.randint(num_topics))
            lo = 1 + t * words_per_topic
            q = rng.randint(lo, lo + words_per_topic, rng.randint(4, 10))
            k = int(rng.randint(1, 3))
            truth = rng.choice(topic_answers[t], k, replace=False).tolist()
            rows.append((q.tolist(), truth))
        return rows

    idx2word = {i: "w%d" % i for i in range(vocab)}
    return make(n_train), make(n_test), label2answer, idx2word, idx2vec


def _parse_corpus(root, vectors_txt):
    def read_vocab():
        m = {}
        with open(os.path.join(root, "vocabulary"), encoding="utf-8") as f:
            for line in f:
                k, w = line.rstrip("\n").split("\t")
                m[k] = w
        return m

    vocab = read_vocab()
    words = sorted(set(vocab.values()))
    w2i = {w: i + 1 for i, w in enumerate(words)}
    dim = None
    table = {}
    with open(vectors_txt, encoding="utf-8", errors="ignore") as f:
        for line in f:
            parts = line.rstrip().split(" ")
            if len(parts) > 10 and parts[0] in w2i:
                table[parts[0]] = np.asarray(parts[1:], np.float32)
                dim = len(parts) - 1
    idx2vec = np.zeros((len(w2i) + 1, dim), np.float32)
    for w, v in table.items():
        idx2vec[w2i[w]] = v
    enc = lambda toks: [w2i[vocab[t]] for t in toks.split(" ") if t in vocab]  # noqa: E731
    label2answer = {}
    with open(os.path.join(root, "InsuranceQA.label2answer.token.encoded"),
              encoding="utf-8") as f:
        for line in f:
            lab, toks = line.rstrip("\n").split("\t")
            label2answer[int(lab)] = enc(toks)

    def read_split(name):
        rows = []
        with open(os.path.join(root, name), encoding="utf-8") as f:
            for line in f:
                cols = line.rstrip("\n").split("\t")
                rows.append((enc(cols[1]), [int(x) for x in
                                            cols[2].split(" ")]))
        return rows

    train = read_split("InsuranceQA.question.anslabel.token.100.pool.solr."
                       "train.encoded")
    test = read_split("InsuranceQA.question.anslabel.token.100.pool.solr."
                      "test.encoded")
    return train, test, label2answer, {i: w for w, i in w2i.items()}, idx2vec


def prepare_data():
    root = os.path.join(os.environ.get("SINGA_DATA", "/tmp"), "insuranceQA")
    vec = os.path.join(root, "vectors.txt")
    if os.path.isdir(root) and os.path.isfile(vec):
        return _parse_corpus(root, vec)
    print("InsuranceQA not found under %s: using a synthetic corpus" % root)
    return _synthetic()
