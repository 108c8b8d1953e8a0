"""Train and evaluate the answer-selection model with the ranking loss
(counterpart of the reference's examples/qabot/qabot_train.py:31-159)."""
import argparse
import os
import random
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..", "..")]

from singa_amd import autograd, device, opt, tensor  # noqa: E402
from qabot_data import (limit_encode_eval, limit_encode_train,  # noqa: E402
                        prepare_data)
from qabot_model import QAModel_maxpooling  # noqa: E402


def do_train(m, tq, ta, train, meta, args):
    m.train()
    last = None
    for epoch in range(args.epochs):
        total, start = 0.0, time.time()
        q, ans_p, ans_n = limit_encode_train(
            train, meta["label2answer"], meta["idx2word"], args.q_seq_limit,
            args.ans_seq_limit, meta["idx2vec"])
        bs = args.bs
        for i in range(len(q) // bs):
            s = slice(i * bs, (i + 1) * bs)
            tq.copy_from_numpy(q[s])
            ta.copy_from_numpy(np.concatenate([ans_p[s], ans_n[s]]))
            p_sim, n_sim = m.forward(tq, ta)
            loss = autograd.ranking_loss(p_sim, n_sim)
            m.optimizer(loss)
            total += float(tensor.to_numpy(loss).reshape(-1)[0])
        last = total * bs / len(q)
        print("epoch %d, time used %d sec, loss: " %
              (epoch, time.time() - start), last, flush=True)
    return last


def do_eval(m, tq, ta, test, meta, args):
    q, candis, ans_count = limit_encode_eval(
        test, meta["label2answer"], meta["idx2word"], args.q_seq_limit,
        args.ans_seq_limit, meta["idx2vec"], args.number_of_candidates)
    m.eval()
    pool = candis.shape[1]
    correct, start = 0, time.time()
    for i in range(len(q)):
        # the question is repeated once per answer of one half of the pool
        tq.copy_from_numpy(np.repeat(q[i][None], pool // 2, axis=0))
        ta.copy_from_numpy(candis[i])
        first, second = m.forward(tq, ta)
        scores = np.concatenate((tensor.to_numpy(first).reshape(-1),
                                 tensor.to_numpy(second).reshape(-1)))
        if int(np.argmax(scores)) < ans_count[i]:
            correct += 1
    acc = correct / max(len(q), 1)
    print("eval top %s " % pool, " accuracy", acc,
          " time used %d sec" % (time.time() - start))
    return acc


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--max-epoch", default=30, type=int,
                        dest="epochs")
    parser.add_argument("-b", "--batch-size", default=50, type=int, dest="bs")
    parser.add_argument("-l", "--learning-rate", default=0.01, type=float,
                        dest="lr")
    parser.add_argument("-i", "--device-id", default=0, type=int,
                        dest="device_id")
    args = parser.parse_args(argv)
    args.hid_s = 64
    args.q_seq_limit = 10
    args.ans_seq_limit = 50
    args.number_of_candidates = args.bs * 2
    assert args.number_of_candidates <= 100

    train_raw, test_raw, label2answer, idx2word, idx2vec = prepare_data()
    meta = {"label2answer": label2answer, "idx2word": idx2word,
            "idx2vec": idx2vec}
    args.embed_size = idx2vec.shape[1]

    dev = device.create_cuda_gpu_on(args.device_id) if device.get_num_gpus() \
        else device.get_default_device()
    tq = tensor.random((args.bs, args.q_seq_limit, args.embed_size), dev)
    ta = tensor.random((args.bs * 2, args.ans_seq_limit, args.embed_size),
                       dev)
    m = QAModel_maxpooling(args.hid_s, q_seq=args.q_seq_limit,
                           a_seq=args.ans_seq_limit)
    m.compile([tq, ta], is_train=True, use_graph=False, sequential=False)
    m.optimizer = opt.SGD(args.lr, 0.9)

    print("training...")
    do_train(m, tq, ta, train_raw, meta, args)
    print("Eval with train data...")
    do_eval(m, tq, ta, random.sample(train_raw, min(2000, len(train_raw)) // 4),
            meta, args)
    print("Eval with test data...")
    return do_eval(m, tq, ta, test_raw, meta, args)


if __name__ == "__main__":
    main()
