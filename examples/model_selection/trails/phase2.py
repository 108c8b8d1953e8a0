"""Refinement phase: spend training epochs on the K kept candidates.

    successive halving   rounds of U, U*eta, ... epochs, the best 1/eta survive
    successive rejects   K - 1 phases, the worst candidate leaves after each
    uniform              every candidate gets the same number of epochs

``schedule_epochs`` returns the exact total epoch count of a strategy without
training (the coordinator plans with it).  Role: reference
src/eva_engine/phase2/{run_sh,run_sr,run_uniform,evaluator}.py and
algo/trainer.py."""
import math
import time

import numpy as np

from singa_amd import tensor

from . import evaluators


class Trainer(object):
    """Trains an architecture for a number of epochs and returns its
    validation accuracy; keeps the measured seconds per epoch."""

    def __init__(self, space, dev, train, val, batch_size=64, lr=0.05,
                 feed=None):
        self.space, self.dev = space, dev
        self.train, self.val, self.batch_size = train, val, batch_size
        self.lr = lr
        self.feed = feed or (lambda b: b)      # batch -> (*inputs, labels)
        self.epoch_times = []

    def _accuracy(self, m, data):
        m.eval()
        hit = n = 0
        for b in data.batches(self.batch_size, drop_last=True):
            *xs, y = self.feed(b)
            out = tensor.to_numpy(m.forward(*self.space.inputs(self.dev, *xs)))
            hit += int((out.argmax(1) == y).sum())
            n += len(y)
        m.train()
        return hit / max(n, 1)

    def p2_evaluate(self, arch_id, epochs):
        first = self.feed(next(self.train.batches(self.batch_size)))
        m, _, _ = evaluators.build(self.space, arch_id, self.dev, first)
        m.train()
        for e in range(epochs):
            t0 = time.perf_counter()
            for b in self.train.batches(self.batch_size, shuffle=True, seed=e):
                *xs, y = self.feed(b)
                m.train_one_batch(*self.space.inputs(self.dev, *xs),
                                  tensor.from_numpy(y, self.dev))
            self.epoch_times.append(time.perf_counter() - t0)
        return self._accuracy(m, self.val)

    def mean_epoch_time(self):
        return float(np.mean(self.epoch_times)) if self.epoch_times else 0.0


def _rounds_sh(K, U, eta):
    """[(survivors entering the round, epochs each)]"""
    rounds, n, ep = [], K, U
    while n > 1:
        rounds.append((n, ep))
        n = max(1, n // eta)
        ep *= eta
    return rounds


def schedule_epochs(strategy, K, U, eta=3):
    if K <= 1:
        return 0 if K == 0 else U
    if strategy == "sh":
        return sum(n * ep for n, ep in _rounds_sh(K, U, eta))
    if strategy == "uniform":
        return K * U
    if strategy == "sr":
        # phase j = 1 .. K-1 trains the K - j + 1 remaining ones U more epochs
        return sum((K - j + 1) * U for j in range(1, K))
    raise ValueError(strategy)


class SuccessiveHalving(object):
    name = "sh"

    def __init__(self, trainer, eta=3):
        self.trainer, self.eta = trainer, eta

    def run(self, U, candidates):
        alive = list(candidates)
        epochs_used, ep, trace = 0, U, []
        while len(alive) > 1:
            scored = [(self.trainer.p2_evaluate(a, ep), a) for a in alive]
            epochs_used += ep * len(alive)
            scored.sort(reverse=True)
            trace.append([(a, s, ep) for s, a in scored])
            alive = [a for _, a in scored[:max(1, len(alive) // self.eta)]]
            ep *= self.eta
        return alive[0], epochs_used, trace


class SuccessiveRejects(object):
    name = "sr"

    def __init__(self, trainer):
        self.trainer = trainer

    def run(self, U, candidates):
        alive = list(candidates)
        epochs_used, trace, total = 0, [], 0
        while len(alive) > 1:
            total += U     # cumulative epochs of a surviving candidate
            scored = [(self.trainer.p2_evaluate(a, total), a) for a in alive]
            epochs_used += U * len(alive)
            scored.sort(reverse=True)
            trace.append([(a, s, total) for s, a in scored])
            alive = [a for _, a in scored[:-1]]
        return alive[0], epochs_used, trace


class UniformAllocation(object):
    name = "uniform"

    def __init__(self, trainer):
        self.trainer = trainer

    def run(self, U, candidates):
        scored = sorted(((self.trainer.p2_evaluate(a, U), a)
                         for a in candidates), reverse=True)
        return scored[0][1], U * len(candidates), \
            [[(a, s, U) for s, a in scored]]


def make(strategy, trainer, eta=3):
    if strategy == "sh":
        return SuccessiveHalving(trainer, eta)
    if strategy == "sr":
        return SuccessiveRejects(trainer)
    if strategy == "uniform":
        return UniformAllocation(trainer)
    raise ValueError(strategy)


def largest_K(strategy, budget_epochs, U, eta=3, cap=4096):
    """The most candidates whose refinement fits into ``budget_epochs``."""
    K = 0
    while K < cap and schedule_epochs(strategy, K + 1, U, eta) <= \
            budget_epochs:
        K += 1
    return K
