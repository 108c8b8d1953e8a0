"""End-to-end model selection under a time budget.

    profile   measure t1 (seconds to score one architecture) and t2 (seconds
              per training epoch) on the given data
    plan      coordinator.schedule -> N, K, U
    filter    score N architectures proposed by a sampler, keep the top K
    refine    train the K candidates with successive halving / rejects /
              uniform allocation, return the winner

``python -m trails.run_ms --budget 20`` runs it on synthetic tabular data.
Role: reference src/eva_engine/run_ms.py (RunModelSelection)."""
import argparse
import json
import time

import numpy as np

from singa_amd import device

from . import cache as cache_mod
from . import coordinator, dataset, evaluators, phase2, samplers, space


class RunModelSelection(object):

    def __init__(self, search_space, train, val, dev=None, metric="express_flow",
                 sampler="ea", strategy="sh", eta=3, batch_size=64, seed=0,
                 cache=None, dataset_name="data", feed=None):
        self.space, self.train, self.val = search_space, train, val
        self.dev = dev or device.get_default_device()
        self.metric, self.strategy, self.eta = metric, strategy, eta
        self.batch_size, self.seed = batch_size, seed
        self.feed = feed or (lambda b: b)
        first = self.feed(next(train.batches(batch_size)))
        self.p1 = evaluators.P1Evaluator(search_space, self.dev, first, metric,
                                         cache, dataset_name)
        self.trainer = phase2.Trainer(search_space, self.dev, train, val,
                                      batch_size, feed=self.feed)
        self.sampler = samplers.make(sampler, search_space, seed=seed)

    # ---- profiling ---------------------------------------------------------
    def profile_filtering(self, samples=3):
        rng = np.random.RandomState(self.seed + 1)
        for _ in range(samples):
            self.p1.p1_evaluate(self.space.random_architecture_id(rng))
        return self.p1.mean_time()

    def profile_refinement(self):
        rng = np.random.RandomState(self.seed + 2)
        self.trainer.p2_evaluate(self.space.random_architecture_id(rng), 1)
        return self.trainer.mean_epoch_time()

    # ---- phases -------------------------------------------------------------
    def filtering_phase(self, N, K):
        seen = 0
        while seen < N:
            arch = self.sampler.sample_next_arch()
            if arch is None:
                break
            if arch in self.sampler.history:
                seen += 1          # a repeated proposal still costs a turn
                continue
            self.sampler.fit_sampler(arch, self.p1.p1_evaluate(arch))
            seen += 1
        return [a for a, _ in self.sampler.best(K)]

    def refinement_phase(self, U, candidates):
        return phase2.make(self.strategy, self.trainer,
                           self.eta).run(U, candidates)

    def select_model_online(self, budget, kn_rate=8, only_phase1=False):
        t0 = time.perf_counter()
        t1 = self.profile_filtering()
        t2 = self.profile_refinement()
        left = budget - (time.perf_counter() - t0)
        try:
            plan = coordinator.schedule(max(left, 1e-3), t1, t2,
                                        self.strategy, kn_rate, self.eta,
                                        only_phase1=only_phase1,
                                        space_size=len(self.space))
        except ValueError:
            # the budget is gone after profiling: the smallest useful run
            plan = dict(K=1, U=1, N=min(kn_rate, len(self.space)),
                        phase1_time=kn_rate * t1, phase2_time=0.0)
        kept = self.filtering_phase(plan["N"], max(plan["K"], 1))
        if only_phase1 or len(kept) <= 1:
            best, epochs, trace = kept[0], 0, []
        else:
            best, epochs, trace = self.refinement_phase(plan["U"], kept)
        return dict(best_arch=best, plan=plan, kept=kept, epochs_used=epochs,
                    score_time_per_model=t1, train_time_per_epoch=t2,
                    seconds=time.perf_counter() - t0, trace=trace)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=20.0, help="seconds")
    ap.add_argument("--sampler", default="ea", choices=sorted(samplers.SAMPLERS))
    ap.add_argument("--metric", default="express_flow",
                    choices=sorted(evaluators.METRICS))
    ap.add_argument("--strategy", default="sh", choices=["sh", "sr", "uniform"])
    ap.add_argument("--libsvm", default=None, help="training file")
    ap.add_argument("--cache", default=":memory:")
    args = ap.parse_args(argv)
    data = dataset.load_libsvm(args.libsvm) if args.libsvm \
        else dataset.synthetic()
    train, val = data.split(0.8)
    sp = space.MlpSpace(data.nfield, data.nfeat)
    dev = device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()
    ms = RunModelSelection(sp, train, val, dev, args.metric, args.sampler,
                           args.strategy,
                           cache=cache_mod.ScoreCache(args.cache))
    out = ms.select_model_online(args.budget)
    out["trace"] = [[list(t) for t in r] for r in out["trace"]]
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
