"""Scoring architectures in several worker processes (role of the reference's
src/eva_engine/phase1/concurrent_evaluator.py): the parent hands architecture
ids to a pool over a queue, every worker owns a host device and the same
mini-batch, results come back as (arch id, score, seconds)."""
import multiprocessing as mp
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(space, batch, metric, jobs, results):
    os.environ["CUDA_VISIBLE_DEVICES"] = ""     # workers score on the host
    sys.path[:0] = [os.path.join(_HERE, ".."),
                    os.path.join(_HERE, "..", "..", "..")]
    from singa_amd import device
    from trails import evaluators
    dev = device.get_default_device()
    ev = evaluators.P1Evaluator(space, dev, batch, metric=metric)
    while True:
        arch = jobs.get()
        if arch is None:
            return
        t0 = time.perf_counter()
        try:
            results.put((arch, ev.p1_evaluate(arch),
                         time.perf_counter() - t0))
        except Exception as e:          # noqa: BLE001 (reported to the parent)
            results.put((arch, e, time.perf_counter() - t0))


class ConcurrentEvaluator(object):

    def __init__(self, space, batch, metric="express_flow", workers=2):
        ctx = mp.get_context("spawn")
        self.jobs, self.results = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_worker, args=(
            space, batch, metric, self.jobs, self.results), daemon=True)
            for _ in range(workers)]
        for p in self.procs:
            p.start()

    def score(self, arch_ids, timeout=300):
        """{arch id: score} for all of ``arch_ids``."""
        arch_ids = list(arch_ids)
        for a in arch_ids:
            self.jobs.put(a)
        out = {}
        for _ in arch_ids:
            arch, score, _ = self.results.get(timeout=timeout)
            if isinstance(score, Exception):
                raise score
            out[arch] = score
        return out

    def close(self):
        for _ in self.procs:
            self.jobs.put(None)
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
