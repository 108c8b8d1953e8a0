"""The model-selection application (examples/model_selection/trails):
search spaces, samplers, training-free scores, refinement strategies, the
budget coordinator, the score cache / service and the JSON entry points
behind the database functions.  Mirrors the roles of the reference's Trails
tree (examples/model_selection/Trails/internal/ml/model_selection); its
benchmark tables are not available here, so samplers are checked on synthetic
objectives and the end-to-end run on synthetic tabular data."""
import json
import os
import sys
import urllib.request

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "examples", "model_selection"))

from trails import (cache, coordinator, dataset, evaluators, phase2,  # noqa
                    pg_interface, run_ms, samplers, space)
from singa_amd import device  # noqa: E402


def test_libsvm_round_trip_and_mini_batch_json(tmp_path):
    d = dataset.synthetic(n=50, nfield=4, vocab_per_field=5)
    assert d.ids.shape == (50, 4) and d.nfeat == 20
    assert set(np.unique(d.labels)) == {0, 1}
    p = os.path.join(str(tmp_path), "d.libsvm")
    dataset.save_libsvm(d, p)
    back = dataset.load_libsvm(p, nfeat=d.nfeat)
    np.testing.assert_array_equal(back.ids, d.ids)
    np.testing.assert_array_equal(back.labels, d.labels)
    rows = [{"id": 7, "label": 1, "c1": "3:1", "c2": "9:0.5"},
            {"id": 8, "label": 0, "c1": "4:1", "c2": "11:2"}]
    mb = dataset.from_mini_batch_json(json.dumps(rows))
    np.testing.assert_array_equal(mb.ids, [[3, 9], [4, 11]])
    np.testing.assert_allclose(mb.values, [[1, 0.5], [1, 2]])
    np.testing.assert_array_equal(mb.labels, [1, 0])
    a, b = d.split(0.8)
    assert len(a) == 40 and len(b) == 10
    assert sum(len(y) for _, _, y in d.batches(16)) == 48


def test_search_spaces():
    sp = space.MlpSpace(4, 20, num_layers=3, layer_choices=(8, 16, 32))
    assert len(sp) == 27 and len(list(sp.sample_all_models())) == 27
    rng = np.random.RandomState(0)
    a = sp.random_architecture_id(rng)
    assert sp.decode(sp.encode(a)) == a
    for _ in range(20):
        m = sp.mutate_architecture(a, rng)
        assert sum(x != y for x, y in zip(sp.encode(a), sp.encode(m))) == 1
    small, big = sp.profiling("8-8-8"), sp.profiling("32-32-32")
    assert big[0] > small[0] and big[1] > small[1]
    cn = space.CnnSpace(num_stages=2, depths=(1, 2), widths=(4, 8))
    assert len(cn) == 16
    assert cn.decode(cn.encode("2x8-1x4")) == "2x8-1x4"
    from singa_amd import tensor
    dev = device.get_default_device()
    net = cn.new_architecture("2x4-1x8")
    x = tensor.from_numpy(np.random.randn(2, 3, 8, 8).astype(np.float32), dev)
    net.compile([x], is_train=False, use_graph=False, sequential=False)
    assert net.forward(x).shape == (2, 10)


def _objective(sp):
    """A score that prefers wide late layers: known optimum."""
    def f(arch_id):
        h = sp.hidden_sizes(arch_id)
        return sum((i + 1) * np.log2(w) for i, w in enumerate(h))
    return f


@pytest.mark.parametrize("name", ["random", "ea", "bohb", "rl"])
def test_samplers_find_good_architectures(name):
    sp = space.MlpSpace(4, 20, num_layers=4,
                        layer_choices=(8, 16, 32, 64, 128))
    f = _objective(sp)
    best_possible = f("128-128-128-128")
    s = samplers.make(name, sp, seed=1)
    for _ in range(120):
        a = s.sample_next_arch()
        s.fit_sampler(a, f(a))
    top = s.best(1)[0][1]
    scores = list(s.history.values())
    if name == "random":
        assert top > 0.7 * best_possible
    else:
        # guided samplers spend their proposals in the good region: a uniform
        # draw has mean 50 (of at most 70) on this objective
        assert top >= 0.9 * best_possible, (name, top, best_possible)
        assert np.mean(scores[-40:]) > 54, (name, np.mean(scores[-40:]))


def test_sequence_sampler_and_ea_aging():
    sp = space.MlpSpace(4, 20, num_layers=2, layer_choices=(8, 16))
    s = samplers.make("sequence", sp)
    got = []
    while True:
        a = s.sample_next_arch()
        if a is None:
            break
        got.append(a)
    assert got == list(sp.sample_all_models())
    ea = samplers.RegularizedEASampler(sp, population_size=3, sample_size=2)
    for i, a in enumerate(["8-8", "16-16", "8-16", "16-8"]):
        ea.fit_sampler(a, 100 - i)
    # the oldest member left although it had the best score
    assert [a for a, _ in ea.population] == ["16-16", "8-16", "16-8"]


def test_training_free_scores():
    d = dataset.synthetic(n=128, nfield=4, vocab_per_field=5)
    sp = space.MlpSpace(d.nfield, d.nfeat, num_layers=2,
                        layer_choices=(4, 64))
    dev = device.get_default_device()
    batch = next(d.batches(32))
    for name, fn in evaluators.METRICS.items():
        vals = {}
        for arch in ("4-4", "64-64"):
            m, xs, y = evaluators.build(sp, arch, dev, batch)
            before = {k: np.array(v) for k, v in
                      ((k, __import__("singa_amd").tensor.to_numpy(v))
                       for k, v in m.get_params().items())}
            vals[arch] = fn(m, xs, y)
            after = m.get_params()
            for k, v in before.items():      # scoring never changes weights
                np.testing.assert_array_equal(
                    v, __import__("singa_amd").tensor.to_numpy(after[k]))
        assert all(np.isfinite(v) for v in vals.values()), (name, vals)
        if name in ("synflow", "weight_norm", "nas_wot", "express_flow"):
            assert vals["64-64"] > vals["4-4"], (name, vals)
    ev = evaluators.P1Evaluator(sp, dev, batch, "synflow",
                                cache.ScoreCache(), "toy")
    s1 = ev.p1_evaluate("64-4")
    n = len(ev.times)
    assert ev.p1_evaluate("64-4") == s1 and len(ev.times) == n   # cache hit


class _FakeTrainer(object):
    """Accuracy grows with width and with epochs; counts the epochs."""

    def __init__(self):
        self.epochs = 0

    def p2_evaluate(self, arch_id, epochs):
        self.epochs += epochs
        w = sum(int(t) for t in arch_id.split("-"))
        return 1 - 1.0 / (w * (1 + 0.1 * epochs))


@pytest.mark.parametrize("strategy", ["sh", "sr", "uniform"])
@pytest.mark.parametrize("K,U", [(1, 2), (5, 1), (9, 2), (27, 1)])
def test_refinement_strategies_and_epoch_accounting(strategy, K, U):
    cands = ["%d-%d" % (8 * (i + 1), 8) for i in range(K)]
    tr = _FakeTrainer()
    best, used, trace = phase2.make(strategy, tr, eta=3).run(U, cands)
    assert best == cands[-1]                       # the widest wins
    if K > 1:
        assert used == phase2.schedule_epochs(strategy, K, U, 3)
        if strategy != "sr":                       # sr retrains cumulatively
            assert used == tr.epochs
        assert len(trace[-1]) >= 1
    budget = phase2.schedule_epochs(strategy, K, U, 3)
    if K > 1:
        assert phase2.largest_K(strategy, budget, U, 3) >= K
        assert phase2.largest_K(strategy, budget - 1, U, 3) < K or \
            strategy == "sh"     # halving has plateaus in K


def test_coordinator_respects_the_budget():
    for T in (5.0, 60.0, 900.0):
        for strategy in ("sh", "sr", "uniform"):
            p = coordinator.schedule(T, t1=0.01, t2=0.4, strategy=strategy,
                                     kn_rate=10)
            cost = p["N"] * 0.01 + phase2.schedule_epochs(
                strategy, p["K"], p["U"]) * 0.4
            assert cost <= T and p["N"] == p["K"] * 10 and p["K"] >= 1
            # one more candidate at the same U would not fit
            over = (p["K"] + 1) * 10 * 0.01 + phase2.schedule_epochs(
                strategy, p["K"] + 1, p["U"]) * 0.4
            assert over > T
    bigger = coordinator.schedule(600, 0.01, 0.4)["K"]
    smaller = coordinator.schedule(60, 0.01, 0.4)["K"]
    assert bigger > smaller
    p = coordinator.schedule(10, 0.5, 3.0, only_phase1=True)
    assert p["N"] == 20 and p["U"] == 0
    with pytest.raises(ValueError):
        coordinator.schedule(0.1, 1.0, 1.0)
    capped = coordinator.schedule(1e6, 0.01, 0.4, space_size=27)
    assert capped["N"] <= 27 and capped["K"] <= 27


def test_score_cache_and_service(tmp_path):
    path = os.path.join(str(tmp_path), "scores.db")
    c = cache.ScoreCache(path)
    assert c.get("frappe", "synflow", "8-8") is None
    c.put("frappe", "synflow", "8-8", 1.5)
    c.put("frappe", "synflow", "16-8", 2.5)
    c.put("frappe", "synflow", "8-8", 1.75)          # replaced
    assert cache.ScoreCache(path).get("frappe", "synflow", "8-8") == 1.75
    assert len(c) == 2
    svc = cache.ScoreService(c).start()
    try:
        base = "http://127.0.0.1:%d" % svc.port
        got = json.load(urllib.request.urlopen(
            base + "/score?dataset=frappe&metric=synflow&arch=16-8"))
        assert got == {"score": 2.5}
        req = urllib.request.Request(
            base + "/score", method="POST",
            data=json.dumps(dict(dataset="frappe", metric="synflow",
                                 arch="32-8", score=9.0)).encode())
        assert json.load(urllib.request.urlopen(req)) == {"ok": True}
        top = json.load(urllib.request.urlopen(
            base + "/top?dataset=frappe&metric=synflow&k=2"))
        assert top == [["32-8", 9.0], ["16-8", 2.5]]
    finally:
        svc.stop()


def test_end_to_end_and_database_entry_points():
    d = dataset.synthetic(n=480, nfield=4, vocab_per_field=6)
    train, val = d.split(0.8)
    sp = space.MlpSpace(d.nfield, d.nfeat, num_layers=2,
                        layer_choices=(8, 32, 64))
    ms = run_ms.RunModelSelection(sp, train, val, sampler="ea", batch_size=32)
    out = ms.select_model_online(4.0, kn_rate=2)
    assert out["best_arch"] in set(sp.sample_all_models())
    assert out["plan"]["K"] >= 1 and out["epochs_used"] >= 0
    assert out["score_time_per_model"] > 0 and out["train_time_per_epoch"] > 0
    # the JSON functions behind the UDFs, in the order a session calls them
    rows = [{"label": int(y), **{"c%d" % j: "%d:%g" % (i[j], v[j])
                                 for j in range(d.nfield)}}
            for i, v, y in zip(d.ids[:200], d.values[:200], d.labels[:200])]
    mb = json.dumps(rows)
    cfg = json.dumps({"num_layers": 2, "layer_choices": [8, 32],
                      "nfeat": d.nfeat})
    t1 = json.loads(pg_interface.profiling_filtering_phase(mb, cfg))["time"]
    t2 = json.loads(pg_interface.profiling_refinement_phase(mb, cfg))["time"]
    assert t1 > 0 and t2 > 0
    plan = json.loads(pg_interface.coordinator(t1, t2, 30.0, False, cfg))
    assert set(plan) == {"k", "u", "n"} and plan["k"] >= 1
    kept = json.loads(pg_interface.filtering_phase(mb, 4, 2, cfg))["k_models"]
    assert 1 <= len(kept) <= 2
    best = json.loads(pg_interface.refinement_phase(json.dumps({"u": 1})))
    assert best["best_arch"] in kept
    whole = json.loads(pg_interface.model_selection(mb, 3.0, cfg))
    assert "best_arch" in whole and "plan" in whole
    bad = json.loads(pg_interface.coordinator("x", 1, 1))
    assert "Errored" in bad


# ------------------------------------------------- study tooling and slicing
def test_rank_correlations_against_scipy():
    from trails import correlation
    from scipy import stats
    rng = np.random.RandomState(3)
    a = rng.randint(0, 6, 40).astype(float)       # many ties
    b = a + rng.randn(40)
    assert abs(correlation.spearman(a, b) -
               stats.spearmanr(a, b)[0]) < 1e-9
    assert abs(correlation.kendall_tau(a, b) -
               stats.kendalltau(a, b)[0]) < 1e-9
    assert correlation.spearman([1, 2, 3], [3, 2, 1]) == -1.0
    assert correlation.kendall_tau([1, 1, 1], [1, 2, 3]) == 0.0


def test_score_accuracy_study():
    from trails import correlation
    d = dataset.synthetic(n=256, nfield=4, vocab_per_field=5)
    sp = space.MlpSpace(d.nfield, d.nfeat, num_layers=2,
                        layer_choices=(4, 16, 64))
    archs = ["4-4", "16-16", "64-64"]     # widths far enough apart that no
    # draw of the initial weights reorders the norms
    device.get_default_device().SetRandSeed(5)
    rows, acc = correlation.study(sp, device.get_default_device(),
                                  next(d.batches(32)), _FakeTrainer(), archs,
                                  ["weight_norm", "synflow", "fisher"])
    assert len(acc) == 3 and [r["metric"] for r in rows]
    # the stand-in trainer's accuracy grows with width, and so do these scores
    by = {r["metric"]: r for r in rows}
    assert by["weight_norm"]["spearman"] > 0.7
    assert all(-1.0 <= r["kendall_tau"] <= 1.0 for r in rows)


def test_resource_measure(tmp_path):
    from trails import res_measure
    log = tmp_path / "res.jsonl"
    with res_measure.Measure("phase", str(log)) as m:
        np.linalg.svd(np.random.rand(200, 200))
    assert m.result["tag"] == "phase" and m.result["wall_s"] > 0
    rec = json.loads(log.read_text().strip())
    assert rec["tag"] == "phase" and rec["cpu_s"] >= 0


def test_concurrent_evaluator_matches_the_serial_one():
    from trails import concurrent
    d = dataset.synthetic(n=64, nfield=4, vocab_per_field=5)
    sp = space.MlpSpace(d.nfield, d.nfeat, num_layers=2,
                        layer_choices=(4, 16))
    batch = next(d.batches(16))
    archs = ["4-4", "4-16", "16-4", "16-16"]
    pool = concurrent.ConcurrentEvaluator(sp, batch, "weight_norm", workers=2)
    try:
        got = pool.score(archs, timeout=120)
    finally:
        pool.close()
    assert set(got) == set(archs) and all(np.isfinite(v) for v in got.values())
    # weight norms grow with the layer widths in every process alike
    assert got["16-16"] > got["4-4"]


def test_sliced_inference_entry_points(tmp_path):
    from slicing import pg_interface as spg, sams
    d = dataset.synthetic(n=512, nfield=4, vocab_per_field=5)
    dev = device.get_default_device()
    m = sams.SlicedMoE(d.nfield, d.nfeat, experts=4, top=2)
    losses = sams.train(m, dev, d, epochs=4, batch_size=32)
    assert np.mean(losses[-4:]) < np.mean(losses[:4])
    path = str(tmp_path / "sams.zip")
    m.save_states(path)
    cfg = {"nfield": d.nfield, "nfeat": d.nfeat, "model_path": path}
    assert json.loads(spg.model_inference_load_model(json.dumps(cfg)))["ok"]
    ids, values, _ = next(d.batches(8))
    rows = [{"id": i.tolist(), "value": v.tolist()}
            for i, v in zip(ids, values)]
    res = json.loads(spg.model_inference_compute(json.dumps(
        {"where_cond": {"1": int(ids[0, 1])}, "mini_batch": rows})))
    assert res["ok"], res
    p = np.array(res["result"]["probabilities"])
    assert p.shape == (8, 2) and np.allclose(p.sum(1), 1.0, atol=1e-5)
    assert len(res["result"]["experts_used"]) == 2
    # the loaded model reproduces the trained one
    pred = np.tile(sams.predicate_ids(m, {1: int(ids[0, 1])}), (8, 1))
    from singa_amd import tensor as st
    ref = st.to_numpy(m.forward(st.from_numpy(ids.astype(np.int32), dev),
                                st.from_numpy(values.astype(np.float32), dev),
                                st.from_numpy(pred, dev)))
    z = np.exp(ref - ref.max(1, keepdims=True))
    np.testing.assert_allclose(p, z / z.sum(1, keepdims=True), atol=1e-5)
    # errors come back as JSON, never as exceptions
    bad = json.loads(spg.model_inference_compute(json.dumps(
        {"mini_batch": [{"id": [0], "value": [1.0]}]})))
    assert not bad["ok"] and "fields" in bad["error"]
    assert not json.loads(spg.model_inference_compute("{not json"))["ok"]
