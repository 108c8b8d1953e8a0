"""Communicator and DistOpt with two ranks over gloo on the host (the collective
calls are the ones RCCL serves on the GPUs).  Pattern of the reference's
test/python/test_dist.py: every rank contributes known values and checks the
reduced result; plus end-to-end equivalence of 2-rank data-parallel training
with single-process training on the concatenated batch.
"""
import os
import sys
import traceback

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(fn, world=2, args=()):
    sys.path.insert(0, ROOT)
    from singa_amd.parallel.communicator import _free_port
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args))
             for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, ok, payload = q.get(timeout=180)
        assert ok, "rank %d failed:\n%s" % (r, payload)
        results[r] = payload
    for p in procs:
        p.join(60)
    return results


def _entry(fn, rank, world, port, q, args):
    try:
        sys.path.insert(0, ROOT)
        os.environ["CUDA_VISIBLE_DEVICES"] = ""
        from singa_amd import singa_wrap as singa
        holder = singa.NcclIdHolder(port=port)
        out = fn(rank, world, holder, *args)
        q.put((rank, True, out))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


# ------------------------------------------------------------ raw collectives
def _collectives(rank, world, holder):
    from singa_amd import singa_wrap as singa, tensor
    comm = singa.Communicator(rank, world, holder, 1 << 20)
    out = {}
    t = tensor.from_numpy(np.full((3, 4), rank + 1.0, np.float32))
    comm.synch(t.data)
    comm.wait()
    out["synch"] = tensor.to_numpy(t)
    a = tensor.from_numpy(np.arange(6, dtype=np.float32) * (rank + 1))
    b = tensor.from_numpy(np.ones((2, 2), np.float32) * (rank + 1))
    comm.fusedSynch([a.data], False)
    comm.fusedSynch([b.data], False)
    comm.fusedSynch([a.data, b.data], True)
    comm.wait()
    out["fused"] = (tensor.to_numpy(a), tensor.to_numpy(b))
    c = tensor.from_numpy(np.arange(6, dtype=np.float32) * (rank + 1))
    comm.synchHalf(c.data)
    comm.wait()
    out["half"] = tensor.to_numpy(c)
    d = tensor.from_numpy(np.arange(4, dtype=np.float32) + rank)
    e = tensor.from_numpy(np.ones(3, np.float32) * (rank + 2))
    comm.fusedSynchHalf([d.data, e.data], True)
    comm.wait()
    out["fused_half"] = (tensor.to_numpy(d), tensor.to_numpy(e))
    base = np.array([0.1, -2.0, 0.5, 3.0, -0.2, 1.0], np.float32) * (rank + 1)
    g = tensor.from_numpy(base.copy())
    acc = tensor.from_numpy(np.zeros(6, np.float32))
    comm.sparsification(g.data, acc.data, 1.5, False)
    comm.wait()
    out["thr"] = (tensor.to_numpy(g), tensor.to_numpy(acc))
    g = tensor.from_numpy(base.copy())
    comm.fusedSparsification([g.data], 0.34, True)
    comm.wait()
    out["topk"] = tensor.to_numpy(g)
    g1 = tensor.from_numpy(base[:2].copy())
    g2 = tensor.from_numpy(base[2:].copy())
    acc = tensor.from_numpy(np.zeros(6, np.float32))
    comm.fusedSparsification([g1.data, g2.data], acc.data, 0.34, True)
    comm.wait()
    out["topk_acc"] = (tensor.to_numpy(g1), tensor.to_numpy(g2),
                       tensor.to_numpy(acc))
    # one bucket announced with tensors of two element types (16-bit weight
    # gradients next to float32 batch-norm gradients)
    h = tensor.from_numpy((np.arange(5) * (rank + 1)).astype(np.float16))
    f = tensor.from_numpy(np.full(3, 0.25 * (rank + 1), np.float32))
    h2 = tensor.from_numpy(np.full(2, rank + 1.0, np.float16))
    comm.fusedSynch([h.data, f.data, h2.data], True)
    comm.wait()
    out["mixed"] = (tensor.to_numpy(h), tensor.to_numpy(f),
                    tensor.to_numpy(h2))
    # threshold selection that overflows a capped payload keeps the rest in
    # the accumulator
    comm.spars_cap = 0.5
    g = tensor.from_numpy(base.copy())
    acc = tensor.from_numpy(np.zeros(6, np.float32))
    comm.sparsification(g.data, acc.data, 0.05, False)
    comm.wait()
    out["thr_capped"] = (tensor.to_numpy(g), tensor.to_numpy(acc))
    comm.spars_cap = 1.0
    return out


def test_collectives_two_ranks():
    res = _spawn(_collectives)
    base = np.array([0.1, -2.0, 0.5, 3.0, -0.2, 1.0], np.float32)
    for r in (0, 1):
        o = res[r]
        np.testing.assert_array_equal(o["synch"], np.full((3, 4), 3.0))
        np.testing.assert_array_equal(o["fused"][0], np.arange(6) * 3.0)
        np.testing.assert_array_equal(o["fused"][1], np.full((2, 2), 3.0))
        np.testing.assert_array_equal(o["half"], np.arange(6) * 3.0)
        np.testing.assert_array_equal(o["fused_half"][0],
                                      2 * np.arange(4) + 1.0)
        np.testing.assert_array_equal(o["fused_half"][1], np.full(3, 5.0))
        # |x| >= 1.5: rank 0 sends -2, 3; rank 1 sends -4, 6, 2
        np.testing.assert_allclose(o["thr"][0], [0, -6, 0, 9, 0, 2])
        mine = base * (r + 1)
        keep = np.abs(mine) >= 1.5
        np.testing.assert_allclose(o["thr"][1], np.where(keep, 0, mine),
                                   rtol=1e-6)
        # top-3 of 6 by magnitude on both ranks: entries 1, 3, 5
        np.testing.assert_allclose(o["topk"], [0, -6, 0, 9, 0, 3])
        g1, g2, acc = o["topk_acc"]
        np.testing.assert_allclose(np.concatenate([g1, g2]),
                                   [0, -6, 0, 9, 0, 3])
        np.testing.assert_allclose(acc, mine * np.array([1, 0, 1, 0, 1, 0]),
                                   rtol=1e-6)
        h, f, h2 = o["mixed"]
        assert h.dtype == np.float16 and f.dtype == np.float32
        np.testing.assert_array_equal(h, np.arange(5) * 3.0)
        np.testing.assert_array_equal(f, np.full(3, 0.75))
        np.testing.assert_array_equal(h2, np.full(2, 3.0))
        # capacity 3 of 6: what was sent plus what stayed behind is the input
        g, acc = o["thr_capped"]
        assert np.count_nonzero(acc) == 3
        np.testing.assert_allclose(
            res[0]["thr_capped"][1] + res[1]["thr_capped"][1] + g, base * 3,
            rtol=1e-6)


# ------------------------------------------------------------ DistOpt training
def _make_model():
    from singa_amd import layer, model

    class MLP(model.Model):

        def __init__(self):
            super(MLP, self).__init__()
            self.fc1 = layer.Linear(8)
            self.relu = layer.ReLU()
            self.fc2 = layer.Linear(3)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            return self.fc2(self.relu(self.fc1(x)))

        def train_one_batch(self, x, y, dist_option, spars):
            out = self.forward(x)
            l = self.loss(out, y)
            if dist_option == "plain":
                self.optimizer(l)
            elif dist_option == "half":
                self.optimizer.backward_and_update_half(l)
            elif dist_option == "partialUpdate":
                self.optimizer.backward_and_partial_update(l)
            elif dist_option == "sparseTopK":
                self.optimizer.backward_and_sparse_update(l, topK=True,
                                                          spars=spars)
            elif dist_option == "sparseThreshold":
                self.optimizer.backward_and_sparse_update(l, topK=False,
                                                          spars=spars)
            return out, l

    return MLP()


def _data():
    rs = np.random.RandomState(0)
    return (rs.randn(16, 5).astype(np.float32),
            rs.randint(0, 3, 16).astype(np.int32))


def _init(m):
    from singa_amd import tensor
    rs = np.random.RandomState(1)
    for k, v in sorted(m.get_params().items()):
        v.copy_from_numpy(rs.randn(*v.shape).astype(np.float32) * 0.3)


def _dist_train(rank, world, holder, option, use_graph, spars):
    from singa_amd import opt, tensor, device
    dev = device.get_default_device()
    sgd = opt.DistOpt(opt.SGD(lr=0.1, momentum=0.9), nccl_id=holder,
                      local_rank=rank, world_size=world)
    x, y = _data()
    n = x.shape[0] // world
    tx = tensor.Tensor(device=dev, data=x[rank * n:(rank + 1) * n])
    ty = tensor.Tensor(device=dev, data=y[rank * n:(rank + 1) * n])
    m = _make_model()
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=use_graph, sequential=True)
    _init(m)
    for _ in range(4):
        out, l = m(tx, ty, option, spars)
    return {k: tensor.to_numpy(v) for k, v in m.get_params().items()}


def _single_train():
    sys.path.insert(0, ROOT)
    from singa_amd import opt, tensor, device
    dev = device.get_default_device()
    x, y = _data()
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    m = _make_model()
    m.set_optimizer(opt.SGD(lr=0.1, momentum=0.9))
    m.compile([tx], is_train=True, use_graph=False, sequential=False)
    _init(m)
    for _ in range(4):
        m(tx, ty, "plain", None)
    return {k: tensor.to_numpy(v) for k, v in m.get_params().items()}


@pytest.mark.parametrize("use_graph", [False, True])
def test_distopt_matches_single_process(use_graph):
    """Mean loss over 2 x 8 samples with averaged gradients == mean loss over
    16 samples in one process."""
    ref = _single_train()
    res = _spawn(_dist_train, args=("plain", use_graph, None))
    for r in (0, 1):
        for k in ref:
            np.testing.assert_allclose(res[r][k], ref[k], rtol=1e-4,
                                       atol=1e-5)


def test_distopt_half_close_to_fp32():
    ref = _single_train()
    res = _spawn(_dist_train, args=("half", False, None))
    for k in ref:
        np.testing.assert_allclose(res[0][k], ref[k], rtol=2e-2, atol=2e-3)
        np.testing.assert_array_equal(res[0][k], res[1][k])


@pytest.mark.parametrize("option,spars", [("sparseTopK", 1.0),
                                          ("sparseThreshold", 0.0)])
def test_distopt_sparse_full_density_matches(option, spars):
    """Sending every entry (top-100 % / threshold 0) is the dense all-reduce."""
    ref = _single_train()
    res = _spawn(_dist_train, args=(option, False, spars))
    for k in ref:
        np.testing.assert_allclose(res[0][k], ref[k], rtol=1e-4, atol=1e-5)


def test_sparse_residuals_survive_graph_replay():
    """Top-30 % exchange with error feedback: what is held back must be added
    the next iteration, also when the step is a replayed graph (the residual
    tensors are only touched by communication nodes; the scheduler must not
    take them for graph-internal buffers and release them)."""
    eager = _spawn(_dist_train, args=("sparseTopK", False, 0.3))
    graph = _spawn(_dist_train, args=("sparseTopK", True, 0.3))
    for k in eager[0]:
        np.testing.assert_allclose(graph[0][k], eager[0][k], rtol=1e-5,
                                   atol=1e-6)


def test_distopt_partial_update_runs_and_syncs():
    res = _spawn(_dist_train, args=("partialUpdate", False, None))
    for k in res[0]:
        assert np.all(np.isfinite(res[0][k]))


def _partial_fp16(rank, world, holder):
    from singa_amd import opt, tensor, device
    dev = device.get_default_device()
    sgd = opt.DistOpt(opt.SGD(lr=0.1, momentum=0.9), nccl_id=holder,
                      local_rank=rank, world_size=world)
    x, y = _data()
    n = x.shape[0] // world
    tx = tensor.Tensor(device=dev,
                       data=x[rank * n:(rank + 1) * n].astype(np.float16))
    ty = tensor.Tensor(device=dev, data=y[rank * n:(rank + 1) * n])
    m = _make_model()
    m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=False, sequential=True)
    rs = np.random.RandomState(1)
    for k, v in sorted(m.get_params().items()):
        assert v.dtype == tensor.float16
        v.copy_from_numpy((rs.randn(*v.shape) * 0.3).astype(np.float16))
    m(tx, ty, "partialUpdate", None)
    params = {k: tensor.to_numpy(v) for k, v in m.get_params().items()}
    masters = [tensor.to_numpy(t) for t in sgd.opt.masters.values()]
    return params, masters


def test_partial_update_reaches_fp16_master_weights():
    """The MLP's parameters form one partition, so the first iteration
    averages all of them: they must agree across the ranks, and the float32
    master copies the next step starts from must hold the averaged values."""
    res = _spawn(_partial_fp16)
    p0, m0 = res[0]
    p1, _ = res[1]
    assert len(m0) == len(p0) > 0
    for k in p0:
        np.testing.assert_array_equal(p0[k], p1[k])
    have = sorted(v.astype(np.float32).reshape(-1).tolist() for v in p0.values())
    want = sorted(v.reshape(-1).tolist() for v in m0)
    assert have == want


@pytest.mark.gpu
@pytest.mark.parametrize("option", ["plain", "partialUpdate", "sparseTopK"])
def test_failed_capture_falls_back_without_extra_collectives(option):
    """The first capture of the step (collectives inside) is made to fail:
    the step is captured again around eagerly issued collectives, and the
    loss trajectory stays the eager one (no collective ran twice, no
    parameter was averaged twice, no residual was consumed)."""
    import subprocess
    out = subprocess.run(
        [sys.executable, "-u",
         os.path.join(ROOT, "tools", "check_rccl_captured.py"), option, "fp32",
         "0", "failcapture"],
        capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "RCCL-CAPTURED-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("option,dtype,verbosity", [
    ("plain", "fp32", 0), ("plain", "fp16", 0), ("half", "fp32", 0),
    ("half", "fp16", 0),
    ("sparseTopK", "fp32", 0), ("sparseThreshold", "fp32", 0),
    ("plain", "fp32", 3)])
def test_rccl_collectives_are_captured(option, dtype, verbosity):
    """Forced single-rank RCCL group: the DistOpt step (pack kernel, RCCL
    collective on its side stream, unpack kernel, fused update) replays as
    ONE hipGraph and follows the eager single-process loss trajectory; with
    a verbosity the communication profile comes from that same replay."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(
        [sys.executable, "-u",
         os.path.join(root, "tools", "check_rccl_captured.py"), option, dtype,
         str(verbosity)],
        capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "RCCL-CAPTURED-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-3000:])
    if verbosity == 3:
        assert "Dist_" in out.stdout, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("graph,option", [
    ("1", "plain"), ("0", "plain"), ("1", "half"), ("1", "partialUpdate"),
    ("1", "sparseTopK"), ("1", "sparseThreshold")])
def test_two_ranks_share_one_gpu(graph, option):
    """The production data-parallel path (DistOpt buckets, pack / unpack
    kernels, side streams, hipGraph segments with eager collectives between
    them) with two ranks on one GPU exchanging through gloo: with identical
    batches the averaged gradient equals the single-process one, so the loss
    trajectories must agree."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MR_GRAPH=graph, MR_OPTION=option)
    out = subprocess.run(
        [sys.executable, "-u", os.path.join(root, "tools",
                                            "check_multirank_gpu.py")],
        env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "MULTIRANK-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("case,ranks", [("collectives", "4"),
                                        ("collectives", "2"),
                                        ("collectives", "8"),
                                        ("timeout", "3")])
def test_xgmi_direct_collectives(case, ranks):
    """All-reduce (one- and two-shot, three element types) and all-gather
    kernels that read the other ranks' windows through IPC mappings: ranks as
    processes sharing the GPU, results against a host sum, bit-identical on
    every rank, replayed from a hipGraph; a missing peer ends in the bounded
    wait's error, not in a hang."""
    import subprocess
    env = dict(os.environ, XG_CASE=case, XG_RANKS=ranks)
    out = subprocess.run(
        [sys.executable, "-u",
         os.path.join(ROOT, "tools", "check_xgmi_direct.py")],
        env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "XGMI-DIRECT-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("option", ["plain", "half", "sparseTopK"])
def test_two_ranks_train_through_xgmi_windows(option):
    """DistOpt with the direct collectives: buckets are packed into the
    rank's window, summed by the peer-reading kernel and unpacked from it, all
    inside the one captured hipGraph of the step; the loss trajectory is the
    single-process one."""
    import subprocess
    env = dict(os.environ, MR_GRAPH="1", MR_OPTION=option,
               SINGA_XGMI_DIRECT="1", SINGA_XGMI_WINDOW_MB="256")
    out = subprocess.run(
        [sys.executable, "-u",
         os.path.join(ROOT, "tools", "check_multirank_gpu.py")],
        env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "MULTIRANK-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-2000:])
