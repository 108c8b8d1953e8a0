"""Periodic checkpoints, resuming, and the restart launcher: a two-rank host
job that loses a rank half way (fault injection) is started again by
tools/launch_restart.py, resumes from its newest checkpoint and ends with the
parameters of the run that was never interrupted."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKER = os.path.join(ROOT, "tests", "_resume_worker.py")
LAUNCHER = os.path.join(ROOT, "tools", "launch_restart.py")


def _launch(tmp, tag, extra_env, steps=10, restarts=2):
    from singa_amd.parallel.communicator import _free_port
    out = os.path.join(tmp, tag + ".json")
    env = dict(os.environ, SINGA_HEARTBEAT="0.2",
               SINGA_HEARTBEAT_TIMEOUT="3",
               SINGA_HEARTBEAT_PORT="0", **extra_env)
    env.pop("SINGA_HEARTBEAT_PORT")
    p = subprocess.run(
        [sys.executable, LAUNCHER, "--nproc", "2", "--max-restarts",
         str(restarts), "--master-port", str(_free_port()), "--grace", "10",
         WORKER, os.path.join(tmp, tag + "_ckpt"), str(steps), out],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
        timeout=240)
    return p.returncode, p.stdout, (json.load(open(out))
                                    if os.path.exists(out) else None)


def test_checkpoint_manager_files(tmp_path):
    from singa_amd import device, layer, model, opt, tensor
    from singa_amd.utils.checkpoint import CheckpointManager

    class Tiny(model.Model):
        def __init__(self):
            super(Tiny, self).__init__()
            self.fc = layer.Linear(3)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            return self.fc(x)

        def train_one_batch(self, x, y):
            out = self.forward(x)
            loss = self.loss(out, y)
            self.optimizer(loss)
            return out, loss

        def set_optimizer(self, o):
            self.optimizer = o

    dev = device.get_default_device()
    tx = tensor.from_numpy(np.random.RandomState(0).randn(4, 5)
                           .astype(np.float32), dev)
    ty = tensor.from_numpy(np.array([0, 1, 2, 1], np.int32), dev)

    def fresh():
        m = Tiny()
        sgd = opt.SGD(lr=0.1, momentum=0.9)
        m.set_optimizer(sgd)
        m.compile([tx], is_train=True, use_graph=False, sequential=False)
        return m, sgd
    m, sgd = fresh()
    man = CheckpointManager(str(tmp_path / "c"), keep=2)
    assert man.latest() is None and man.restore(m, sgd) is None
    for step in range(1, 6):
        m(tx, ty)
        man.save(m, step, sgd, {"epoch": step // 2, "best": 0.5 * step})
    assert man.steps() == [4, 5]                 # the two newest are kept
    assert not [n for n in os.listdir(str(tmp_path / "c")) if "tmp" in n]
    m(tx, ty)                                    # step 6 of the original
    want = {k: tensor.to_numpy(v) for k, v in m.get_states().items()}
    m2, sgd2 = fresh()
    extra = man.restore(m2, sgd2)
    assert extra["step"] == 5 and extra["epoch"] == 2 and extra["best"] == 2.5
    m2(tx, ty)                                   # step 6 of the resumed run
    for k, v in m2.get_states().items():         # moments came back too
        np.testing.assert_allclose(tensor.to_numpy(v), want[k], rtol=1e-6)
    # a damaged newest file is passed over for the one before
    with open(man.latest()[0], "wb") as f:
        f.write(b"not a zip file")
    m3, sgd3 = fresh()
    assert man.restore(m3, sgd3)["step"] == 4
    # ranks other than 0 write nothing
    other = CheckpointManager(str(tmp_path / "d"), rank=1)
    other.save(m, 1, sgd)
    assert not os.path.exists(str(tmp_path / "d"))


def test_job_that_loses_a_rank_restarts_and_matches_the_clean_run(tmp_path):
    tmp = str(tmp_path)
    rc, log, clean = _launch(tmp, "clean", {})
    assert rc == 0 and clean["start"] == 0 and len(clean["losses"]) == 10, log
    rc, log, hurt = _launch(tmp, "hurt", {
        "SINGA_FAULT_INJECT": "rank=1,step=5,kind=kill"})
    assert rc == 0, log
    assert "attempt 1: rank" in log and "job finished (attempt 2)" in log, log
    # the second attempt resumed from the checkpoint of step 4
    assert hurt["start"] == 4 and len(hurt["losses"]) == 6, (hurt, log)
    np.testing.assert_allclose(hurt["losses"], clean["losses"][4:], rtol=1e-5)
    for k, v in clean["states"].items():
        np.testing.assert_allclose(np.array(hurt["states"][k]), np.array(v),
                                   rtol=1e-5, atol=1e-7)


def test_launcher_gives_up(tmp_path):
    p = subprocess.run(
        [sys.executable, LAUNCHER, "--nproc", "1", "--max-restarts", "1",
         "--grace", "1", "--", "-c", "import sys; sys.exit(3)"],
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
        timeout=60)
    assert p.returncode == 1 and "giving up after 2 attempts" in p.stdout


def test_restored_state_follows_the_parameters_device():
    """State read from a checkpoint is created on the host; the first step
    moves it to where the parameter lives (a kernel must never be handed a
    host address), and the fused update refuses operands of two devices."""
    from singa_amd import device, opt, tensor
    from singa_amd.core import device as core_device
    from singa_amd.core import tensor_core

    host = device.get_default_device()
    other = core_device.CppCPU()        # stands in for a GPU: its own id
    other.id = lambda: 7
    p = tensor.from_numpy(np.ones((3, 4), np.float32), other)
    g = tensor.from_numpy(np.full((3, 4), .5, np.float32), other)
    for make, table in ((lambda: opt.SGD(lr=.1, momentum=.9), "moments"),
                        (lambda: opt.AdaGrad(lr=.1), "history")):
        o = make()
        table = table if hasattr(o, table) else o.state_names[0]
        o.set_states({"step_counter": 3, table: {
            "w": tensor.from_numpy(np.full((3, 4), 2, np.float16), host)}})
        q = p.clone()
        o.apply("w", q, g)
        adopted = getattr(o, table)["w"]
        assert adopted.data._dev is other and adopted.device.id() == 7
        assert adopted.dtype == tensor.float32
    np.testing.assert_allclose(
        tensor.to_numpy(opt_step_sgd(p, g, other)), 0.77, rtol=1e-6)
    stray = tensor.from_numpy(np.zeros((3, 4), np.float32), host)
    o = opt.SGD(lr=.1, momentum=.9)
    o.device_check(p, o.step_counter, *o._values())
    with pytest.raises(ValueError, match="operand 2 lives on another device"):
        tensor_core.SGDUpdateMulti(
            [p.data], [g.data], [stray.data], o.lr_value.data,
            o.mom_value.data, o.dam_value.data, o.decay_value.data, False)


def opt_step_sgd(p, g, dev):
    from singa_amd import device, opt, tensor
    o = opt.SGD(lr=.1, momentum=.9)
    o.set_states({"step_counter": 0, "moments": {"w": tensor.from_numpy(
        np.full((3, 4), 2, np.float32), device.get_default_device())}})
    q = p.clone()
    o.apply("w", q, g)
    return q


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["float32", "float16"])
def test_resume_under_graph_replay_on_the_gpu(tmp_path, precision):
    """Recorded (hipGraph) training with momentum and, for fp16, float32
    master weights: six steps in one go equal three steps, a checkpoint, a
    fresh process-like model restored from it, and three more steps."""
    from singa_amd import device, layer, model, opt, tensor
    from singa_amd.utils.checkpoint import CheckpointManager

    class Net(model.Model):
        def __init__(self):
            super(Net, self).__init__()
            self.conv = layer.Conv2d(8, 3, padding=1)
            self.bn = layer.BatchNorm2d()
            self.relu = layer.ReLU()
            self.flat = layer.Flatten()
            self.fc = layer.Linear(4)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            return self.fc(self.flat(self.relu(self.bn(self.conv(x)))))

        def train_one_batch(self, x, y):
            out = self.forward(x)
            loss = self.loss(out, y)
            self.optimizer(loss)
            return out, loss

        def set_optimizer(self, o):
            self.optimizer = o

    dev = device.create_cuda_gpu()
    dt = {"float32": tensor.float32, "float16": tensor.float16}[precision]
    rs = np.random.RandomState(3)
    xs = [rs.randn(8, 3, 8, 8).astype(np.float32) for _ in range(6)]
    ys = [rs.randint(0, 4, 8).astype(np.int32) for _ in range(6)]
    tx = tensor.Tensor((8, 3, 8, 8), dev, dt)
    ty = tensor.Tensor((8,), dev, tensor.int32)

    def fresh():
        dev.SetRandSeed(11)
        m = Net()
        sgd = opt.SGD(lr=0.05, momentum=0.9, dtype=dt)
        m.set_optimizer(sgd)
        m.compile([tx], is_train=True, use_graph=True, sequential=False)
        return m, sgd

    def steps(m, lo, hi):
        for i in range(lo, hi):
            tx.copy_from_numpy(xs[i].astype(np.float16 if precision ==
                                            "float16" else np.float32))
            ty.copy_from_numpy(ys[i])
            m(tx, ty)

    m, sgd = fresh()
    steps(m, 0, 6)
    want = {k: tensor.to_numpy(v).astype(np.float32)
            for k, v in m.get_states().items()}
    m1, sgd1 = fresh()
    steps(m1, 0, 3)
    man = CheckpointManager(str(tmp_path / "g"))
    man.save(m1, 3, sgd1)
    m2, sgd2 = fresh()
    assert man.restore(m2, sgd2)["step"] == 3
    steps(m2, 3, 6)
    tol = 2e-3 if precision == "float16" else 1e-5
    for k, v in m2.get_states().items():
        np.testing.assert_allclose(tensor.to_numpy(v).astype(np.float32),
                                   want[k], rtol=tol, atol=tol, err_msg=k)
