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
