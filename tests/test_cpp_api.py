"""C++ API (device / memory pool / tensor / layers / FeedForwardNet /
optimizers / updater / IO / endpoint): runs the self-checking native test
program, on the host here and additionally on HipGPU under the gpu marker
(where every layer is compared against the host implementation)."""
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
EXE = os.path.join(ROOT, "singa_amd", "bin", "singa_cpp_test")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        from singa_amd import build
        build.build_cpp(verbose=False)
    return EXE


def _run(args, timeout):
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=timeout)
    tail = "\n".join(r.stdout.splitlines()[-40:])
    assert r.returncode == 0, tail
    assert " 0 failed" in r.stdout, tail
    return r.stdout


def test_cpp_api_host(exe):
    out = _run([exe], 600)
    for name in ("TensorMath/cpu", "LayersGradient/cpu", "FeedForwardNet/cpu",
                 "FeedForwardNet/cpu-graph", "LocalUpdater", "Rnn/cpu", "Io",
                 "EndPoint"):
        assert "[  OK  ] " + name in out


@pytest.mark.gpu
def test_cpp_api_gpu(exe):
    out = _run([exe, "--gpu"], 600)
    for name in ("TensorMath/gpu", "LayersGpuVsCpu", "Rnn/cpu+gpu",
                 "FeedForwardNet/gpu",
                 "FeedForwardNet/gpu-graph"):
        assert "[  OK  ] " + name in out
