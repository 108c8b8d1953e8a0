"""singa_amd.native: the C++ tensor library seen from Python through its C
interface, on the host device here and, on a GPU machine, on the HIP and the
OpenCL device (the same operations, compared with numpy).  The C++ self-test
covers the OpenCL device in depth (``singa_cpp_test --opencl``)."""
import os
import subprocess

import numpy as np
import pytest

from singa_amd import device, native

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _exercise(dev):
    rs = np.random.RandomState(3)
    a = rs.randn(37, 19).astype(np.float32)
    b = rs.randn(19, 23).astype(np.float32)
    ta, tb = dev.tensor(a), dev.tensor(b)
    np.testing.assert_allclose((ta @ tb).numpy(), a @ b, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose((tb.T @ ta.T).numpy(), (a @ b).T, rtol=1e-4,
                               atol=1e-4)
    np.testing.assert_allclose((ta + ta).numpy(), a + a)
    np.testing.assert_allclose((ta * ta - ta).numpy(), a * a - a, rtol=1e-6,
                               atol=1e-6)
    np.testing.assert_allclose((ta / (ta * ta).unary("sqrt")).numpy(),
                               a / np.abs(a), rtol=1e-5)
    for op, fn in (("exp", np.exp), ("tanh", np.tanh), ("abs", np.abs),
                   ("relu", lambda v: np.maximum(v, 0)),
                   ("sigmoid", lambda v: 1 / (1 + np.exp(-v))),
                   ("sign", np.sign), ("square", np.square)):
        np.testing.assert_allclose(ta.unary(op).numpy(), fn(a), rtol=1e-5,
                                   atol=1e-6)
    np.testing.assert_allclose(ta.sum(0).numpy(), a.sum(0), rtol=1e-5,
                               atol=1e-5)
    np.testing.assert_allclose(ta.sum(1).numpy(), a.sum(1), rtol=1e-5,
                               atol=1e-5)
    e = np.exp(a - a.max(1, keepdims=True))
    np.testing.assert_allclose(ta.softmax().numpy(), e / e.sum(1, keepdims=True),
                               rtol=1e-5, atol=1e-6)
    acc = dev.tensor(np.ones_like(a))
    acc.axpy(0.5, ta)
    np.testing.assert_allclose(acc.numpy(), 1 + 0.5 * a, rtol=1e-6)
    assert ta.reshape(19, 37).shape == (19, 37)
    ints = dev.tensor(np.arange(12).reshape(3, 4))
    assert ints.numpy().dtype == np.int32 and ints.numpy()[2, 3] == 11
    with pytest.raises(RuntimeError):
        ta @ ta                      # inner dimensions differ
    host = native.Device.host()
    np.testing.assert_allclose(ta.to(host).numpy(), a)
    dev.sync()


def test_host_device():
    dev = native.Device.host()
    assert dev.lang() == "host"
    _exercise(dev)


def test_opencl_queries_without_a_runtime_device():
    """On a machine whose OpenCL runtime shows no GPU the counts are zero and
    creating the device reports it; with one the device opens."""
    n = device.get_num_opencl_devices()
    assert device.get_num_opencl_platforms() == (1 if n else 0)
    if n == 0:
        with pytest.raises(RuntimeError):
            device.create_opencl_device()


@pytest.mark.gpu
def test_hip_device_through_the_c_interface():
    dev = native.Device.gpu(0)
    assert dev.lang() == "hip"
    _exercise(dev)


@pytest.mark.gpu
def test_opencl_device():
    assert device.get_num_opencl_devices() >= 1
    dev = device.create_opencl_device()
    assert dev.lang() == "opencl"
    assert "compute units" in dev.description()
    _exercise(dev)


@pytest.mark.gpu
def test_opencl_device_cpp_suite():
    exe = os.path.join(ROOT, "singa_amd", "bin", "singa_cpp_test")
    r = subprocess.run([exe, "--opencl"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for name in ("Opencl/device", "TensorMath/opencl", "LayersOpenclVsCpu",
                 "Optimizers/opencl", "LossMetric/opencl",
                 "DropoutAndMulti/opencl", "FeedForwardNet/opencl"):
        assert "[  OK  ] " + name in r.stdout


def test_opencl_sources_compile_for_gfx950():
    """The OpenCL C sources of the OpenCL device parse and type-check for
    gfx950 with the operator codes of the C++ enums as -D options (what the
    device does at run time with the runtime's own compiler); needs only the
    ROCm clang, no GPU."""
    import re
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang")
    csrc = os.path.join(ROOT, "singa_amd", "csrc")
    header = open(os.path.join(csrc, "hip", "kernels.h")).read()
    defines = []
    for enum in ("UnaryOp", "BinaryOp", "ReduceOp"):
        body = re.search(r"enum %s : int \{(.*?)\};" % enum, header,
                         re.S).group(1)
        names = [n.split("=")[0].strip() for n in body.replace("\n", " ")
                 .split(",") if n.strip()]
        defines += ["-D%s=%d" % (n, i) for i, n in enumerate(names)]
    cl_dir = os.path.join(csrc, "cpp", "src", "core", "opencl")
    sources = sorted(f for f in os.listdir(cl_dir) if f.endswith(".cl"))
    assert sources == ["nn.cl", "tensor_math.cl"]
    for f in sources:
        r = subprocess.run(
            [clang, "-x", "cl", "-cl-std=CL2.0", "-fsyntax-only", "-Werror",
             "-Xclang", "-finclude-default-header", "-target",
             "amdgcn-amd-amdhsa", "-mcpu=gfx950"] + defines +
            [os.path.join(cl_dir, f)], capture_output=True, text=True,
            timeout=120)
        assert r.returncode == 0, f + "\n" + r.stderr[-3000:]
    # the text embedded in the library is these files, line for line
    from singa_amd import build
    build._embed_opencl_sources(cl_dir)
    inc = open(os.path.join(cl_dir, "cl_sources.inc")).read()
    first = open(os.path.join(cl_dir, "nn.cl")).read().splitlines()[0]
    assert first.replace("\\", "\\\\").replace('"', '\\"') in inc
