"""Model zoo: every network builds, trains one step in graph mode and the
loss is finite; small ones are trained until the loss drops (reference
examples/cnn/train_cnn.py, examples/mlp/model.py)."""
import os
import sys
import numpy as np
import pytest

from singa_amd import models, opt, tensor


def _step(dev, name, shape, classes, steps=2, graph=True, **kw):
    rs = np.random.RandomState(0)
    x = rs.randn(*shape).astype(np.float32)
    y = rs.randint(0, classes, shape[0]).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    m = models.create(name, num_classes=classes, **kw)
    m.set_optimizer(opt.SGD(lr=0.01, momentum=0.9))
    m.compile([tx], is_train=True, use_graph=graph, sequential=False)
    losses = []
    for _ in range(steps):
        out, loss = m(tx, ty, 'plain', None)
        losses.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    dev.EnableGraph(False)
    dev.ResetGraph()
    assert out.shape == (shape[0], classes)
    assert np.all(np.isfinite(losses))
    return losses, m


def test_mlp_learns(dev):
    # the initial weights come from the device's own generator, so the exact
    # trajectory differs between host and GPU: train long enough for any seed
    losses, _ = _step(dev, "mlp", (32, 10), 4, steps=120, perceptron_size=32)
    assert losses[-1] < 0.7 * losses[0]


def test_cnn_learns(dev):
    losses, m = _step(dev, "cnn", (8, 1, 28, 28), 10, steps=12,
                      num_channels=1)
    assert losses[-1] < losses[0]
    assert len(m.get_params()) == 8


def test_resnet18_step(dev):
    losses, m = _step(dev, "resnet18", (2, 3, 224, 224), 10, steps=2)
    assert len(m.get_params()) == 62


def test_alexnet_step(dev):
    _step(dev, "alexnet", (2, 3, 224, 224), 10, steps=2, num_channels=3)


@pytest.mark.gpu
def test_xception_step(gpu_dev):
    _step(gpu_dev, "xceptionnet", (2, 3, 299, 299), 10, steps=2)


@pytest.mark.gpu
def test_resnet50_step(gpu_dev):
    _step(gpu_dev, "resnet50", (4, 3, 224, 224), 1000, steps=3)


def test_xception_builds_and_runs_forward(cpu_dev):
    from singa_amd import autograd
    autograd.training = False
    try:
        m = models.create("xceptionnet", num_classes=5)
        x = tensor.Tensor(device=cpu_dev, data=np.random.randn(
            1, 3, 299, 299).astype(np.float32))
        m.compile([x], is_train=False, use_graph=False, sequential=False)
        out = m(x)
        assert out.shape == (1, 5)
        assert len(m.get_params()) > 150
    finally:
        autograd.training = True


@pytest.mark.gpu
def test_half_precision_training_step():
    """fp16 activations / weights with fp32 BatchNorm parameters train through
    the same graph path (reference train_cnn.py -p float16)."""
    from singa_amd import device
    dev = device.create_cuda_gpu()
    rs = np.random.RandomState(0)
    x = rs.randn(4, 3, 224, 224).astype(np.float16)
    y = rs.randint(0, 10, 4).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x)
    ty = tensor.Tensor(device=dev, data=y)
    assert tx.dtype == tensor.float16
    m = models.create("resnet18", num_classes=10)
    m.set_optimizer(opt.SGD(lr=0.01, momentum=0.9, dtype=tensor.float16))
    m.compile([tx], is_train=True, use_graph=True, sequential=False)
    losses = []
    for _ in range(4):
        out, loss = m(tx, ty, 'plain', None)
        losses.append(float(tensor.to_numpy(loss).astype(np.float32)
                            .reshape(-1)[0]))
    dev.EnableGraph(False)
    dev.ResetGraph()
    assert np.all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_bfloat16_training_step():
    """bf16 activations / weights (an addition over the reference's fp16 mode):
    bf16 MFMA convolutions and GEMMs with float accumulation, float32
    BatchNorm parameters, float32 master weights in the optimizer."""
    from singa_amd import device
    dev = device.create_cuda_gpu()
    rs = np.random.RandomState(0)
    x = rs.randn(4, 3, 224, 224).astype(np.float32)
    y = rs.randint(0, 10, 4).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x).as_type(tensor.bfloat16)
    ty = tensor.Tensor(device=dev, data=y)
    assert tx.dtype == tensor.bfloat16
    m = models.create("resnet18", num_classes=10)
    m.set_optimizer(opt.SGD(lr=0.01, momentum=0.9))
    m.compile([tx], is_train=True, use_graph=True, sequential=False)
    losses = []
    for _ in range(4):
        out, loss = m(tx, ty, 'plain', None)
        assert out.dtype == tensor.bfloat16
        losses.append(float(tensor.to_numpy(loss.as_type(tensor.float32))
                            .reshape(-1)[0]))
    dev.EnableGraph(False)
    dev.ResetGraph()
    assert np.all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_training_with_winograd_routes_follows_direct_kernels():
    """ResNet-18, 300 replayed float32 steps on the same data and seed, once
    on the direct convolution kernels and once with the measured Winograd
    routes: some layers take a transform, the loss curves stay together."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(
        [sys.executable, "-u",
         os.path.join(root, "tools", "check_winograd_training.py")],
        capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "WINOGRAD-TRAINING-OK" in out.stdout, (
        out.stdout[-3000:] + out.stderr[-2000:])
