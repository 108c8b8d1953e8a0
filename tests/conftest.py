import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: test needs an MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def gpu_dev():
    """One GPU device for the whole session (devices are cached per GPU id)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from singa_amd import device
    return device.create_cuda_gpu_on(0)


@pytest.fixture()
def cpu_dev():
    from singa_amd import device
    return device.get_default_device()


def _dev_params():
    return ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=_dev_params())
def dev(request):
    """Every test taking ``dev`` runs on the host and, under ``-m gpu``, on
    the MI355X through the HIP kernels."""
    from singa_amd import device
    if request.param == "cpu":
        return device.get_default_device()
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = device.create_cuda_gpu_on(0)
    return d
