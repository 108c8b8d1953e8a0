"""The ``singa`` import name resolves to the ``singa_amd`` modules."""
import numpy as np


def test_alias_modules_are_shared():
    from singa import autograd, device, layer, model, opt, tensor  # noqa: F401
    from singa import singa_wrap as singa
    import singa.utils as su
    import singa_amd
    import singa_amd.utils
    assert tensor is singa_amd.tensor
    assert su is singa_amd.utils
    assert singa is singa_amd.singa_wrap
    assert singa_amd.tensor.__spec__.name == "singa_amd.tensor"


def test_reference_style_script_runs():
    from singa import autograd, device, opt, tensor
    dev = device.get_default_device()
    x = tensor.from_numpy(np.random.randn(4, 3).astype(np.float32))
    w = tensor.Tensor((3, 2), dev)
    w.gaussian(0.0, 0.1)
    w.requires_grad = True
    w.stores_grad = True
    autograd.training = True
    y = autograd.matmul(x, w)
    loss = autograd.mse_loss(y, tensor.from_numpy(
        np.zeros((4, 2), np.float32)))
    sgd = opt.SGD(0.1)
    for p, g in autograd.backward(loss):
        sgd.apply(p.name, p, g)
    autograd.training = False
