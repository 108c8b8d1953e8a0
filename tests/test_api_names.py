"""Every public name of the reference's Python modules -- functions, classes,
module constants and the public methods of classes -- resolves in the module
of the same name here.  Read from the reference's sources with ``ast`` (none
of its code runs); skipped where the reference tree is absent."""
import ast
import importlib
import os

import pytest

REF = "/root/reference/python/singa"
MODULES = ["autograd", "data", "device", "image_tool", "initializer", "layer",
           "model", "opt", "snapshot", "sonnx", "tensor", "utils"]
# names of the reference that are imports of other packages, not its API
NOT_API = {"sonnx": {"OrderedDict", "namedtuple"}}
# steps inside the reference's own classes that nothing outside them calls:
# the decorators its metaclasses apply, and the per-node hooks of its ONNX
# exporter / importer loops (the loops here are organised differently)
INTERNAL = {"LayerMeta.init_wrapper", "LayerMeta.forward_wrapper",
            "ModelMeta.buffer_operation", "SingaFrontend.handle_special_ops",
            "SingaFrontend.singa_op_to_onnx_node",
            "SingaRep.handle_special_ops"}


def _public(name):
    return not name.startswith("_")


def _reference_names(path):
    names = {}
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.FunctionDef) and _public(node.name):
            names[node.name] = ()
        elif isinstance(node, ast.ClassDef) and _public(node.name):
            names[node.name] = tuple(
                m.name for m in node.body if isinstance(m, ast.FunctionDef)
                and (_public(m.name) or m.name == "__call__"))
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                if isinstance(t, ast.Name) and _public(t.id):
                    names[t.id] = ()
    return names


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("module", MODULES)
def test_public_names_of_the_reference_resolve(module):
    ours = importlib.import_module("singa_amd." + module)
    missing = []
    for name, methods in _reference_names(
            os.path.join(REF, module + ".py")).items():
        if name in NOT_API.get(module, ()):
            continue
        if not hasattr(ours, name):
            missing.append(name)
            continue
        cls = getattr(ours, name)
        missing += ["%s.%s" % (name, m) for m in methods
                    if not hasattr(cls, m) and
                    "%s.%s" % (name, m) not in INTERNAL]
    assert not missing, missing


def test_the_singa_package_forwards_the_same_modules():
    for module in MODULES:
        a = importlib.import_module("singa." + module)
        b = importlib.import_module("singa_amd." + module)
        assert a is b


def test_keyword_names_of_the_reference_are_accepted():
    """Parameters a caller of the reference may pass by name
    (python/singa/tensor.py: gaussian(mean, std), uniform(low, high),
    bernoulli(p), lt(t, x), contiguous(tensor); sonnx.SingaBackend.prepare(
    model, device))."""
    import numpy as np
    from singa_amd import autograd, layer, sonnx, tensor
    t = tensor.Tensor((64, 64))
    t.gaussian(mean=3.0, std=0.5)
    assert abs(float(tensor.to_numpy(t).mean()) - 3.0) < 0.1
    t.uniform(low=-1.0, high=0.0)
    a = tensor.to_numpy(t)
    assert a.min() >= -1.0 and a.max() <= 0.0
    t.bernoulli(p=0.25)
    assert abs(float(tensor.to_numpy(t).mean()) - 0.25) < 0.05
    t.uniform(-2.0, high=2.0)
    with pytest.raises(TypeError):
        t.gaussian(0.0)
    with pytest.raises(TypeError):
        t.uniform(0.0, 1.0, scale=2)
    ones = tensor.from_numpy(np.ones((2, 2), np.float32))
    assert tensor.to_numpy(tensor.lt(t=ones, x=2.0)).all()
    assert tensor.to_numpy(tensor.ge(t=ones, x=ones)).all()
    assert tensor.contiguous(tensor=ones).shape == (2, 2)
    assert tensor.from_raw_tensor(t=ones.data).shape == (2, 2)
    assert layer.Flatten(axis=1).axis == 1
    assert layer.Dropout(ratio=0.3).ratio == 0.3
    assert layer.Cat(axis=2).axis == 2
    assert autograd.CosSim.dot(ones.data, ones.data).shape() == [2] or \
        tuple(autograd.CosSim.dot(ones.data, ones.data).shape()) == (2,)
