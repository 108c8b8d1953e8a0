"""Caffe import (singa_amd/converter.py): schema against the reference's
src/proto/caffe.proto, a network with every supported layer type against a
torch reference built from the same blobs, the .caffemodel round trip, and
the solver translation."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from singa_amd import converter, device, opt, tensor

NET = """
name: "tiny"
layer { name: "data" type: "Input" top: "data"
        input_param { shape { dim: 2 dim: 3 dim: 15 dim: 15 } } }
layer { name: "conv1" type: "Convolution" bottom: "data" top: "conv1"
  param { lr_mult: 1 } param { lr_mult: 2 }
  convolution_param { num_output: 8 kernel_size: 3 pad: 1 stride: 1
                      weight_filler { type: "xavier" } } }
layer { name: "bn1" type: "BatchNorm" bottom: "conv1" top: "conv1" }
layer { name: "sc1" type: "Scale" bottom: "conv1" top: "conv1"
        scale_param { bias_term: true } }
layer { name: "relu1" type: "ReLU" bottom: "conv1" top: "conv1" }
layer { name: "pool1" type: "Pooling" bottom: "conv1" top: "pool1"
        pooling_param { pool: MAX kernel_size: 3 stride: 2 } }
layer { name: "norm1" type: "LRN" bottom: "pool1" top: "norm1"
        lrn_param { local_size: 3 alpha: 0.1 beta: 0.75 } }
layer { name: "conv2a" type: "Convolution" bottom: "norm1" top: "a"
        convolution_param { num_output: 4 kernel_size: 1 } }
layer { name: "conv2b" type: "Convolution" bottom: "norm1" top: "b"
        convolution_param { num_output: 4 kernel_h: 3 kernel_w: 3 pad_h: 1
                            pad_w: 1 bias_term: false } }
layer { name: "sum" type: "Eltwise" bottom: "a" bottom: "b" top: "s"
        eltwise_param { operation: SUM } }
layer { name: "mx" type: "Eltwise" bottom: "a" bottom: "b" top: "m"
        eltwise_param { operation: MAX } }
layer { name: "cat" type: "Concat" bottom: "s" bottom: "m" top: "c" }
layer { name: "scale2" type: "Scale" bottom: "c" top: "c2"
        scale_param { bias_term: true } }
layer { name: "pool2" type: "Pooling" bottom: "c2" top: "p2"
        pooling_param { pool: AVE global_pooling: true } }
layer { name: "drop" type: "Dropout" bottom: "p2" top: "p2"
        dropout_param { dropout_ratio: 0.3 } include { phase: TRAIN } }
layer { name: "ip" type: "InnerProduct" bottom: "p2" top: "ip"
        inner_product_param { num_output: 5 } }
layer { name: "acc" type: "Accuracy" bottom: "ip" bottom: "label" top: "acc"
        include { phase: TEST } }
layer { name: "loss" type: "SoftmaxWithLoss" bottom: "ip" bottom: "label"
        top: "loss" }
"""

SOLVER = """
net: "net.prototxt"
base_lr: 0.05
momentum: 0.9
weight_decay: 0.0005
lr_policy: "step"
gamma: 0.1
stepsize: 100
max_iter: 1000
solver_mode: GPU
"""


@pytest.fixture()
def files(tmp_path):
    (tmp_path / "net.prototxt").write_text(NET)
    (tmp_path / "solver.prototxt").write_text(SOLVER)
    return tmp_path


def _compiled(conv, phase, train=False):
    net = conv.create_net(phase)
    x = tensor.from_numpy(np.zeros((2, 3, 15, 15), np.float32))
    if train:
        net.set_optimizer(conv.create_optimizer())
    net.compile([x], is_train=train, use_graph=False, sequential=False)
    return net


def test_schema_is_a_subset_of_the_reference_proto():
    path = "/root/reference/src/proto/caffe.proto"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    text = re.sub(r"//[^\n]*", "", open(path).read())
    declared = {}
    for m in re.finditer(r"message (\w+) \{", text):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        body = text[m.end():i]
        declared[m.group(1)] = {
            f.group(3): (f.group(1), f.group(2), int(f.group(4)))
            for f in re.finditer(
                r"(optional|repeated)\s+([\w.]+)\s+(\w+)\s*=\s*(\d+)", body)}
    for msg, fields in converter._SCHEMA.items():
        for (name, number, typ, label, _) in fields:
            rule, ref_type, ref_number = declared[msg][name]
            assert ref_number == number, (msg, name)
            assert (rule == "repeated") == bool(label), (msg, name)
            assert ref_type.split(".")[-1] == typ, (msg, name)


def test_phases_select_layers(files):
    conv = converter.CaffeConverter(str(files / "net.prototxt"))
    train = conv.create_net("train")
    test = conv.create_net("test")
    assert "Dropout" in [s[0] for s in train.steps]
    assert "Dropout" not in [s[0] for s in test.steps]
    assert train.input_sample_shape == (3, 15, 15)
    assert train.loss_type == "SoftmaxWithLoss" and train.loss_bottom == "ip"
    # BatchNorm + in-place Scale became one layer with affine parameters
    assert train.caffe_layers["bn1"] == ["bn1", "sc1"]
    assert not hasattr(train, "sc1")


def _torch_reference(blobs, x):
    t = {k: [torch.from_numpy(a) for a in v] for k, v in blobs.items()}
    y = F.conv2d(x, t["conv1"][0], t["conv1"][1], padding=1)
    mean, var = t["bn1"][0], t["bn1"][1]
    y = F.batch_norm(y, mean, var, t["sc1"][0], t["sc1"][1], False, 0., 1e-5)
    y = F.relu(y)
    y = F.max_pool2d(y, 3, 2, ceil_mode=True)
    y = F.local_response_norm(y, 3, alpha=0.1, beta=0.75, k=1.0)
    a = F.conv2d(y, t["conv2a"][0], t["conv2a"][1])
    b = F.conv2d(y, t["conv2b"][0], None, padding=1)
    c = torch.cat([a + b, torch.maximum(a, b)], 1)
    c = c * t["scale2"][0].view(1, -1, 1, 1) + t["scale2"][1].view(1, -1, 1, 1)
    p = c.mean((2, 3))
    return p @ t["ip"][0].t() + t["ip"][1]


def test_forward_matches_a_torch_network_built_from_the_blobs(files):
    rng = np.random.RandomState(3)
    shapes = {"conv1": [(8, 3, 3, 3), (8,)], "bn1": [(8,), (8,), (1,)],
              "sc1": [(8,), (8,)], "conv2a": [(4, 8, 1, 1), (4,)],
              "conv2b": [(4, 8, 3, 3)], "scale2": [(8,), (8,)],
              "ip": [(5, 8), (5,)]}
    blobs = {k: [rng.randn(*s).astype(np.float32) * 0.5 for s in v]
             for k, v in shapes.items()}
    blobs["bn1"][1] = np.abs(blobs["bn1"][1]) + 0.5
    # Caffe stores the accumulated statistics and their weight
    factor = 4.0
    stored = dict(blobs)
    stored["bn1"] = [blobs["bn1"][0] * factor, blobs["bn1"][1] * factor,
                     np.array([factor], np.float32)]
    trained = converter.NetParameter()
    for name, arrays in stored.items():
        lp = trained.layer.add()
        lp.name = name
        for a in arrays:
            lp.blobs.append(converter.array_blob(a))
    path = files / "tiny.caffemodel"
    path.write_bytes(trained.SerializeToString())

    conv = converter.CaffeConverter(str(files / "net.prototxt"),
                                    param_path=str(path))
    net = _compiled(conv, "test")
    used = conv.convert_params(net)
    assert sorted(used) == sorted(shapes)
    net.eval()
    x = rng.randn(2, 3, 15, 15).astype(np.float32)
    got = tensor.to_numpy(net(tensor.from_numpy(x)))
    want = _torch_reference(blobs, torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)

    # and back: the exported file restores the same function
    out = files / "exported.caffemodel"
    converter.export_caffemodel(net, str(out))
    again = _compiled(converter.CaffeConverter(
        str(files / "net.prototxt"), param_path=str(out)), "test")
    converter.CaffeConverter(str(files / "net.prototxt"),
                             param_path=str(out)).convert_params(again)
    again.eval()
    np.testing.assert_allclose(
        tensor.to_numpy(again(tensor.from_numpy(x))), got, rtol=1e-6,
        atol=1e-6)


def test_legacy_blob_dimensions():
    blob = converter.BlobProto()
    blob.num, blob.channels, blob.height, blob.width = 2, 3, 1, 1
    blob.data.extend(range(6))
    assert converter.blob_array(blob).shape == (2, 3, 1, 1)


def test_solver_translation_and_training(files):
    conv = converter.CaffeConverter(str(files / "net.prototxt"),
                                    str(files / "solver.prototxt"))
    sgd = conv.create_optimizer()
    assert isinstance(sgd, opt.SGD)
    assert isinstance(sgd.lr, opt.ExponentialDecay)
    assert sgd.lr.decay_steps == 100 and sgd.lr.staircase
    assert abs(sgd.lr.init_value - 0.05) < 1e-7
    net = _compiled(conv, "train", train=True)
    net.train()
    rng = np.random.RandomState(0)
    x = tensor.from_numpy(rng.randn(2, 3, 15, 15).astype(np.float32))
    y = tensor.from_numpy(np.array([1, 3], np.int32))
    losses = [float(tensor.to_numpy(net(x, y)[1]).ravel()[0])
              for _ in range(12)]
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("policy,extra,expect", [
    ("inv", "gamma: 0.01 power: 0.75", 0.1 * (1 + 0.01 * 50) ** -0.75),
    ("poly", "power: 2 max_iter: 100", 0.1 * (1 - 50 / 100.) ** 2),
    ("exp", "gamma: 0.99", 0.1 * 0.99 ** 50),
    ("fixed", "", 0.1)])
def test_learning_rate_policies(tmp_path, policy, extra, expect):
    (tmp_path / "s.prototxt").write_text(
        'base_lr: 0.1 lr_policy: "%s" %s type: "Adam" momentum: 0.9' %
        (policy, extra))
    conv = converter.CaffeConverter(None, str(tmp_path / "s.prototxt"))
    adam = conv.create_optimizer()
    assert isinstance(adam, opt.Adam)
    step = tensor.from_numpy(np.array([50.], np.float32))
    got = float(tensor.to_numpy(adam.lr(step)).ravel()[0])
    assert abs(got - expect) < 1e-6 * max(1, abs(expect)) + 1e-7


@pytest.mark.parametrize("size,kernel,stride,pad", [
    (16, 3, 2, 0), (32, 3, 2, 0), (13, 3, 2, 0), (14, 3, 2, 1), (7, 2, 2, 0)])
def test_max_pooling_rounds_the_window_count_up(size, kernel, stride, pad):
    p = converter._CLS["PoolingParameter"]()
    p.kernel_size, p.stride, p.pad = kernel, stride, pad
    pool = converter._CaffePooling(p)
    x = np.random.RandomState(size).randn(2, 3, size, size).astype(np.float32)
    got = tensor.to_numpy(pool(tensor.from_numpy(x)))
    want = F.max_pool2d(torch.from_numpy(x), kernel, stride, pad,
                        ceil_mode=True).numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want)


@pytest.mark.gpu
def test_caffe_network_trains_as_a_replayed_graph_on_the_gpu(files):
    dev = device.create_cuda_gpu()
    conv = converter.CaffeConverter(str(files / "net.prototxt"),
                                    str(files / "solver.prototxt"))
    net = conv.create_net("train")
    net.set_optimizer(conv.create_optimizer())
    rng = np.random.RandomState(0)
    x = tensor.Tensor((2, 3, 15, 15), dev)
    x.copy_from_numpy(rng.randn(2, 3, 15, 15).astype(np.float32))
    y = tensor.Tensor((2,), dev, tensor.int32)
    y.copy_from_numpy(np.array([1, 3], np.int32))
    net.compile([x], is_train=True, use_graph=True, sequential=False)
    net.train()
    losses = [float(tensor.to_numpy(net(x, y)[1]).ravel()[0])
              for _ in range(12)]
    assert losses[-1] < losses[0]


def test_imported_network_exports_to_onnx(tmp_path):
    """A Caffe description becomes an ONNX file through the model it was
    turned into; the ONNX backend gives the same outputs."""
    from singa_amd import sonnx
    here = os.path.dirname(os.path.abspath(__file__))
    conv = converter.CaffeConverter(
        os.path.join(here, "..", "examples", "caffe", "lenet.prototxt"))
    net = conv.create_net("test")
    x = tensor.from_numpy(
        np.random.RandomState(0).randn(2, 1, 28, 28).astype(np.float32))
    net.compile([x], is_train=False, use_graph=False, sequential=False)
    net.eval()
    y = net(x)
    proto = sonnx.to_onnx([x], [y], "lenet")
    assert [n.op_type for n in proto.graph.node].count("Conv") == 2
    path = tmp_path / "lenet.onnx"
    path.write_bytes(proto.SerializeToString())
    rep = sonnx.prepare(sonnx.onnx.load(str(path)), device="CPU")
    got = rep.run([x])[0]
    got = got if isinstance(got, np.ndarray) else tensor.to_numpy(got)
    np.testing.assert_allclose(got, tensor.to_numpy(y), rtol=1e-6, atol=1e-6)
