"""ONNX export / import: autograd graphs are exported with ``sonnx.to_onnx``,
serialised, parsed back, imported with ``sonnx.prepare`` and must reproduce
the original outputs; hand-written ONNX graphs check operators and attribute
conventions against numpy (reference test/python/test_onnx.py).
"""
import numpy as np
import pytest

from singa_amd import autograd, layer, model, opt, sonnx, tensor
from singa_amd import onnx_proto as onnx
from singa_amd.onnx_proto import TensorProto, helper, numpy_helper

RS = np.random.RandomState(21)


def T(a, dev):
    return tensor.Tensor(device=dev, data=np.asarray(a))


def roundtrip(dev, fn, xs, rtol=1e-5, atol=1e-5):
    """Export fn(*tensors), re-import and compare."""
    autograd.training = False
    try:
        ts = [T(x, dev) for x in xs]
        y = fn(*ts)
        ys = y if isinstance(y, (list, tuple)) else [y]
        m = sonnx.to_onnx(ts, list(ys) if len(ys) > 1 else ys[0], "test")
        m2 = onnx.load(m.SerializeToString())
        rep = sonnx.prepare(m2, dev)
        out = rep.run([np.asarray(x) for x in xs])
        assert len(out) == len(ys)
        for o, r in zip(out, ys):
            np.testing.assert_allclose(o, tensor.to_numpy(r), rtol=rtol,
                                       atol=atol)
        return m2
    finally:
        autograd.training = True


def test_proto_roundtrip_file(tmp_path):
    w = numpy_helper.from_array(np.arange(6, dtype=np.float32).reshape(2, 3),
                                "w")
    node = helper.make_node("MatMul", ["x", "w"], ["y"], name="mm")
    g = helper.make_graph(
        [node], "g",
        [helper.make_tensor_value_info("x", TensorProto.FLOAT, [4, 2])],
        [helper.make_tensor_value_info("y", TensorProto.FLOAT, [4, 3])], [w])
    m = helper.make_model(g)
    path = str(tmp_path / "m.onnx")
    sonnx.save(m, path)
    m2 = sonnx.load(path)
    assert m2.graph.node[0].op_type == "MatMul"
    assert m2.opset_import[0].version == 11 and m2.ir_version == 6
    np.testing.assert_array_equal(numpy_helper.to_array(m2.graph.initializer[0]),
                                  np.arange(6, dtype=np.float32).reshape(2, 3))


UNARY = ["relu", "sigmoid", "tanh", "cos", "sin", "atan", "abs", "exp",
         "softplus", "softsign", "negative", "sign", "ceil", "floor", "erf",
         "identity", "reciprocal", "sqrt", "log"]


@pytest.mark.parametrize("name", UNARY)
def test_export_import_unary(dev, name):
    x = RS.uniform(0.5, 2.0, (3, 4)).astype(np.float32)
    m = roundtrip(dev, getattr(autograd, name), [x])
    assert len(m.graph.node) == 1


@pytest.mark.parametrize("name", ["add", "sub", "mul", "div", "pow"])
def test_export_import_binary(dev, name):
    a = RS.uniform(0.5, 2.0, (3, 4)).astype(np.float32)
    b = RS.uniform(0.5, 2.0, (4,)).astype(np.float32)
    roundtrip(dev, getattr(autograd, name), [a, b])


def test_export_import_attribute_ops(dev):
    x = RS.randn(2, 3, 4).astype(np.float32)
    roundtrip(dev, lambda t: autograd.leakyrelu(t, 0.1), [x])
    roundtrip(dev, lambda t: autograd.elu(t, 0.5), [x])
    roundtrip(dev, lambda t: autograd.selu(t), [x])
    roundtrip(dev, lambda t: autograd.hardsigmoid(t, 0.3, 0.4), [x])
    roundtrip(dev, lambda t: autograd.softmax(t, 2), [x])
    roundtrip(dev, lambda t: autograd.flatten(t, 2), [x])
    roundtrip(dev, lambda t: autograd.transpose(t, (2, 0, 1)), [x])
    roundtrip(dev, lambda t: autograd.unsqueeze(t, [0]), [x])
    roundtrip(dev, lambda t: autograd.squeeze(t, [1]), [x[:, :1]])
    roundtrip(dev, lambda t: autograd.reduce_sum(t, [1], 1), [x])
    roundtrip(dev, lambda t: autograd.reduce_mean(t, [0, 2], 0), [x])
    roundtrip(dev, lambda t: autograd.reshape(t, (6, 4)), [x])
    roundtrip(dev, lambda t: autograd.clip(t, -0.5, 0.5), [x])
    roundtrip(dev, lambda t: autograd.slice(t, [0, 1], [2, 3], [1, 2]), [x])
    roundtrip(dev, lambda t: autograd.tile(t, [1, 2, 1]), [x])
    roundtrip(dev, lambda t: autograd.expand(t, [2, 2, 3, 4]), [x])
    roundtrip(dev, lambda t: autograd.gather(t, 1, [0, 2]), [x])
    roundtrip(dev, lambda t: autograd.pad(t, "constant", [0, 1, 0, 0, 1, 2],
                                          0.5), [x])
    roundtrip(dev, lambda a, b: autograd.cat([a, b], 1), [x, x])
    roundtrip(dev, lambda t: autograd.cast(t, tensor.int32), [x * 3])
    roundtrip(dev, lambda t: list(autograd.split(t, 2, [1, 3])), [x])
    x4 = RS.randn(1, 4, 2, 3).astype(np.float32)
    roundtrip(dev, lambda t: autograd.depth_to_space(t, 2), [x4])
    roundtrip(dev, lambda t: autograd.upsample(t, "nearest", [1, 1, 2, 2]),
              [x4])
    roundtrip(dev, autograd.globalaveragepool, [x4])
    c = (RS.rand(2, 3, 4) > 0.5).astype(np.float32)
    roundtrip(dev, lambda a, b: autograd.where(a, b, T(c, dev)), [x, -x])


def test_export_import_linear_gemm(dev):
    x = RS.randn(5, 6).astype(np.float32)
    lin = layer.Linear(4)
    roundtrip(dev, lambda t: lin(t), [x], rtol=1e-4)
    w = tensor.Tensor(device=dev, data=RS.randn(6, 3).astype(np.float32),
                      requires_grad=True, stores_grad=True)
    b = tensor.Tensor(device=dev, data=RS.randn(1, 3).astype(np.float32),
                      requires_grad=True, stores_grad=True)
    roundtrip(dev, lambda t: autograd.gemm(t, w, b, 0.5, 2.0, 0, 0), [x],
              rtol=1e-4)


class CNN(model.Model):

    def __init__(self):
        super(CNN, self).__init__()
        self.conv1 = layer.Conv2d(8, 3, padding=1)
        self.bn1 = layer.BatchNorm2d()
        self.relu = layer.ReLU()
        self.pool1 = layer.MaxPool2d(2, 2)
        self.conv2 = layer.Conv2d(8, 3, stride=2, padding=1, bias=False)
        self.pool2 = layer.AvgPool2d(2, 2)
        self.flat = layer.Flatten()
        self.fc = layer.Linear(5)
        self.sm = layer.SoftMax()

    def forward(self, x):
        y = self.pool1(self.relu(self.bn1(self.conv1(x))))
        y = self.pool2(self.relu(self.conv2(y)))
        return self.sm(self.fc(self.flat(y)))


def test_export_import_cnn(dev):
    x = RS.randn(2, 3, 16, 16).astype(np.float32)
    net = CNN()
    tx = T(x, dev)
    net.compile([tx], is_train=False, use_graph=False, sequential=False)
    # non-trivial running statistics
    net.bn1.running_mean.copy_from_numpy(RS.randn(8).astype(np.float32) * 0.1)
    net.bn1.running_var.copy_from_numpy(
        RS.uniform(0.5, 1.5, 8).astype(np.float32))
    net.eval()
    m = roundtrip(dev, lambda t: net.forward(t), [x], rtol=1e-4, atol=1e-5)
    types = [n.op_type for n in m.graph.node]
    for t in ("Conv", "BatchNormalization", "Relu", "MaxPool", "AveragePool",
              "Flatten", "MatMul", "Add", "Softmax"):
        assert t in types


def test_sonnx_model_finetune(dev):
    """Imported graph as the backbone of a trainable Model (reference
    examples/onnx/training/train.py)."""
    x = RS.randn(8, 3, 16, 16).astype(np.float32)
    y = RS.randint(0, 3, 8).astype(np.int32)
    net = CNN()
    tx = T(x, dev)
    net.compile([tx], is_train=False, use_graph=False, sequential=False)
    net.eval()
    autograd.training = False
    out = net.forward(tx)
    proto = onnx.load(sonnx.to_onnx([tx], out, "cnn").SerializeToString())
    autograd.training = True

    class FineTune(sonnx.SONNXModel):

        def __init__(self, onnx_model):
            super(FineTune, self).__init__(onnx_model)
            self.head = layer.Linear(3)
            self.loss = layer.SoftMaxCrossEntropy()

        def forward(self, x):
            # stop before the exported classifier: last 3 nodes are
            # MatMul, Add, Softmax
            feat = super(FineTune, self).forward(x, last_layers=-4)[0]
            return self.head(feat)

        def train_one_batch(self, x, y):
            out = self.forward(x)
            l = self.loss(out, y)
            self.optimizer(l)
            return out, l

    ft = FineTune(proto)
    ft.set_optimizer(opt.SGD(lr=0.05))
    ty = T(y, dev)
    ft.compile([tx], is_train=True, use_graph=False, sequential=False)
    losses = []
    for _ in range(8):
        _, l = ft(tx, ty)
        losses.append(float(tensor.to_numpy(l).reshape(-1)[0]))
    assert losses[-1] < losses[0]
    assert len(ft.get_params()) >= 6


def _run_graph(dev, nodes, inputs, outputs, inits=(), feeds=()):
    g = helper.make_graph(list(nodes), "g", list(inputs), list(outputs),
                          list(inits))
    m = onnx.load(helper.make_model(g).SerializeToString())
    autograd.training = False  # inference semantics (BN running stats)
    try:
        return sonnx.prepare(m, dev).run(list(feeds))
    finally:
        autograd.training = True


def test_import_attribute_inputs(dev):
    """Operators whose ONNX inputs are attributes here: Reshape, Slice, Pad,
    Clip, Tile, Expand, Gather, OneHot, ConstantOfShape, Constant folding."""
    x = RS.randn(2, 3, 4).astype(np.float32)
    vi = lambda n, s, t=TensorProto.FLOAT: helper.make_tensor_value_info(n, t, s)
    i64 = lambda n, v: numpy_helper.from_array(np.asarray(v, np.int64), n)
    f32 = lambda n, v: numpy_helper.from_array(np.asarray(v, np.float32), n)
    out = _run_graph(
        dev,
        [helper.make_node("Reshape", ["x", "shape"], ["r"]),
         helper.make_node("Slice", ["r", "st", "en", "ax"], ["s"]),
         helper.make_node("Clip", ["s", "lo", "hi"], ["c"]),
         helper.make_node("Tile", ["c", "rep"], ["y"])],
        [vi("x", [2, 3, 4])], [vi("y", None)],
        [i64("shape", [6, 4]), i64("st", [1]), i64("en", [5]), i64("ax", [0]),
         f32("lo", -0.3), f32("hi", 0.3), i64("rep", [1, 2])], [x])
    want = np.tile(np.clip(x.reshape(6, 4)[1:5], -0.3, 0.3), (1, 2))
    np.testing.assert_allclose(out[0], want, rtol=1e-6)
    # shape computed inside the graph feeds Expand through Constant folding
    cst = helper.make_node(
        "Constant", [], ["shp"],
        value=numpy_helper.from_array(np.array([2, 2, 3, 4], np.int64)))
    out = _run_graph(dev, [cst, helper.make_node("Expand", ["x", "shp"],
                                                 ["y"])],
                     [vi("x", [2, 3, 4])], [vi("y", None)], [], [x])
    np.testing.assert_allclose(out[0], np.broadcast_to(x, (2, 2, 3, 4)))
    out = _run_graph(
        dev, [helper.make_node("Gather", ["x", "idx"], ["y"], axis=2)],
        [vi("x", [2, 3, 4])], [vi("y", None)], [i64("idx", [3, 0])], [x])
    np.testing.assert_allclose(out[0], x[:, :, [3, 0]])
    out = _run_graph(
        dev, [helper.make_node("Pad", ["x", "pads", "cv"], ["y"],
                               mode="constant")],
        [vi("x", [2, 3, 4])], [vi("y", None)],
        [i64("pads", [0, 1, 0, 0, 0, 2]), f32("cv", 1.5)], [x])
    np.testing.assert_allclose(
        out[0], np.pad(x, ((0, 0), (1, 0), (0, 2)), constant_values=1.5))
    ids = np.array([0, 2, 1], np.int64)
    out = _run_graph(
        dev, [helper.make_node("OneHot", ["i", "d", "v"], ["y"], axis=-1)],
        [vi("i", [3], TensorProto.INT64)], [vi("y", None)],
        [i64("d", [3]), f32("v", [0, 1])], [ids])
    np.testing.assert_allclose(out[0], np.eye(3)[[0, 2, 1]])


def test_import_gemm_conv_bn_pool_nodes(dev):
    vi = lambda n, s: helper.make_tensor_value_info(n, TensorProto.FLOAT, s)
    f32 = lambda n, v: numpy_helper.from_array(np.asarray(v, np.float32), n)
    x = RS.randn(3, 5).astype(np.float32)
    W = RS.randn(4, 5).astype(np.float32)
    b = RS.randn(4).astype(np.float32)
    out = _run_graph(
        dev, [helper.make_node("Gemm", ["x", "W", "b"], ["y"], alpha=0.5,
                               beta=2.0, transB=1)],
        [vi("x", [3, 5])], [vi("y", None)], [f32("W", W), f32("b", b)], [x])
    np.testing.assert_allclose(out[0], 0.5 * x @ W.T + 2 * b, rtol=1e-4,
                               atol=1e-5)
    import torch
    import torch.nn.functional as F
    xi = RS.randn(2, 3, 9, 9).astype(np.float32)
    Wc = RS.randn(4, 3, 3, 3).astype(np.float32)
    sc, bi = RS.rand(4).astype(np.float32) + 0.5, RS.randn(4).astype(np.float32)
    mu, var = RS.randn(4).astype(np.float32), RS.rand(4).astype(np.float32) + 0.5
    out = _run_graph(
        dev,
        [helper.make_node("Conv", ["x", "W"], ["c"], kernel_shape=[3, 3],
                          pads=[1, 1, 1, 1], strides=[2, 2]),
         helper.make_node("BatchNormalization", ["c", "s", "b", "m", "v"],
                          ["n"], epsilon=1e-5),
         helper.make_node("MaxPool", ["n"], ["y"], kernel_shape=[2, 2],
                          strides=[2, 2])],
        [vi("x", [2, 3, 9, 9])], [vi("y", None)],
        [f32("W", Wc), f32("s", sc), f32("b", bi), f32("m", mu),
         f32("v", var)], [xi])
    t = F.conv2d(torch.from_numpy(xi), torch.from_numpy(Wc), None, 2, 1)
    t = F.batch_norm(t, torch.from_numpy(mu), torch.from_numpy(var),
                     torch.from_numpy(sc), torch.from_numpy(bi), False, 0.1,
                     1e-5)
    t = F.max_pool2d(t, 2, 2)
    np.testing.assert_allclose(out[0], t.numpy(), rtol=1e-4, atol=1e-4)


def test_run_node(dev):
    node = helper.make_node("LeakyRelu", ["x"], ["y"], alpha=0.2)
    x = RS.randn(3, 4).astype(np.float32)
    out = sonnx.run_node(node, [x], dev)
    np.testing.assert_allclose(tensor.to_numpy(out["y"]),
                               np.where(x > 0, x, 0.2 * x), rtol=1e-6)
    node = helper.make_node("Reshape", ["x", "s"], ["y"])
    out = sonnx.run_node(node, [x, np.array([4, 3], np.int64)], dev)
    assert out["y"].shape == (4, 3)


def test_export_fused_batchnorm_activation_round_trip():
    """relu(bn(x)) and relu(bn(x) + r) run as one operator; export writes the
    separate ONNX nodes and the re-imported graph gives the same outputs."""
    from singa_amd import autograd, layer, sonnx, tensor, device
    dev = device.get_default_device()
    autograd.training = False
    rng = np.random.RandomState(0)
    x = tensor.from_numpy(rng.randn(2, 4, 6, 6).astype(np.float32), dev)
    bn1 = layer.BatchNorm2d(activation="RELU")
    bn2 = layer.BatchNorm2d(activation="RELU")
    conv = layer.Conv2d(4, 3, padding=1)
    h = bn1(conv(x))
    y = bn2(conv(h), h)                    # with a residual
    model = sonnx.to_onnx([x], [y])
    kinds = [n.op_type for n in model.graph.node]
    assert kinds.count("BatchNormalization") == 2 and "Add" in kinds
    rep = sonnx.prepare(model, device=dev)
    out = rep.run([x])[0]
    if isinstance(out, tensor.Tensor):
        out = tensor.to_numpy(out)
    np.testing.assert_allclose(out, tensor.to_numpy(y), rtol=1e-5, atol=1e-5)


def _ln_gelu_chain_model(D=24, variant=0):
    """x -> exported layer-norm chain -> exported GELU chain (two spellings)."""
    from singa_amd.sonnx import helper, numpy_helper, TensorProto
    rng = np.random.RandomState(3)
    gamma = (rng.rand(D) + 0.5).astype(np.float32)
    beta = rng.randn(D).astype(np.float32)
    inits = [numpy_helper.from_array(gamma, "gamma"),
             numpy_helper.from_array(beta, "beta"),
             numpy_helper.from_array(np.array(2.0, np.float32), "two"),
             numpy_helper.from_array(np.array(1e-5, np.float32), "eps"),
             numpy_helper.from_array(np.array(2 ** 0.5, np.float32), "rt2"),
             numpy_helper.from_array(np.array(1.0, np.float32), "one"),
             numpy_helper.from_array(np.array(0.5, np.float32), "half")]
    n = helper.make_node
    nodes = [
        n("ReduceMean", ["x"], ["m"], axes=[-1], keepdims=1),
        n("Sub", ["x", "m"], ["d"]),
        n("Pow", ["d", "two"], ["p"]),
        n("ReduceMean", ["p"], ["v"], axes=[-1], keepdims=1),
        n("Add", ["v", "eps"], ["ve"]),
        n("Sqrt", ["ve"], ["s"]),
        n("Div", ["d", "s"], ["q"]),
        n("Mul", ["q", "gamma"], ["qg"]),
        n("Add", ["qg", "beta"], ["ln"]),
        n("Div", ["ln", "rt2"], ["t"]),
        n("Erf", ["t"], ["e"]),
        n("Add", ["e", "one"], ["e1"]),
    ]
    if variant == 0:
        nodes += [n("Mul", ["ln", "e1"], ["xe"]), n("Mul", ["xe", "half"], ["y"])]
    else:
        nodes += [n("Mul", ["ln", "half"], ["xh"]), n("Mul", ["xh", "e1"], ["y"])]
    g = helper.make_graph(
        nodes, "ln_gelu",
        [helper.make_tensor_value_info("x", TensorProto.FLOAT, [5, 7, D])],
        [helper.make_tensor_value_info("y", TensorProto.FLOAT, [5, 7, D])],
        inits)
    m = helper.make_model(g, opset_imports=[helper.make_operatorsetid("", 11)])
    return m, gamma, beta


@pytest.mark.parametrize("variant", [0, 1])
def test_import_fuses_layernorm_and_gelu_chains(dev, variant):
    from math import erf
    from singa_amd import sonnx
    m, gamma, beta = _ln_gelu_chain_model(variant=variant)
    x = np.random.RandomState(4).randn(5, 7, 24).astype(np.float32)
    mu = x.mean(-1, keepdims=True)
    ln = (x - mu) / np.sqrt(((x - mu) ** 2).mean(-1, keepdims=True) + 1e-5) \
        * gamma + beta
    want = 0.5 * ln * (1 + np.vectorize(erf)(ln / 2 ** 0.5))
    rep = sonnx.prepare(m, device=dev)
    kinds = [n.op_type for n, _ in rep.layers]
    assert kinds == ["LayerNormalization", "Gelu"], kinds
    np.testing.assert_allclose(rep.run([x])[0], want, rtol=1e-4, atol=1e-5)
    sonnx.SingaBackend.fuse_patterns = False
    try:
        plain = sonnx.prepare(m, device=dev)
        assert len(plain.layers) == 14
        np.testing.assert_allclose(plain.run([x])[0], want, rtol=1e-4,
                                   atol=1e-5)
    finally:
        sonnx.SingaBackend.fuse_patterns = True


def test_layernorm_chain_with_second_consumer_is_left_alone(dev):
    """The normalised difference feeding anything outside the chain blocks
    the rewrite."""
    from singa_amd import sonnx
    from singa_amd.sonnx import helper, TensorProto
    m, _, _ = _ln_gelu_chain_model()
    m.graph.node.extend([helper.make_node("Identity", ["q"], ["q_out"])])
    m.graph.output.extend(
        [helper.make_tensor_value_info("q_out", TensorProto.FLOAT, [5, 7, 24])])
    rep = sonnx.prepare(m, device=dev)
    kinds = [n.op_type for n, _ in rep.layers]
    assert "LayerNormalization" not in kinds and "Gelu" in kinds


def test_export_layernorm_gelu_round_trip(dev):
    from singa_amd import autograd, layer, sonnx, tensor
    autograd.training = False
    x = tensor.from_numpy(
        np.random.RandomState(5).randn(3, 16).astype(np.float32), dev)
    ln = layer.LayerNorm()
    y = autograd.gelu(ln(x))
    model = sonnx.to_onnx([x], [y])
    assert [n.op_type for n in model.graph.node] == ["LayerNormalization",
                                                     "Gelu"]
    out = sonnx.prepare(model, device=dev).run([x])[0]
    if isinstance(out, tensor.Tensor):
        out = tensor.to_numpy(out)
    np.testing.assert_allclose(out, tensor.to_numpy(y), rtol=1e-5, atol=1e-6)
