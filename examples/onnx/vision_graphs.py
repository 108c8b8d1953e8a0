"""Building blocks for the vision networks of the ONNX model zoo, written as
ONNX graphs with random weights, plus an independent evaluator of such graphs.

The per-model scripts next to this file (resnet18.py, vgg16.py, ...) describe
their architecture with ``GraphBuilder`` -- the same operator vocabulary and
layout the zoo files use (Conv + BatchNormalization + Relu stems, grouped and
depthwise convolutions, channel shuffle as Reshape-Transpose-Reshape, Concat
growth, PRelu, LeakyRelu, pixel shuffle ...) -- so every script runs end to end
on a machine without the downloaded files, at full size on a GPU and at a
reduced width in the tests.

``evaluate(model, feeds)`` executes a graph node by node with numpy and
float64 PyTorch CPU convolutions / poolings: the oracle the importer's output
is compared against.  It knows nothing of the importer.

Workloads: reference examples/onnx/{resnet18,vgg16,vgg19,mobilenet,
densenet121,shufflenetv1,shufflenetv2,squeezenet,arcface,fer_emotion,
tiny_yolov2,superresolution}.py (downloaded zoo files there).
"""
import numpy as np

import utils  # noqa: F401  (sets sys.path)
from singa_amd import onnx_proto as onnx
from singa_amd.onnx_proto import helper, numpy_helper

F, I64 = onnx.DataType.FLOAT, onnx.DataType.INT64


def scaled(channels, width, divisor=8):
    """Channel count at a width multiplier, kept a multiple of ``divisor``."""
    return max(divisor, int(channels * width + divisor / 2) // divisor * divisor)


class GraphBuilder:
    """Accumulates nodes and initializers; every method returns the name of
    the value it produced."""

    def __init__(self, name, seed=0):
        self.name = name
        self.rng = np.random.RandomState(seed)
        self.nodes, self.weights = [], {}
        self.uid = 0
        self.inputs, self.outputs = [], []

    # ---- values ----------------------------------------------------------
    def input(self, name, shape, elem=F):
        self.inputs.append(helper.make_tensor_value_info(name, elem,
                                                         list(shape)))
        return name

    def output(self, value, shape, name="output", elem=F):
        self.nodes.append(helper.make_node("Identity", [value], [name],
                                           name="out_" + name))
        self.outputs.append(helper.make_tensor_value_info(name, elem,
                                                          list(shape)))
        return name

    def const(self, arr, hint="c"):
        self.uid += 1
        name = "%s_%d" % (hint, self.uid)
        self.weights[name] = np.asarray(arr)
        return name

    def weight(self, shape, fan_in=None, hint="w"):
        fan_in = fan_in or int(np.prod(shape[1:])) or 1
        return self.const((self.rng.randn(*shape) *
                           np.sqrt(2.0 / fan_in)).astype(np.float32), hint)

    def op(self, kind, ins, n_out=1, **attrs):
        self.uid += 1
        outs = ["%s_%d%s" % (kind.lower(), self.uid, "" if n_out == 1
                             else "_%d" % i) for i in range(n_out)]
        self.nodes.append(helper.make_node(kind, list(ins), outs,
                                           name=outs[0], **attrs))
        return outs[0] if n_out == 1 else outs

    # ---- layers ----------------------------------------------------------
    def conv(self, x, cin, cout, k, stride=1, pad=None, group=1, bias=True):
        pad = k // 2 if pad is None else pad
        ins = [x, self.weight((cout, cin // group, k, k))]
        if bias:
            ins.append(self.const((self.rng.randn(cout) * 0.05)
                                  .astype(np.float32), "b"))
        return self.op("Conv", ins, kernel_shape=[k, k],
                       strides=[stride, stride], pads=[pad] * 4, group=group,
                       dilations=[1, 1])

    def bn(self, x, c, eps=1e-5):
        r = self.rng
        return self.op("BatchNormalization", [
            x, self.const((1 + 0.1 * r.randn(c)).astype(np.float32), "g"),
            self.const((0.1 * r.randn(c)).astype(np.float32), "beta"),
            self.const((0.1 * r.randn(c)).astype(np.float32), "mean"),
            self.const((1 + 0.1 * r.rand(c)).astype(np.float32), "var")],
            epsilon=eps, momentum=0.9)

    def relu(self, x):
        return self.op("Relu", [x])

    def conv_bn_relu(self, x, cin, cout, k, stride=1, pad=None, group=1,
                     act=True):
        y = self.bn(self.conv(x, cin, cout, k, stride, pad, group,
                              bias=False), cout)
        return self.relu(y) if act else y

    def maxpool(self, x, k, stride, pad=0, ceil=False):
        return self.op("MaxPool", [x], kernel_shape=[k, k],
                       strides=[stride, stride], pads=[pad] * 4)

    def avgpool(self, x, k, stride, pad=0):
        return self.op("AveragePool", [x], kernel_shape=[k, k],
                       strides=[stride, stride], pads=[pad] * 4)

    def global_pool(self, x):
        return self.op("GlobalAveragePool", [x])

    def flatten(self, x):
        return self.op("Flatten", [x], axis=1)

    def reshape(self, x, shape):
        return self.op("Reshape", [x, self.const(np.array(shape, np.int64),
                                                 "shape")])

    def dense(self, x, cin, cout):
        """Gemm with the (out, in) weight layout of the zoo files."""
        return self.op("Gemm", [x, self.weight((cout, cin)),
                                self.const((self.rng.randn(cout) * 0.05)
                                           .astype(np.float32), "b")],
                       alpha=1.0, beta=1.0, transA=0, transB=1)

    def channel_shuffle(self, x, c, groups, hw):
        y = self.reshape(x, [-1, groups, c // groups, hw, hw])
        y = self.op("Transpose", [y], perm=[0, 2, 1, 3, 4])
        return self.reshape(y, [-1, c, hw, hw])

    # ---- result ----------------------------------------------------------
    def model(self):
        graph = helper.make_graph(
            self.nodes, self.name, self.inputs, self.outputs,
            initializer=[numpy_helper.from_array(v, k)
                         for k, v in self.weights.items()])
        return helper.make_model(graph)


# ---------------------------------------------------------------- evaluator
def _attrs(node):
    out = {}
    for a in node.attribute:
        v = helper.get_attribute_value(a)
        out[a.name] = v.decode() if isinstance(v, bytes) else v
    return out


def _torch64(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float64))


def _pads2d(x, pads, value):
    import torch.nn.functional as TF
    t, l, b, r = pads
    return TF.pad(x, (l, r, t, b), value=value)


def evaluate(model, feeds):
    """Run ``model`` on the dict ``feeds`` (name -> ndarray); float tensors are
    carried in float64.  Returns the list of graph outputs."""
    import torch
    import torch.nn.functional as TF
    env = {}
    for t in model.graph.initializer:
        a = numpy_helper.to_array(t)
        env[t.name] = a.astype(np.float64) if a.dtype.kind == "f" else a
    for k, v in feeds.items():
        env[k] = v.astype(np.float64) if v.dtype.kind == "f" else v
    for n in model.graph.node:
        a = _attrs(n)
        x = [env[i] for i in n.input if i]
        k = n.op_type
        if k == "Conv":
            xp = _pads2d(_torch64(x[0]), a.get("pads", [0] * 4), 0.0)
            y = TF.conv2d(xp, _torch64(x[1]),
                          _torch64(x[2]) if len(x) > 2 else None,
                          stride=tuple(a.get("strides", [1, 1])),
                          groups=a.get("group", 1)).numpy()
        elif k == "BatchNormalization":
            sh = (1, -1) + (1,) * (x[0].ndim - 2)
            y = (x[0] - x[3].reshape(sh)) / np.sqrt(
                x[4].reshape(sh) + a.get("epsilon", 1e-5)) * \
                x[1].reshape(sh) + x[2].reshape(sh)
        elif k == "Relu":
            y = np.maximum(x[0], 0)
        elif k == "LeakyRelu":
            y = np.where(x[0] > 0, x[0], a.get("alpha", 0.01) * x[0])
        elif k == "PRelu":
            y = np.where(x[0] > 0, x[0], x[1] * x[0])
        elif k == "Clip":
            y = np.clip(x[0], x[1] if len(x) > 1 else a.get("min"),
                        x[2] if len(x) > 2 else a.get("max"))
        elif k == "Sigmoid":
            y = 1 / (1 + np.exp(-x[0]))
        elif k == "Tanh":
            y = np.tanh(x[0])
        elif k == "MaxPool":
            xp = _pads2d(_torch64(x[0]), a.get("pads", [0] * 4), -np.inf)
            y = TF.max_pool2d(xp, tuple(a["kernel_shape"]),
                              tuple(a.get("strides", [1, 1]))).numpy()
        elif k == "AveragePool":
            # count_include_pad = 0: divide by the number of valid cells
            pads = a.get("pads", [0] * 4)
            ks, st = tuple(a["kernel_shape"]), tuple(a.get("strides", [1, 1]))
            xs = TF.avg_pool2d(_pads2d(_torch64(x[0]), pads, 0.0), ks, st)
            ones = TF.avg_pool2d(_pads2d(torch.ones_like(_torch64(x[0])),
                                         pads, 0.0), ks, st)
            if a.get("count_include_pad", 0):
                y = xs.numpy()
            else:
                y = (xs / ones).numpy()
        elif k == "GlobalAveragePool":
            y = x[0].mean(axis=(2, 3), keepdims=True)
        elif k in ("Add", "Sub", "Mul", "Div"):
            y = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply,
                 "Div": np.divide}[k](x[0], x[1])
        elif k == "Sum":
            y = sum(x[1:], x[0])
        elif k == "Concat":
            y = np.concatenate(x, axis=a["axis"])
        elif k == "Split":
            parts = a.get("split")
            if parts is None and len(x) > 1:
                parts = [int(v) for v in x[1]]
            ax = a.get("axis", 0)
            if parts is None:
                ys = np.split(x[0], len(n.output), axis=ax)
            else:
                ys = np.split(x[0], np.cumsum(parts)[:-1], axis=ax)
            for name, v in zip(n.output, ys):
                env[name] = v
            continue
        elif k == "Reshape":
            shape = [x[0].shape[i] if int(d) == 0 else int(d)
                     for i, d in enumerate(x[1])]
            y = x[0].reshape(shape)
        elif k == "Transpose":
            y = np.transpose(x[0], a.get("perm"))
        elif k == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)
        elif k == "Gemm":
            A = x[0].T if a.get("transA", 0) else x[0]
            B = x[1].T if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ B)
            if len(x) > 2:
                y = y + a.get("beta", 1.0) * x[2]
        elif k == "MatMul":
            y = x[0] @ x[1]
        elif k == "Softmax":
            ax = a.get("axis", 1)
            e = np.exp(x[0] - x[0].max(axis=ax, keepdims=True))
            y = e / e.sum(axis=ax, keepdims=True)
        elif k in ("Dropout", "Identity"):
            y = x[0]
        elif k == "DepthToSpace":
            b = a["blocksize"]
            N, C, H, W = x[0].shape
            if a.get("mode", "DCR") == "DCR":
                t = x[0].reshape(N, b, b, C // (b * b), H, W)
                t = t.transpose(0, 3, 4, 1, 5, 2)
            else:
                t = x[0].reshape(N, C // (b * b), b, b, H, W)
                t = t.transpose(0, 1, 4, 2, 5, 3)
            y = t.reshape(N, C // (b * b), H * b, W * b)
        else:
            raise NotImplementedError("evaluator: operator " + k)
        env[n.output[0]] = y
    return [np.asarray(env[o.name]) for o in model.graph.output]
