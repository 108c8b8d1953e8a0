"""Caffe models as singa_amd models: ``CaffeConverter`` reads a network
description (``.prototxt``), optionally a solver description and a trained
``.caffemodel``, and returns a ``model.Model`` whose layers are this package's
layers, an optimizer, and the trained parameters copied into the model.

The messages are the subset of the reference's src/proto/caffe.proto that
describes feed-forward vision networks (NetParameter, LayerParameter, the
per-layer parameter messages listed in ``_SCHEMA``, SolverParameter, BlobProto);
they are generated at import time with protobuf's descriptor API from the field
table below, so both the text and the binary encodings are those of Caffe.
Fields and messages outside the table (data sources, transformations, the V1
layer list) are skipped while parsing.

    conv = CaffeConverter("lenet.prototxt", "solver.prototxt",
                          input_sample_shape=(1, 28, 28),
                          param_path="lenet.caffemodel")
    net = conv.create_net()                 # a model.Model
    net.set_optimizer(conv.create_optimizer())
    net.compile([x], is_train=True, use_graph=True, sequential=False)
    conv.convert_params(net)                # trained blobs -> parameters
"""
import math
import re

import numpy as np
from google.protobuf import (descriptor_pb2, descriptor_pool, message_factory,
                             text_format)

from . import autograd, layer, model, opt, tensor
from .models.common import apply_update

_F = descriptor_pb2.FieldDescriptorProto
_SCALAR = {"int64": _F.TYPE_INT64, "int32": _F.TYPE_INT32,
           "uint32": _F.TYPE_UINT32, "float": _F.TYPE_FLOAT,
           "double": _F.TYPE_DOUBLE, "bool": _F.TYPE_BOOL,
           "string": _F.TYPE_STRING}

# enum -> (message it is declared in or None, value names in number order);
# the nesting keeps equal value names of different enums (MAX) apart
_ENUMS = {
    "Phase": (None, ["TRAIN", "TEST"]),
    "Engine": (None, ["DEFAULT", "CAFFE", "CUDNN"]),
    "PoolMethod": ("PoolingParameter", ["MAX", "AVE", "STOCHASTIC"]),
    "EltwiseOp": ("EltwiseParameter", ["PROD", "SUM", "MAX"]),
    "NormRegion": ("LRNParameter", ["ACROSS_CHANNELS", "WITHIN_CHANNEL"]),
    "SolverType": ("SolverParameter", ["SGD", "NESTEROV", "ADAGRAD",
                                       "RMSPROP", "ADADELTA", "ADAM"]),
}

# message -> [(name, number, type, label, default)]; label: 0 optional,
# 1 repeated, 2 repeated packed; a type that is neither a scalar nor an enum
# of _ENUMS names another message of this table
_SCHEMA = {
    "BlobShape": [("dim", 1, "int64", 2, None)],
    "BlobProto": [
        ("shape", 7, "BlobShape", 0, None), ("data", 5, "float", 2, None),
        ("diff", 6, "float", 2, None), ("double_data", 8, "double", 2, None),
        ("double_diff", 9, "double", 2, None), ("num", 1, "int32", 0, "0"),
        ("channels", 2, "int32", 0, "0"), ("height", 3, "int32", 0, "0"),
        ("width", 4, "int32", 0, "0")],
    "FillerParameter": [
        ("type", 1, "string", 0, "constant"), ("value", 2, "float", 0, "0"),
        ("min", 3, "float", 0, "0"), ("max", 4, "float", 0, "1"),
        ("mean", 5, "float", 0, "0"), ("std", 6, "float", 0, "1"),
        ("sparse", 7, "int32", 0, "-1")],
    "NetStateRule": [
        ("phase", 1, "Phase", 0, None), ("min_level", 2, "int32", 0, None),
        ("max_level", 3, "int32", 0, None), ("stage", 4, "string", 1, None),
        ("not_stage", 5, "string", 1, None)],
    "ParamSpec": [
        ("name", 1, "string", 0, None), ("lr_mult", 3, "float", 0, "1"),
        ("decay_mult", 4, "float", 0, "1")],
    "NetParameter": [
        ("name", 1, "string", 0, None), ("input", 3, "string", 1, None),
        ("input_shape", 8, "BlobShape", 1, None),
        ("input_dim", 4, "int32", 1, None),
        ("force_backward", 5, "bool", 0, "false"),
        ("debug_info", 7, "bool", 0, "false"),
        ("layer", 100, "LayerParameter", 1, None)],
    "SolverParameter": [
        ("net", 24, "string", 0, None),
        ("net_param", 25, "NetParameter", 0, None),
        ("train_net", 1, "string", 0, None),
        ("test_net", 2, "string", 1, None),
        ("test_iter", 3, "int32", 1, None),
        ("test_interval", 4, "int32", 0, "0"),
        ("base_lr", 5, "float", 0, None), ("display", 6, "int32", 0, None),
        ("max_iter", 7, "int32", 0, None), ("iter_size", 36, "int32", 0, "1"),
        ("lr_policy", 8, "string", 0, None), ("gamma", 9, "float", 0, None),
        ("power", 10, "float", 0, None), ("momentum", 11, "float", 0, None),
        ("weight_decay", 12, "float", 0, None),
        ("regularization_type", 29, "string", 0, "L2"),
        ("stepsize", 13, "int32", 0, None),
        ("stepvalue", 34, "int32", 1, None),
        ("clip_gradients", 35, "float", 0, "-1"),
        ("snapshot", 14, "int32", 0, "0"),
        ("snapshot_prefix", 15, "string", 0, None),
        ("device_id", 18, "int32", 0, "0"),
        ("random_seed", 20, "int64", 0, "-1"),
        ("type", 40, "string", 0, "SGD"), ("delta", 31, "float", 0, "1e-8"),
        ("momentum2", 39, "float", 0, "0.999"),
        ("rms_decay", 38, "float", 0, "0.99"),
        ("solver_type", 30, "SolverType", 0, "SGD")],
    "LayerParameter": [
        ("name", 1, "string", 0, None), ("type", 2, "string", 0, None),
        ("bottom", 3, "string", 1, None), ("top", 4, "string", 1, None),
        ("phase", 10, "Phase", 0, None), ("loss_weight", 5, "float", 1, None),
        ("param", 6, "ParamSpec", 1, None), ("blobs", 7, "BlobProto", 1, None),
        ("propagate_down", 11, "bool", 1, None),
        ("include", 8, "NetStateRule", 1, None),
        ("exclude", 9, "NetStateRule", 1, None),
        ("batch_norm_param", 139, "BatchNormParameter", 0, None),
        ("concat_param", 104, "ConcatParameter", 0, None),
        ("convolution_param", 106, "ConvolutionParameter", 0, None),
        ("dropout_param", 108, "DropoutParameter", 0, None),
        ("eltwise_param", 110, "EltwiseParameter", 0, None),
        ("elu_param", 140, "ELUParameter", 0, None),
        ("flatten_param", 135, "FlattenParameter", 0, None),
        ("inner_product_param", 117, "InnerProductParameter", 0, None),
        ("input_param", 143, "InputParameter", 0, None),
        ("lrn_param", 118, "LRNParameter", 0, None),
        ("pooling_param", 121, "PoolingParameter", 0, None),
        ("relu_param", 123, "ReLUParameter", 0, None),
        ("scale_param", 142, "ScaleParameter", 0, None),
        ("softmax_param", 125, "SoftmaxParameter", 0, None)],
    "BatchNormParameter": [
        ("use_global_stats", 1, "bool", 0, None),
        ("moving_average_fraction", 2, "float", 0, "0.999"),
        ("eps", 3, "float", 0, "1e-5")],
    "ConcatParameter": [("axis", 2, "int32", 0, "1"),
                        ("concat_dim", 1, "uint32", 0, "1")],
    "ConvolutionParameter": [
        ("num_output", 1, "uint32", 0, None),
        ("bias_term", 2, "bool", 0, "true"), ("pad", 3, "uint32", 1, None),
        ("kernel_size", 4, "uint32", 1, None),
        ("stride", 6, "uint32", 1, None), ("dilation", 18, "uint32", 1, None),
        ("pad_h", 9, "uint32", 0, "0"), ("pad_w", 10, "uint32", 0, "0"),
        ("kernel_h", 11, "uint32", 0, None),
        ("kernel_w", 12, "uint32", 0, None),
        ("stride_h", 13, "uint32", 0, None),
        ("stride_w", 14, "uint32", 0, None), ("group", 5, "uint32", 0, "1"),
        ("weight_filler", 7, "FillerParameter", 0, None),
        ("bias_filler", 8, "FillerParameter", 0, None),
        ("engine", 15, "Engine", 0, "DEFAULT"), ("axis", 16, "int32", 0, "1")],
    "DropoutParameter": [("dropout_ratio", 1, "float", 0, "0.5")],
    "EltwiseParameter": [("operation", 1, "EltwiseOp", 0, "SUM"),
                         ("coeff", 2, "float", 1, None)],
    "ELUParameter": [("alpha", 1, "float", 0, "1")],
    "FlattenParameter": [("axis", 1, "int32", 0, "1"),
                         ("end_axis", 2, "int32", 0, "-1")],
    "InnerProductParameter": [
        ("num_output", 1, "uint32", 0, None),
        ("bias_term", 2, "bool", 0, "true"),
        ("weight_filler", 3, "FillerParameter", 0, None),
        ("bias_filler", 4, "FillerParameter", 0, None),
        ("axis", 5, "int32", 0, "1"), ("transpose", 6, "bool", 0, "false")],
    "InputParameter": [("shape", 1, "BlobShape", 1, None)],
    "LRNParameter": [
        ("local_size", 1, "uint32", 0, "5"), ("alpha", 2, "float", 0, "1"),
        ("beta", 3, "float", 0, "0.75"),
        ("norm_region", 4, "NormRegion", 0, "ACROSS_CHANNELS"),
        ("k", 5, "float", 0, "1"), ("engine", 6, "Engine", 0, "DEFAULT")],
    "PoolingParameter": [
        ("pool", 1, "PoolMethod", 0, "MAX"), ("pad", 4, "uint32", 0, "0"),
        ("pad_h", 9, "uint32", 0, "0"), ("pad_w", 10, "uint32", 0, "0"),
        ("kernel_size", 2, "uint32", 0, None),
        ("kernel_h", 5, "uint32", 0, None), ("kernel_w", 6, "uint32", 0, None),
        ("stride", 3, "uint32", 0, "1"), ("stride_h", 7, "uint32", 0, None),
        ("stride_w", 8, "uint32", 0, None),
        ("engine", 11, "Engine", 0, "DEFAULT"),
        ("global_pooling", 12, "bool", 0, "false")],
    "ReLUParameter": [("negative_slope", 1, "float", 0, "0"),
                      ("engine", 2, "Engine", 0, "DEFAULT")],
    "ScaleParameter": [
        ("axis", 1, "int32", 0, "1"), ("num_axes", 2, "int32", 0, "1"),
        ("filler", 3, "FillerParameter", 0, None),
        ("bias_term", 4, "bool", 0, "false"),
        ("bias_filler", 5, "FillerParameter", 0, None)],
    "SoftmaxParameter": [("engine", 1, "Engine", 0, "DEFAULT"),
                         ("axis", 2, "int32", 0, "1")],
}


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "singa_amd_caffe.proto"
    fd.package = "caffe"
    fd.syntax = "proto2"
    msgs = {}
    for full in _SCHEMA:
        msgs[full] = fd.message_type.add()
        msgs[full].name = full
    scope = {}
    for name, (owner, values) in _ENUMS.items():
        e = (msgs[owner] if owner else fd).enum_type.add()
        e.name = name
        for number, value in enumerate(values):
            v = e.value.add()
            v.name, v.number = value, number
        scope[name] = ".caffe." + (owner + "." if owner else "") + name
    for full, fields in _SCHEMA.items():
        for (name, number, typ, label, default) in fields:
            f = msgs[full].field.add()
            f.name, f.number = name, number
            f.label = _F.LABEL_REPEATED if label else _F.LABEL_OPTIONAL
            if typ in _SCALAR:
                f.type = _SCALAR[typ]
                if label == 2:
                    f.options.packed = True
            elif typ in _ENUMS:
                f.type = _F.TYPE_ENUM
                f.type_name = scope[typ]
            else:
                f.type = _F.TYPE_MESSAGE
                f.type_name = ".caffe." + typ
            if default is not None:
                f.default_value = default
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {full: message_factory.GetMessageClass(
        pool.FindMessageTypeByName("caffe." + full)) for full in _SCHEMA}


_CLS = _build()
NetParameter = _CLS["NetParameter"]
SolverParameter = _CLS["SolverParameter"]
LayerParameter = _CLS["LayerParameter"]
BlobProto = _CLS["BlobProto"]

def parse_text(text, cls=NetParameter):
    """A message of this module's schema from Caffe's text format."""
    msg = cls()
    text_format.Parse(text, msg, allow_unknown_field=True)
    return msg


def blob_array(blob):
    """numpy array of a BlobProto (shape from ``shape`` or, for files of
    older Caffe versions, from num / channels / height / width)."""
    values = blob.data if len(blob.data) else blob.double_data
    a = np.asarray(values, dtype=np.float32)
    if blob.HasField("shape"):
        return a.reshape([int(d) for d in blob.shape.dim])
    legacy = [blob.num, blob.channels, blob.height, blob.width]
    if a.size == int(np.prod([max(d, 1) for d in legacy])) and any(legacy):
        return a.reshape([max(d, 1) for d in legacy])
    return a


def array_blob(a):
    """BlobProto holding ``a``."""
    blob = BlobProto()
    a = np.asarray(a, dtype=np.float32)
    blob.shape.dim.extend(a.shape)
    blob.data.extend(a.ravel().tolist())
    return blob


# ---------------------------------------------------------------------------
# layers without a counterpart in layer.py
# ---------------------------------------------------------------------------
class ChannelScale(layer.Layer):
    """Caffe's Scale layer along the channel axis: y = x * gamma [+ beta]."""

    def __init__(self, bias=False):
        super().__init__()
        self.bias = bias

    def initialize(self, x):
        shape = (1, x.shape[1]) + (1,) * (len(x.shape) - 2)
        self._new_param('gamma', shape, x, lambda t: t.set_value(1.0))
        if self.bias:
            self._new_param('beta', shape, x, lambda t: t.set_value(0.0))

    def forward(self, x):
        y = autograd.mul(x, self.gamma)
        return autograd.add(y, self.beta) if self.bias else y


class ChannelLRN(layer.Layer):
    """Local response normalisation across channels (Caffe's default region):
    y = x / (k + alpha / n * sum of x^2 over the n nearest channels)^beta,
    the sum taken as an average pooling over the channel axis."""

    def __init__(self, local_size=5, alpha=1.0, beta=0.75, k=1.0):
        super().__init__()
        assert local_size % 2 == 1, "local_size must be odd"
        self.local_size, self.alpha, self.beta, self.k = \
            local_size, alpha, beta, k
        self.window = layer.AvgPool2d((local_size, 1), (1, 1), 0)

    def forward(self, x):
        n, c, h, w = x.shape
        sq = autograd.reshape(autograd.mul(x, x), (n, 1, c, h * w))
        # the pooling layer leaves padding out of its divisor: the channels
        # beyond either end are added as zeros that count
        edge = tensor.Tensor((n, 1, self.local_size // 2, h * w), x.device,
                             x.dtype)
        edge.set_value(0.0)
        sq = autograd.cat([edge, sq, edge], 2)
        mean = autograd.reshape(self.window(sq), (n, c, h, w))
        scale = tensor.Tensor((1,), x.device, x.dtype)
        scale.set_value(self.alpha)      # mean * n * alpha / n
        shift = tensor.Tensor((1,), x.device, x.dtype)
        shift.set_value(self.k)
        expo = tensor.Tensor((1,), x.device, x.dtype)
        expo.set_value(-self.beta)
        base = autograd.add(autograd.mul(mean, scale), shift)
        return autograd.mul(x, autograd.pow(base, expo))


# ---------------------------------------------------------------------------
# the network
# ---------------------------------------------------------------------------
_DATA_TYPES = ("Data", "ImageData", "Input", "DummyData", "MemoryData",
               "HDF5Data", "WindowData", "AnnotatedData")
_LOSS_TYPES = ("SoftmaxWithLoss", "EuclideanLoss", "SigmoidCrossEntropyLoss")
_SKIPPED = ("Accuracy", "Silence") + _LOSS_TYPES


def _attr_name(name):
    name = re.sub(r"\W", "_", name)
    return name if not name[:1].isdigit() else "l_" + name


def _ceil_extra(size, kernel, stride, pad):
    """Rows to add at the far edge so that this package's floor rule gives the
    number of windows Caffe's pooling (ceil rule) gives."""
    out = int(math.ceil((size + 2 * pad - kernel) / float(stride))) + 1
    if pad > 0 and (out - 1) * stride >= size + pad:
        out -= 1
    return max((out - 1) * stride + kernel - (size + 2 * pad), 0)


class CaffeNet(model.Model):
    """The layers of a Caffe NetParameter for one phase, executed in file
    order; blobs are looked up by name, so in-place layers (top == bottom)
    and branches (Eltwise, Concat) need nothing special.  Loss and accuracy
    layers are not executed by ``forward``: ``train_one_batch`` applies the
    loss the file names to the blob the loss layer reads."""

    def __init__(self, net, phase="train", input_sample_shape=None):
        super(CaffeNet, self).__init__()
        self.phase = phase
        self.input_sample_shape = input_sample_shape
        self.steps = []             # (kind, attribute or None, bottoms, tops, conf)
        self.caffe_layers = {}      # attribute name -> (caffe layer name, type)
        self.input_blob = net.input[0] if len(net.input) else None
        self.loss_type, self.loss_bottom = None, None
        self.softmax_cross_entropy = layer.SoftMaxCrossEntropy()
        self.mse = layer.MeanSquareError()
        self.bce = layer.BinaryCrossEntropy()
        last_bn = {}                # blob name -> BatchNorm attribute that wrote it
        want = 0 if phase == "train" else 1
        for lp in net.layer:
            if not self._in_phase(lp, want):
                continue
            t = lp.type
            if t in _DATA_TYPES:
                if self.input_blob is None:
                    self.input_blob = lp.top[0]
                continue
            if t in _LOSS_TYPES:
                self.loss_type, self.loss_bottom = t, lp.bottom[0]
                continue
            if t in _SKIPPED:
                continue
            attr = _attr_name(lp.name)
            bottoms, tops = list(lp.bottom), list(lp.top)
            if t == "Scale" and bottoms[0] in last_bn and \
                    tops[0] == bottoms[0]:
                # BatchNorm + in-place Scale is this package's BatchNorm2d
                # with its affine parameters
                self.caffe_layers[last_bn.pop(bottoms[0])].append(lp.name)
                continue
            made = self._make(lp)
            if made is None:
                raise NotImplementedError(
                    "Caffe layer type %r (layer %r)" % (t, lp.name))
            setattr(self, attr, made)
            self.caffe_layers[attr] = [lp.name]
            self.steps.append((t, attr, bottoms, tops, lp))
            for top in tops:
                last_bn.pop(top, None)
            if t == "BatchNorm":
                last_bn[tops[0]] = attr
        if self.input_blob is None:
            raise ValueError("the network names no input blob")
        self.output_blob = self.steps[-1][3][0]

    @staticmethod
    def _in_phase(lp, want):
        if lp.HasField("phase") and lp.phase != want:
            return False
        if len(lp.include) and not any(
                (not r.HasField("phase")) or r.phase == want
                for r in lp.include):
            return False
        if any(r.HasField("phase") and r.phase == want for r in lp.exclude):
            return False
        return True

    @staticmethod
    def _make(lp):
        t = lp.type
        if t == "Convolution":
            c = lp.convolution_param
            if any(d != 1 for d in c.dilation):
                raise NotImplementedError("dilated convolution (%s)" % lp.name)

            def hw(rep, h, w, default):
                if getattr(c, h) or getattr(c, w):
                    return (int(getattr(c, h)), int(getattr(c, w)))
                if len(rep) == 0:
                    return (default, default)
                return (int(rep[0]), int(rep[-1]))
            return layer.Conv2d(
                int(c.num_output),
                hw(c.kernel_size, "kernel_h", "kernel_w", 1),
                stride=hw(c.stride, "stride_h", "stride_w", 1),
                padding=hw(c.pad, "pad_h", "pad_w", 0),
                group=int(c.group), bias=bool(c.bias_term))
        if t == "Pooling":
            return _CaffePooling(lp.pooling_param)
        if t == "InnerProduct":
            p = lp.inner_product_param
            return _InnerProduct(int(p.num_output), bool(p.bias_term),
                                 bool(p.transpose))
        if t == "ReLU":
            slope = lp.relu_param.negative_slope
            return _Lambda(autograd.relu) if slope == 0 else \
                _Lambda(lambda x: autograd.leakyrelu(x, slope))
        if t == "ELU":
            alpha = lp.elu_param.alpha
            return _Lambda(lambda x: autograd.elu(x, alpha))
        if t == "Sigmoid":
            return _Lambda(autograd.sigmoid)
        if t == "TanH":
            return _Lambda(autograd.tanh)
        if t == "Dropout":
            ratio = lp.dropout_param.dropout_ratio
            return _Lambda(lambda x: autograd.dropout(x, ratio=ratio))
        if t == "Softmax":
            axis = lp.softmax_param.axis
            return _Lambda(lambda x: autograd.softmax(x, axis))
        if t == "Flatten":
            return _Lambda(lambda x: autograd.flatten(
                x, lp.flatten_param.axis))
        if t == "LRN":
            p = lp.lrn_param
            if p.norm_region != 0:
                raise NotImplementedError("LRN within channels (%s)" % lp.name)
            return ChannelLRN(int(p.local_size), p.alpha, p.beta, p.k)
        if t == "BatchNorm":
            p = lp.batch_norm_param
            return layer.BatchNorm2d(
                momentum=1.0 - p.moving_average_fraction)
        if t == "Scale":
            return ChannelScale(bool(lp.scale_param.bias_term))
        if t == "Eltwise":
            p = lp.eltwise_param
            return _Eltwise(int(p.operation), list(p.coeff))
        if t == "Concat":
            p = lp.concat_param
            axis = p.axis if p.HasField("axis") or \
                not p.HasField("concat_dim") else int(p.concat_dim)
            return _Lambda(lambda *xs: autograd.cat(list(xs), axis))
        if t == "Split":
            return _Lambda(lambda x: x)
        return None

    def forward(self, x):
        blobs = {self.input_blob: x}
        for (t, attr, bottoms, tops, _) in self.steps:
            y = getattr(self, attr)(*[blobs[b] for b in bottoms])
            for top in tops:             # Split: every top is the same tensor
                blobs[top] = y
        want = self.loss_bottom if self.loss_bottom in blobs \
            else self.output_blob
        return blobs[want]

    def loss(self, out, y):
        if self.loss_type == "EuclideanLoss":
            return self.mse(out, y)
        if self.loss_type == "SigmoidCrossEntropyLoss":
            return self.bce(autograd.sigmoid(out), y)
        return self.softmax_cross_entropy(out, y)

    def train_one_batch(self, x, y, dist_option='plain', spars=None):
        out = self.forward(x)
        loss = self.loss(out, y)
        apply_update(self.optimizer, loss, dist_option, spars)
        return out, loss

class _Lambda(layer.Layer):
    """A parameter-free layer around a function of tensors."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, *xs):
        return self.fn(*xs)


class _InnerProduct(layer.Layer):
    """Caffe's InnerProduct: the input is flattened from axis 1."""

    def __init__(self, num_output, bias, transpose):
        super().__init__()
        self.transpose = transpose      # the blob is stored (in, out) already
        self.fc = layer.Linear(num_output, bias=bias)

    def forward(self, x):
        if len(x.shape) > 2:
            x = autograd.flatten(x, 1)
        return self.fc(x)


class _CaffePooling(layer.Layer):
    """Pooling with Caffe's output size: the number of windows is rounded up,
    which this layer obtains by padding the far edges (max pooling pads with
    the lowest value; average pooling divides by the full window, where Caffe
    divides by the part of the window inside the padded image)."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def initialize(self, x):
        p = self.p
        h, w = x.shape[2], x.shape[3]
        if p.global_pooling:
            kernel, stride, pad = (h, w), (1, 1), (0, 0)
        else:
            kernel = (p.kernel_h, p.kernel_w) if p.HasField("kernel_h") \
                else (p.kernel_size, p.kernel_size)
            stride = (p.stride_h, p.stride_w) if p.HasField("stride_h") \
                else (p.stride, p.stride)
            pad = (p.pad_h, p.pad_w) if (p.pad_h or p.pad_w) \
                else (p.pad, p.pad)
        if p.pool == 2:
            raise NotImplementedError("stochastic pooling")
        eh = _ceil_extra(h, kernel[0], stride[0], pad[0])
        ew = _ceil_extra(w, kernel[1], stride[1], pad[1])
        padding = (pad[0], pad[0] + eh, pad[1], pad[1] + ew) \
            if (eh or ew) else tuple(pad)
        cls = layer.MaxPool2d if p.pool == 0 else layer.AvgPool2d
        self.pool = cls(tuple(kernel), tuple(stride), padding)

    def forward(self, x):
        return self.pool(x)


class _Eltwise(layer.Layer):

    def __init__(self, operation, coeff):
        super().__init__()
        self.operation, self.coeff = operation, coeff

    def forward(self, *xs):
        if self.operation == 0:
            y = xs[0]
            for x in xs[1:]:
                y = autograd.mul(y, x)
            return y
        if self.operation == 2:
            return autograd.max(*xs)
        terms = list(xs)
        if self.coeff and any(c != 1 for c in self.coeff):
            scaled = []
            for x, c in zip(terms, self.coeff):
                f = tensor.Tensor((1,), x.device, x.dtype)
                f.set_value(c)
                scaled.append(autograd.mul(x, f))
            terms = scaled
        y = terms[0]
        for x in terms[1:]:
            y = autograd.add(y, x)
        return y


# ---------------------------------------------------------------------------
# learning-rate policies of the solver that opt.py has no schedule for
# ---------------------------------------------------------------------------
class InverseDecay(opt.DecayScheduler):
    """Caffe's "inv" policy: base_lr * (1 + gamma * step) ^ (-power)."""

    def __init__(self, init_value, gamma, power):
        super(InverseDecay, self).__init__(init_value)
        self.gamma, self.power = gamma, power

    def call(self, step):
        expo = tensor.Tensor((1,), step.device)
        expo.set_value(-self.power)
        return self.init_value * tensor.pow(step * self.gamma + 1.0, expo)


class PolynomialDecay(opt.DecayScheduler):
    """Caffe's "poly" policy: base_lr * (1 - step / max_iter) ^ power."""

    def __init__(self, init_value, max_iter, power):
        super(PolynomialDecay, self).__init__(init_value)
        self.max_iter, self.power = max_iter, power

    def call(self, step):
        expo = tensor.Tensor((1,), step.device)
        expo.set_value(self.power)
        left = tensor.relu(step * (-1.0 / self.max_iter) + 1.0)
        return self.init_value * tensor.pow(left, expo)


# ---------------------------------------------------------------------------
# the converter
# ---------------------------------------------------------------------------
class CaffeConverter(object):
    """``net_proto`` / ``solver_proto``: paths of text-format files;
    ``param_path``: a binary NetParameter with trained blobs (.caffemodel);
    ``input_sample_shape``: (C, H, W) of one sample when the network file
    does not declare its input."""

    def __init__(self, net_proto, solver_proto=None, input_sample_shape=None,
                 param_path=None):
        self.net_proto = net_proto
        self.solver_proto = solver_proto
        self.input_sample_shape = input_sample_shape
        self.param_path = param_path

    def read_net_proto(self):
        with open(self.net_proto) as f:
            return parse_text(f.read(), NetParameter)

    def read_solver_proto(self):
        with open(self.solver_proto) as f:
            return parse_text(f.read(), SolverParameter)

    def read_caffemodel(self):
        net = NetParameter()
        with open(self.param_path, "rb") as f:
            net.ParseFromString(f.read())
        return net

    def create_net(self, phase="train"):
        net = self.read_net_proto()
        shape = self.input_sample_shape
        if shape is None:
            if len(net.input_shape):
                shape = tuple(int(d) for d in net.input_shape[0].dim)[1:]
            elif len(net.input_dim) >= 4:
                shape = tuple(net.input_dim[1:4])
            else:
                for lp in net.layer:
                    if lp.type == "Input" and len(lp.input_param.shape):
                        shape = tuple(
                            int(d) for d in lp.input_param.shape[0].dim)[1:]
                        break
        return CaffeNet(net, phase, shape)

    def create_optimizer(self):
        s = self.read_solver_proto()
        lr = self._schedule(s)
        kind = s.type.upper() if s.HasField("type") else \
            _ENUMS["SolverType"][1][s.solver_type]
        decay = s.weight_decay if s.HasField("weight_decay") else 0.0
        momentum = s.momentum if s.HasField("momentum") else 0.0
        if kind in ("SGD", "NESTEROV"):
            return opt.SGD(lr=lr, momentum=momentum, weight_decay=decay,
                           nesterov=(kind == "NESTEROV" and momentum > 0))
        if kind == "ADAGRAD":
            return opt.AdaGrad(lr=lr, epsilon=s.delta, weight_decay=decay)
        if kind == "RMSPROP":
            return opt.RMSProp(lr=lr, rho=s.rms_decay, epsilon=s.delta,
                               weight_decay=decay)
        if kind == "ADAM":
            return opt.Adam(lr=lr, beta_1=momentum or 0.9, beta_2=s.momentum2,
                            epsilon=s.delta, weight_decay=decay)
        raise NotImplementedError("solver type %s" % kind)

    @staticmethod
    def _schedule(s):
        policy = s.lr_policy if s.HasField("lr_policy") else "fixed"
        if policy == "fixed":
            return s.base_lr
        if policy == "step":
            return opt.ExponentialDecay(s.base_lr, s.stepsize, s.gamma, True)
        if policy == "exp":
            return opt.ExponentialDecay(s.base_lr, 1, s.gamma, False)
        if policy == "inv":
            return InverseDecay(s.base_lr, s.gamma, s.power)
        if policy == "poly":
            return PolynomialDecay(s.base_lr, s.max_iter, s.power)
        raise NotImplementedError("lr_policy %r" % policy)

    def convert_params(self, net, trained=None):
        """Copy the blobs of the .caffemodel (or of ``trained``, a
        NetParameter) into the parameters of ``net`` (a compiled CaffeNet);
        returns the names of the Caffe layers that were used."""
        trained = trained if trained is not None else self.read_caffemodel()
        blobs = {lp.name: [blob_array(b) for b in lp.blobs]
                 for lp in trained.layer if len(lp.blobs)}
        used = []
        for attr, names in net.caffe_layers.items():
            obj = getattr(net, attr)
            given = blobs.get(names[0])
            if given is None:
                continue
            used.append(names[0])
            if isinstance(obj, layer.Conv2d):
                self._set(obj.W, given[0])
                if obj.bias and len(given) > 1:
                    self._set(obj.b, given[1])
            elif isinstance(obj, _InnerProduct):
                w = given[0].reshape(given[0].shape[-2:])
                self._set(obj.fc.W, w if obj.transpose else w.T)
                if len(given) > 1:
                    self._set(obj.fc.b, given[1])
            elif isinstance(obj, layer.BatchNorm2d):
                factor = float(given[2].ravel()[0]) if len(given) > 2 else 1.0
                factor = 1.0 / factor if factor != 0 else 0.0
                self._set(obj.running_mean, given[0] * factor)
                self._set(obj.running_var, given[1] * factor)
                if len(names) > 1 and names[1] in blobs:
                    affine = blobs[names[1]]
                    used.append(names[1])
                    self._set(obj.scale, affine[0])
                    if len(affine) > 1:
                        self._set(obj.bias, affine[1])
            elif isinstance(obj, ChannelScale):
                self._set(obj.gamma, given[0])
                if obj.bias and len(given) > 1:
                    self._set(obj.beta, given[1])
        return used

    @staticmethod
    def _set(param, array):
        array = np.ascontiguousarray(array, dtype=np.float32).reshape(
            param.shape)
        param.copy_from_numpy(array)


def export_caffemodel(net, path):
    """Write the parameters of a compiled CaffeNet as a binary NetParameter
    (one LayerParameter with blobs per Caffe layer), readable by Caffe and by
    ``CaffeConverter(param_path=...)``."""
    out = NetParameter()
    for attr, names in net.caffe_layers.items():
        obj = getattr(net, attr)
        arrays, extra = [], []
        if isinstance(obj, layer.Conv2d):
            arrays = [obj.W] + ([obj.b] if obj.bias else [])
        elif isinstance(obj, _InnerProduct):
            w = tensor.to_numpy(obj.fc.W)
            arrays = [w if obj.transpose else w.T] + \
                ([obj.fc.b] if obj.fc.bias else [])
        elif isinstance(obj, layer.BatchNorm2d):
            arrays = [obj.running_mean, obj.running_var,
                      np.ones((1,), np.float32)]
            if len(names) > 1:
                extra = [obj.scale, obj.bias]
        elif isinstance(obj, ChannelScale):
            arrays = [tensor.to_numpy(obj.gamma).ravel()] + \
                ([tensor.to_numpy(obj.beta).ravel()] if obj.bias else [])
        for name, group in ((names[0], arrays),
                            (names[1] if len(names) > 1 else None, extra)):
            if not group:
                continue
            lp = out.layer.add()
            lp.name = name
            for a in group:
                a = a if isinstance(a, np.ndarray) else tensor.to_numpy(a)
                lp.blobs.append(array_blob(a))
    with open(path, "wb") as f:
        f.write(out.SerializeToString())
