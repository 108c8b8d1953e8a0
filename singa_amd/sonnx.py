"""ONNX import (``prepare`` -> ``SingaRep`` / ``SONNXModel``) and export
(``to_onnx``), opset 11 / IR version 6.

Functionally the reference's python/singa/sonnx.py (backend :1037-1948,
``SingaRep`` :1951-2193, ``SONNXModel`` :2196-2220, frontend :60-987) with the
``onnx`` package replaced by :mod:`singa_amd.onnx_proto` (if ``onnx`` is
installed its ModelProto objects are accepted as well: only protobuf field
access is used).  The importer is table driven: one row per ONNX operator
naming the autograd operator or layer that runs it, how its attributes map
to constructor arguments, and which of its inputs are really attributes
(``shape`` of Reshape, ``pads`` of Pad, ...) or layer weights.
"""
import warnings
from collections import OrderedDict, namedtuple

import numpy as np

from . import autograd, device, layer, model, tensor, utils
from . import onnx_proto as onnx
from .onnx_proto import (TensorProto, helper, mapping, numpy_helper)

_OPSET = 11
_IR_VERSION = 6


def onnx_type_to_singa_type(onnx_type):
    """numpy dtype the framework computes in for an ONNX element type (or a
    numpy dtype): floats -> float32, everything integral -> int32."""
    if isinstance(onnx_type, (int, np.integer)):
        np_type = mapping.TENSOR_TYPE_TO_NP_TYPE[int(onnx_type)]
    else:
        np_type = np.dtype(onnx_type)
    if np_type.kind == "f":
        return np.dtype("float32")
    return np.dtype("int32")


# The table form of the above under the reference's name
# (python/singa/sonnx.py:45-62): numpy type -> the type tensors of it are
# computed in, None where there is none.  Differences from the reference's
# table: bool and the unsigned integers map to int32 (the reference has
# float32 for bool and nothing for unsigned types).
NP_TYPE_TO_SINGA_SUPPORT_TYPE = {
    np.dtype(t): (onnx_type_to_singa_type(np.dtype(t))
                  if np.dtype(t).kind in "fiub" else None)
    for t in ("float32", "uint8", "int8", "uint16", "int16", "int32", "int64",
              "bool", "float16", "float64", "complex64", "complex128",
              "uint32", "uint64", "object")}


def __getattr__(name):
    """``cpu_dev`` / ``gpu_dev`` of the reference module
    (python/singa/sonnx.py:69-72), created when first asked for instead of at
    import time (``gpu_dev`` is None on a machine without a GPU)."""
    if name == "cpu_dev":
        value = device.get_default_device()
    elif name == "gpu_dev":
        from .singa_wrap import USE_CUDA
        value = device.create_cuda_gpu() if USE_CUDA else None
    else:
        raise AttributeError("module %r has no attribute %r" % (__name__, name))
    globals()[name] = value
    return value


def _singa_dtype_to_onnx(dt):
    return {tensor.float32: TensorProto.FLOAT, tensor.float16:
            TensorProto.FLOAT16, tensor.int32: TensorProto.INT32}.get(
                dt, TensorProto.FLOAT)


class OnnxAttributes(dict):
    """Attributes of a node as a dict of python values."""

    @staticmethod
    def from_onnx(args):
        d = OnnxAttributes()
        for a in args:
            d[a.name] = helper.get_attribute_value(a)
        return d


class OnnxNode(object):
    """A NodeProto unpacked; ``weight_inputs`` (input name -> layer state
    name) and ``attr_inputs`` (input name -> operator attribute name) mark
    the inputs that do not flow through the operator call."""

    def __init__(self, node):
        self.name = str(node.name).replace(".", "_")
        self.op_type = str(node.op_type)
        self.attrs = OnnxAttributes.from_onnx(node.attribute)
        self.inputs = list(node.input)
        self.outputs = list(node.output)
        self.weight_inputs = {}
        self.attr_inputs = {}

    def getattr(self, key, default=None):
        return self.attrs[key] if key in self.attrs else default

    def set_weight_inputs(self, key, name):
        self.weight_inputs[key] = name

    def del_weight_inputs(self, key):
        del self.weight_inputs[key]

    def set_attr_inputs(self, key, name):
        self.attr_inputs[key] = name

    def del_attr_inputs(self, key):
        del self.attr_inputs[key]


def _s(v):
    return utils.force_unicode(v)


def _window_pads(pads):
    """ONNX lists the pads of a 2-d window as (top, left, bottom, right); the
    layers take (top, bottom, left, right)."""
    pads = tuple(pads)
    if len(pads) == 4:
        return (pads[0], pads[2], pads[1], pads[3])
    return pads


# ---------------------------------------------------------------------------
# import: ONNX operator -> (target, constructor arguments, attribute inputs)
# ---------------------------------------------------------------------------
def _plain(name):
    return (name, None)


class SingaBackend(object):
    _opset_version = _OPSET
    _ir_version = _IR_VERSION

    _rename_operators = {
        'Relu': 'ReLU', 'Sigmoid': 'Sigmoid', 'Add': 'Add', 'MatMul': 'Matmul',
        'Sum': 'Sum', 'Cos': 'Cos', 'Cosh': 'Cosh', 'Sin': 'Sin',
        'Sinh': 'Sinh', 'Tan': 'Tan', 'Tanh': 'Tanh', 'Acos': 'Acos',
        'Acosh': 'Acosh', 'Asin': 'Asin', 'Asinh': 'Asinh', 'Atan': 'Atan',
        'Atanh': 'Atanh', 'Equal': 'Equal', 'Less': 'Less', 'Sign': 'Sign',
        'Div': 'Div', 'Sub': 'Sub', 'Sqrt': 'Sqrt', 'Log': 'Log',
        'Greater': 'Greater', 'Identity': 'Identity', 'Softplus': 'SoftPlus',
        'Softsign': 'SoftSign', 'Mean': 'Mean', 'Pow': 'Pow', 'PRelu': 'PRelu',
        'Mul': 'Mul', 'Max': 'Max', 'Min': 'Min', 'Shape': 'Shape',
        'And': 'And', 'Or': 'Or', 'Xor': 'Xor', 'Not': 'Not',
        'Neg': 'Negative', 'Reciprocal': 'Reciprocal', 'NonZero': 'NonZero',
        'Ceil': 'Ceil', 'Floor': 'Floor', 'Abs': 'Abs', 'Exp': 'Exp',
        'Round': 'Rounde', 'Erf': 'Erf', 'LayerNormalization': 'LayerNorm',
        'Gelu': 'Gelu', 'FusedAttention': 'Attention',
        'FusedLinear': 'LinearBias',
        'Unsqueeze': 'Unsqueeze', 'ScatterElements': 'ScatterElements',
        'Cast': 'Cast', 'Split': 'Split', 'Squeeze': 'Squeeze',
        'GlobalAveragePool': 'GlobalAveragePool', 'LeakyRelu': 'LeakyRelu',
        'ReduceSum': 'ReduceSum', 'ReduceMean': 'ReduceMean',
        'Dropout': 'Dropout', 'ConstantOfShape': 'ConstantOfShape',
        'Transpose': 'Transpose', 'HardSigmoid': 'HardSigmoid', 'Elu': 'Elu',
        'Selu': 'SeLU', 'Concat': 'Concat', 'Softmax': 'SoftMax',
        'Flatten': 'Flatten', 'OneHot': 'OneHot', 'Tile': 'Tile',
        'Gather': 'Gather', 'Reshape': 'Reshape', 'Slice': 'Slice',
        'Clip': 'Clip', 'Expand': 'Expand', 'Pad': 'Pad',
        'Upsample': 'UpSample', 'DepthToSpace': 'DepthToSpace',
        'SpaceToDepth': 'SpaceToDepth', 'Where': 'Where',
        'Gemm': 'layer.Gemm', 'BatchNormalization': 'layer.BatchNorm2d',
        'Conv': 'layer.Conv2d', 'MaxPool': 'layer.Pooling2d',
        'AveragePool': 'layer.Pooling2d',
    }

    # ---- per-operator translators ---------------------------------------------
    @staticmethod
    def _attr_in(node, idx, name):
        """Treat input ``idx`` (if present and non-empty) as attribute."""
        if len(node.inputs) > idx and node.inputs[idx] != '':
            node.set_attr_inputs(node.inputs[idx], name)

    @classmethod
    def _create(cls, n, op):
        """Instance of operator class ``op`` for node ``n``."""
        t = n.op_type
        g = n.getattr
        if t in ('DepthToSpace', 'SpaceToDepth'):
            return op(g("blocksize"), _s(g("mode", "DCR")))
        if t == 'LayerNormalization':
            assert g("axis", -1) == -1, \
                "LayerNormalization is supported over the last axis only"
            return op(g("epsilon", 1e-5))
        if t == 'Gelu':
            assert _s(g("approximate", "none")) == "none"
            return op()
        if t == 'FusedAttention':
            return op(g("scale", 1.0))
        if t == 'Where':
            n.set_attr_inputs(n.inputs[0], 'condition')
            return op(None)
        if t == 'Pad':
            cls._attr_in(n, 1, 'pads')
            cls._attr_in(n, 2, 'constant')
            return op(_s(g("mode", "constant")), None, None)
        if t == 'Upsample':
            cls._attr_in(n, 1, 'scales')
            return op(_s(g("mode", None)), None)
        if t == 'Expand':
            cls._attr_in(n, 1, 'shape')
            return op(None)
        if t == 'Cast':
            to = onnx_type_to_singa_type(g("to"))
            return op(tensor.float32 if to == np.dtype('float32')
                      else tensor.int32)
        if t == 'Split':
            return op(g("axis", 0), g("split", None), len(n.outputs))
        if t in ('Squeeze', 'Unsqueeze'):
            return op(g("axes"))
        if t == 'GlobalAveragePool':
            return op(g("data_format", 'channels_first'))
        if t == 'LeakyRelu':
            return op(g("alpha", 0.01))
        if t in ('ReduceSum', 'ReduceMean'):
            return op(g("axes", None), g("keepdims", 1))
        if t == 'Dropout':
            return op(g("seed", 0), g("ratio", 0))
        if t == 'ConstantOfShape':
            value = g("value", 0)
            if hasattr(value, "data_type"):
                value = numpy_helper.to_array(value).reshape(-1)[0].item()
            return op(value)
        if t == 'Transpose':
            return op(g("perm"))
        if t == 'HardSigmoid':
            return op(g("alpha", 0.2), g("beta", 0.5))
        if t == 'Elu':
            return op(g("alpha", 1.))
        if t == 'Selu':
            return op(g("alpha", 1.67326), g("gamma", 1.0507))
        if t == 'Concat':
            return op(axis=g('axis'))
        if t in ('Softmax', 'Flatten'):
            return op(axis=g('axis', 1))
        if t == 'Gemm':
            n.set_weight_inputs(n.inputs[1], 'W')
            bias = len(n.inputs) == 3
            if bias:
                n.set_weight_inputs(n.inputs[2], 'b')
            return op(None, alpha=g('alpha', 1.), beta=g('beta', 1.),
                      transA=g('transA', 0), transB=g('transB', 0), bias=bias)
        if t == 'OneHot':
            cls._attr_in(n, 1, 'depth')
            cls._attr_in(n, 2, 'values')
            return op(g("axis", -1), None, None)
        if t == 'Tile':
            cls._attr_in(n, 1, 'repeats')
            return op(None)
        if t == 'Gather':
            cls._attr_in(n, 1, 'indices')
            return op(g("axis", 0), None)
        if t == 'Reshape':
            cls._attr_in(n, 1, 'shape')
            return op(None)
        if t == 'Slice':
            cls._attr_in(n, 1, 'starts')
            cls._attr_in(n, 2, 'ends')
            cls._attr_in(n, 3, 'axes')
            cls._attr_in(n, 4, 'steps')
            return op(None, None, None, None)
        if t == 'Clip':
            cls._attr_in(n, 1, 'min')
            cls._attr_in(n, 2, 'max')
            return op(None, None)
        if t == 'BatchNormalization':
            for i, nm in enumerate(('scale', 'bias', 'running_mean',
                                    'running_var')):
                n.set_weight_inputs(n.inputs[i + 1], nm)
            return op(momentum=g('momentum', 0.9))
        if t == 'Conv':
            kernel = tuple(g('kernel_shape'))
            dilation = g('dilations', 1)
            if dilation != 1 and list(dilation) != [1] * len(kernel):
                raise ValueError("Not implemented yet for dilation")
            if len(kernel) > 2:
                raise ValueError("Only implemented for 1d or 2d")
            n.set_weight_inputs(n.inputs[1], 'W')
            bias = len(n.inputs) == 3
            if bias:
                n.set_weight_inputs(n.inputs[2], 'b')
            return op(None, kernel, stride=tuple(g('strides', (1, 1))),
                      padding=_window_pads(g('pads', (0, 0))), dilation=1,
                      group=g('group', 1), bias=bias,
                      pad_mode=_s(g('auto_pad', 'NOTSET')))
        if t in ('MaxPool', 'AveragePool'):
            kernel = tuple(g('kernel_shape'))
            if g('ceil_mode', 0) != 0 or g('count_include_pad', 0) != 0:
                raise ValueError(
                    "Not implemented yet for count_include_pad or ceil_mode")
            if len(kernel) > 2:
                raise ValueError("Only implemented for 1d or 2d")
            return op(kernel, tuple(g('strides', (1, 1))),
                      _window_pads(g('pads', (0, 0))), t == 'MaxPool',
                      _s(g('auto_pad', 'NOTSET')))
        if t == 'ScatterElements':
            cls._attr_in(n, 1, 'indices')
            cls._attr_in(n, 2, 'updates')
            return op(None, None, g("axis", 0))
        return op()

    @classmethod
    def _onnx_constant_to_np(cls, onnx_node):
        return numpy_helper.to_array(onnx_node.getattr('value'))

    @classmethod
    def _onnx_node_to_singa_op(cls, onnx_node, opset_version=_OPSET):
        t = onnx_node.op_type
        assert t in cls._rename_operators, "not support operator: %s" % t
        renamed = cls._rename_operators[t]
        if renamed.startswith('layer.'):
            op_class = getattr(layer, renamed[6:])
        else:
            op_class = getattr(autograd, renamed)
        op = cls._create(onnx_node, op_class)
        op.name = onnx_node.name
        onnx_node.inputs = [i for i in onnx_node.inputs if i != '']
        return op

    # ---- single node (ONNX backend test protocol) ---------------------------------
    @classmethod
    def run_node(cls, node, inputs, device='CPU', opset_version=_OPSET):
        node = OnnxNode(node)
        valid_inputs = [x for x in node.inputs if x != ""]
        assert len(valid_inputs) == len(inputs), (
            "%s: expected %d inputs, but got %d." %
            (node.op_type, len(valid_inputs), len(inputs)))
        op = cls._onnx_node_to_singa_op(node, opset_version)
        dev = _device_of(device)
        states = {}
        xs = []
        for key, val in zip(valid_inputs, inputs):
            val = np.asarray(val)
            val = val.astype(onnx_type_to_singa_type(val.dtype))
            if key in node.weight_inputs:
                states[node.weight_inputs[key]] = val
            elif key in node.attr_inputs:
                setattr(op, node.attr_inputs[key], val)
            else:
                xs.append(tensor.Tensor(device=dev, data=np.atleast_1d(val)))
        if states:
            _init_layer(op, node, xs, states)
        outputs = cls._run_node(op, xs)
        return OrderedDict(zip(node.outputs, outputs))

    @classmethod
    def _run_node(cls, operator, inputs):
        outputs = operator(*inputs)
        if not isinstance(outputs, (list, tuple)):
            outputs = [outputs]
        return list(outputs)

    # ---- whole graph ------------------------------------------------------------------
    @classmethod
    def _parse_graph_params(cls, graph):
        params = {}
        for tp in graph.initializer:
            val = numpy_helper.to_array(tp)
            params[tp.name] = val.astype(onnx_type_to_singa_type(tp.data_type))
        return params

    @classmethod
    def _parse_graph_inputs_outputs(cls, graph, params):
        info = namedtuple('info_tuple', ['name', 'dtype', 'shape'])
        inputs, outputs = [], []
        for t in graph.input:
            if t.name not in params:
                tt = t.type.tensor_type
                inputs.append(info(t.name, tt.elem_type,
                                   [d.dim_value for d in tt.shape.dim]))
        for t in graph.output:
            tt = t.type.tensor_type
            outputs.append(info(t.name, tt.elem_type,
                                [d.dim_value for d in tt.shape.dim]))
        return inputs, outputs

    # replace exported layer-norm / GELU chains by the fused operators
    fuse_patterns = True

    @classmethod
    def _onnx_model_to_singa_ops(cls, graph, opset_version=_OPSET):
        params = cls._parse_graph_params(graph)
        inputs, outputs = cls._parse_graph_inputs_outputs(graph, params)
        operators = []
        op_tuple = namedtuple('operator_tuple', ['node', 'operator'])
        nodes = _fuse_transformer_patterns(graph) if cls.fuse_patterns \
            else graph.node
        for node in nodes:
            if not node.name:
                node.name = "%s_%d" % (str(node.op_type), len(operators))
            n = OnnxNode(node)
            if n.op_type == 'Constant':
                val = cls._onnx_constant_to_np(n)
                params[n.outputs[0]] = val.astype(
                    onnx_type_to_singa_type(val.dtype))
            else:
                operators.append(op_tuple(
                    n, cls._onnx_node_to_singa_op(n, opset_version)))
        return params, inputs, outputs, operators

    @classmethod
    def prepare(cls, model, device='CPU', **kwargs):
        """ModelProto -> SingaRep.  ``device``: 'CPU', 'CUDA' or a Device."""
        opset_version = None
        for imp in model.opset_import:
            if imp.domain in ("", "ai.onnx"):
                opset_version = imp.version
                if imp.version > cls._opset_version:
                    warnings.warn(
                        "The imported opertor set verion {} is larger than "
                        "the supported version {}.".format(
                            imp.version, cls._opset_version))
            else:
                warnings.warn("Unrecognized operator set %s" % imp.domain)
        if model.ir_version > cls._ir_version:
            warnings.warn(
                "The imported ir verion {} is larger than the supported "
                "version {}.".format(model.ir_version, cls._ir_version))
        params, inputs, outputs, layers = cls._onnx_model_to_singa_ops(
            model.graph, opset_version or cls._opset_version)
        return SingaRep(params, inputs, outputs, layers, device)


def _device_of(dev):
    if isinstance(dev, str):
        if dev.upper() == 'CPU':
            return device.get_default_device()
        return device.create_cuda_gpu()
    return dev


def _init_layer(op, node, inputs, states):
    """Give a layer the geometry the ONNX weights imply, create its
    parameters from the first inputs and load the weights."""
    if node.op_type == "Conv":
        op.nb_kernels = states['W'].shape[0]
    elif node.op_type == "Gemm":
        op.nb_kernels = states['W'].shape[0 if op.transB == 1 else -1]
        if op.bias:
            op.bias_shape = states['b'].shape
    if callable(getattr(op, "initialize", None)) and not op._initialized:
        op.initialize(*inputs)
        op._initialized = True
    named = {getattr(op, k).name: tensor.from_numpy(np.ascontiguousarray(v))
             for k, v in states.items()}
    op.set_states(named)


# ---------------------------------------------------------------------------
# pattern fusion on the imported node list
# ---------------------------------------------------------------------------
def _fuse_transformer_patterns(graph):
    """Exporters write layer normalisation and GELU as chains of elementwise
    and reduction nodes:

        ReduceMean -> Sub -> Pow(2) -> ReduceMean -> Add(eps) -> Sqrt -> Div
                   -> Mul(gamma) -> Add(beta)
        Div(sqrt 2) | Mul(1 / sqrt 2) -> Erf -> Add(1) -> Mul(x) -> Mul(0.5)

    Each chain whose intermediate tensors have no other consumer is replaced
    by one LayerNormalization / Gelu node (one kernel instead of nine / five,
    no intermediate tensors).  The attention chain becomes FusedAttention and
    MatMul -> Add of a constant matrix and vector becomes FusedLinear (bias in
    the epilogue of the product).  Returns the new node list."""
    nodes = list(graph.node)
    consts = {t.name: numpy_helper.to_array(t) for t in graph.initializer}
    for n in nodes:
        if n.op_type == 'Constant':
            for a in n.attribute:
                if a.name == 'value':
                    consts[n.output[0]] = numpy_helper.to_array(a.t)
    outputs = set(o.name for o in graph.output)
    producer, consumers = {}, {}
    for i, n in enumerate(nodes):
        for o in n.output:
            producer[o] = i
        for x in n.input:
            consumers.setdefault(x, []).append(i)

    def scalar(name):
        v = consts.get(name)
        if v is None or v.size != 1:
            return None
        return float(v.reshape(-1)[0])

    def only(name, count=1):
        """indices of the consumers if the tensor has exactly ``count`` of
        them and is not a graph output"""
        c = consumers.get(name, [])
        if len(c) != count or name in outputs:
            return None
        return c

    def other(node, name):
        ins = list(node.input)
        if len(ins) != 2 or name not in ins:
            return None
        return ins[1 - ins.index(name)] if ins[0] != ins[1] else None

    def attr(node, key, default=None):
        for a in node.attribute:
            if a.name == key:
                return helper.get_attribute_value(a)
        return default

    def last_axis_mean(node):
        if node.op_type != 'ReduceMean' or attr(node, 'keepdims', 1) != 1:
            return False
        axes = attr(node, 'axes')
        if axes is None and len(node.input) > 1:
            v = consts.get(node.input[1])
            axes = None if v is None else [int(a) for a in v.reshape(-1)]
        return axes is not None and list(axes) == [-1]

    dead, replace = set(), {}

    # ---- layer normalisation ------------------------------------------------
    for i, n in enumerate(nodes):
        if i in dead or not last_axis_mean(n):
            continue
        x = n.input[0]
        c = only(n.output[0])
        if not c or nodes[c[0]].op_type != 'Sub' or \
                list(nodes[c[0]].input) != [x, n.output[0]]:
            continue
        sub = c[0]
        d = nodes[sub].output[0]
        c = only(d, 2)
        if not c:
            continue
        kinds = sorted(nodes[k].op_type for k in c)
        if kinds != ['Div', 'Pow']:
            continue
        pw = [k for k in c if nodes[k].op_type == 'Pow'][0]
        dv = [k for k in c if nodes[k].op_type == 'Div'][0]
        if nodes[pw].input[0] != d or scalar(nodes[pw].input[1]) != 2.0:
            continue
        c = only(nodes[pw].output[0])
        if not c or not last_axis_mean(nodes[c[0]]):
            continue
        mean2 = c[0]
        c = only(nodes[mean2].output[0])
        if not c or nodes[c[0]].op_type != 'Add':
            continue
        add_eps = c[0]
        eps_name = other(nodes[add_eps], nodes[mean2].output[0])
        eps = scalar(eps_name) if eps_name else None
        if eps is None:
            continue
        c = only(nodes[add_eps].output[0])
        if not c or nodes[c[0]].op_type != 'Sqrt':
            continue
        sq = c[0]
        c = only(nodes[sq].output[0])
        if not c or c[0] != dv or \
                list(nodes[dv].input) != [d, nodes[sq].output[0]]:
            continue
        c = only(nodes[dv].output[0])
        if not c or nodes[c[0]].op_type != 'Mul':
            continue
        mul = c[0]
        gamma = other(nodes[mul], nodes[dv].output[0])
        if gamma not in consts or consts[gamma].ndim != 1:
            continue
        c = only(nodes[mul].output[0])
        if not c or nodes[c[0]].op_type != 'Add':
            continue
        addb = c[0]
        beta = other(nodes[addb], nodes[mul].output[0])
        if beta not in consts or consts[beta].shape != consts[gamma].shape:
            continue
        chain = [i, sub, pw, mean2, add_eps, sq, dv, mul, addb]
        dead.update(chain)
        replace[addb] = helper.make_node(
            'LayerNormalization', [x, gamma, beta], [nodes[addb].output[0]],
            name=(nodes[addb].name or "ln_%d" % addb) + "_fused", axis=-1,
            epsilon=eps)

    # ---- GELU ---------------------------------------------------------------
    for i, n in enumerate(nodes):
        if i in dead or n.op_type != 'Erf':
            continue
        p = producer.get(n.input[0])
        if p is None or p in dead or only(n.input[0]) is None:
            continue
        pre = nodes[p]
        x = None
        if pre.op_type == 'Div' and scalar(pre.input[1]) is not None and \
                abs(scalar(pre.input[1]) - 2 ** 0.5) < 1e-3:
            x = pre.input[0]
        elif pre.op_type == 'Mul':
            for a, b in ((0, 1), (1, 0)):
                v = scalar(pre.input[b])
                if v is not None and abs(v - 0.5 ** 0.5) < 1e-3:
                    x = pre.input[a]
        if x is None:
            continue
        c = only(n.output[0])
        if not c or nodes[c[0]].op_type != 'Add':
            continue
        add1 = c[0]
        o = other(nodes[add1], n.output[0])
        if o is None or scalar(o) != 1.0:
            continue
        c = only(nodes[add1].output[0])
        if not c or nodes[c[0]].op_type != 'Mul':
            continue
        m1 = c[0]
        o = other(nodes[m1], nodes[add1].output[0])
        chain, last = None, None
        if o == x:                                   # (x * (1 + erf)) * 0.5
            c = only(nodes[m1].output[0])
            if c and nodes[c[0]].op_type == 'Mul':
                h = other(nodes[c[0]], nodes[m1].output[0])
                if h is not None and scalar(h) == 0.5:
                    chain, last = [p, i, add1, m1, c[0]], c[0]
        elif o is not None and producer.get(o) is not None:  # (0.5 x) * (1 + erf)
            q = producer[o]
            if nodes[q].op_type == 'Mul' and q not in dead and only(o):
                h = other(nodes[q], x)
                if h is not None and scalar(h) == 0.5:
                    chain, last = [p, i, add1, q, m1], m1
        if chain is None:
            continue
        dead.update(chain)
        replace[last] = helper.make_node(
            'Gelu', [x], [nodes[last].output[0]],
            name=(nodes[last].name or "gelu_%d" % last) + "_fused")

    # ---- scaled-dot-product attention ---------------------------------------
    #   MatMul(q, kT) -> Div(c) | Mul(c) -> [Add(mask)] -> Softmax(last axis)
    #   -> MatMul(., v) -> Transpose(0, 2, 1, 3) -> Reshape(B, S, H * D)
    # with q = Transpose(Reshape(x, [.., .., H, 64]), (0, 2, 1, 3)): the head
    # dimension must be known to be 64, the size the fused kernel is built for
    def const_shape(name):
        v = consts.get(name)
        return None if v is None else [int(a) for a in v.reshape(-1)]

    def head_split(name):
        """(H, D) if ``name`` is Transpose(Reshape(., [..., H, D]), 0213)"""
        p = producer.get(name)
        if p is None or nodes[p].op_type != 'Transpose' or \
                list(attr(nodes[p], 'perm', [])) != [0, 2, 1, 3]:
            return None
        r = producer.get(nodes[p].input[0])
        if r is None or nodes[r].op_type != 'Reshape' or \
                len(nodes[r].input) < 2:
            return None
        shp = const_shape(nodes[r].input[1])
        if shp is None or len(shp) != 4:
            return None
        return shp[2], shp[3]

    for i, n in enumerate(nodes):
        if i in dead or n.op_type != 'Softmax' or \
                attr(n, 'axis', 1) not in (-1, 3):
            continue
        chain = [i]
        cur = producer.get(n.input[0])
        if cur is None or cur in dead or only(n.input[0]) is None:
            continue
        mask = None
        if nodes[cur].op_type == 'Add':
            a0, a1 = nodes[cur].input
            pa, pb = producer.get(a0), producer.get(a1)
            scaled = None
            for cand, oth, pc in ((a0, a1, pa), (a1, a0, pb)):
                if pc is not None and pc not in dead and \
                        nodes[pc].op_type in ('Mul', 'Div', 'MatMul') and \
                        only(cand) is not None:
                    scaled, mask = cand, oth
                    break
            if scaled is None:
                continue
            chain.append(cur)
            cur = producer[scaled]
        scale = 1.0
        if nodes[cur].op_type in ('Mul', 'Div'):
            sc_node = nodes[cur]
            val, src = None, None
            if sc_node.op_type == 'Div':
                val = scalar(sc_node.input[1])
                src = sc_node.input[0]
                if val is not None:
                    val = 1.0 / val
            else:
                for a, b in ((0, 1), (1, 0)):
                    v = scalar(sc_node.input[b])
                    if v is not None:
                        val, src = v, sc_node.input[a]
                        break
            if val is None or only(src) is None:
                continue
            scale = val
            chain.append(cur)
            cur = producer.get(src)
            if cur is None or cur in dead:
                continue
        if nodes[cur].op_type != 'MatMul':
            continue
        mm1 = cur
        chain.append(mm1)
        q_name, kt_name = nodes[mm1].input
        hd = head_split(q_name)
        if hd is None or hd[1] != 64:
            continue
        c = only(n.output[0])
        if not c or nodes[c[0]].op_type != 'MatMul' or \
                nodes[c[0]].input[0] != n.output[0]:
            continue
        mm2 = c[0]
        v_name = nodes[mm2].input[1]
        if head_split(v_name) != hd:
            continue
        c = only(nodes[mm2].output[0])
        if not c or nodes[c[0]].op_type != 'Transpose' or \
                list(attr(nodes[c[0]], 'perm', [])) != [0, 2, 1, 3]:
            continue
        tr = c[0]
        c = only(nodes[tr].output[0])
        if not c or nodes[c[0]].op_type != 'Reshape' or \
                len(nodes[c[0]].input) < 2:
            continue
        rs = c[0]
        shp = const_shape(nodes[rs].input[1])
        if shp is None or len(shp) != 3 or \
                shp[2] not in (-1, hd[0] * hd[1]):
            continue
        chain += [mm2, tr, rs]
        if any(k in dead for k in chain):
            continue
        dead.update(chain)
        ins = [q_name, kt_name, v_name] + ([mask] if mask else [])
        replace[rs] = helper.make_node(
            'FusedAttention', ins, [nodes[rs].output[0]],
            name=(nodes[rs].name or "attn_%d" % rs) + "_fused",
            scale=float(scale))

    # ---- linear layer: MatMul(x, W) -> Add(b), W (K, N) and b (N,) constant --
    for i, n in enumerate(nodes):
        if i in dead or n.op_type != 'MatMul':
            continue
        wname = n.input[1]
        if wname not in consts or consts[wname].ndim != 2 or \
                consts[wname].dtype.kind != 'f':
            continue
        c = only(n.output[0])
        if not c or c[0] in dead or nodes[c[0]].op_type != 'Add':
            continue
        bname = other(nodes[c[0]], n.output[0])
        if bname not in consts or \
                consts[bname].shape != (consts[wname].shape[1],):
            continue
        dead.update([i, c[0]])
        replace[c[0]] = helper.make_node(
            'FusedLinear', [n.input[0], wname, bname],
            [nodes[c[0]].output[0]],
            name=(nodes[c[0]].name or "linear_%d" % c[0]) + "_fused")

    out = []
    for i, n in enumerate(nodes):
        if i in replace:
            out.append(replace[i])
        elif i not in dead:
            out.append(n)
    return out


class SingaRep(object):
    """Interprets the imported graph in topological order.  ``run(x)`` takes
    numpy arrays (or Tensors when driven by ``SONNXModel``) and returns the
    graph outputs; ``last_layers=k`` stops after layer k (transfer learning),
    ``aux_output`` names extra tensors to return."""

    def __init__(self, params, inputs, outputs, layers, device):
        self.inputs = inputs
        self.states = params
        self.outputs = outputs
        self.dev = _device_of(device)
        self.layers = layers
        self.tensor_count = {}
        self.has_initialized = False
        self.is_graph = False
        self._const_cache = {}
        # set by SONNXModel(trainable_weights=True): a registry that receives
        # the float initializers as parameters
        self.weights = None
        # None: floats stay float32; float16 / bfloat16: see set_float_dtype
        self.float_dtype = None

    def initialize(self):
        self.outputs_info = {o.name: o for o in self.outputs}
        self._layers = []
        for node, operator in self.layers:
            for key, name in list(node.weight_inputs.items()):
                if key not in self.states:
                    # weight computed by the graph: treat as attribute input
                    node.set_attr_inputs(key, name)
                    node.del_weight_inputs(key)
            self.__dict__[node.name] = operator
            self._layers.append(node)

    def init_tensor_count(self):
        self.tensor_count = {}
        for node, _ in self.layers:
            for inp in (node.inputs + list(node.attr_inputs.keys()) +
                        list(node.weight_inputs.keys())):
                self.tensor_count[inp] = self.tensor_count.get(inp, 0) + 1

    def set_float_dtype(self, dtype):
        """Run the float32 parts of the graph in ``dtype`` (float16 /
        bfloat16): constants, fed float inputs and Cast-to-float nodes."""
        self.float_dtype = dtype
        self._const_cache = {}
        for _, operator in self.layers:
            if isinstance(operator, autograd.Cast) and \
                    operator.to == tensor.float32:
                operator.to = dtype

    def to_input_tensor(self, x):
        tensor_dict = {}
        for key, val in zip(self.inputs, x):
            if not isinstance(val, tensor.Tensor):
                val = np.asarray(val)
                val = np.atleast_1d(val.astype(
                    onnx_type_to_singa_type(key.dtype)))
                val = tensor.Tensor(device=self.dev, data=val)
            if self.float_dtype is not None and \
                    val.dtype == tensor.float32:
                val = val.as_type(self.float_dtype)
            tensor_dict[key.name] = val
        return tensor_dict

    def to_output_tensor(self, y, out_name):
        if not self.is_graph:
            y = tensor.to_numpy(y)
            if out_name in self.outputs_info:
                dt = self.outputs_info[out_name].dtype
                if dt in mapping.TENSOR_TYPE_TO_NP_TYPE:
                    y = y.astype(mapping.TENSOR_TYPE_TO_NP_TYPE[dt])
        return y

    def get_s(self, name, node, tensor_dict):
        if name in node.attr_inputs:
            return tensor_dict[name]
        return self.states[name]

    def run(self, x, **kwargs):
        if not self.has_initialized:
            self.initialize()
            if isinstance(x[0], tensor.Tensor):
                self.dev = x[0].device
        outputs_dict = OrderedDict()
        last = kwargs.get('last_layers', len(self._layers) - 1)
        if last < 0:
            last = (last + 1) % len(self._layers)
        if last != len(self._layers) - 1:
            for o in self._layers[last].outputs:
                outputs_dict[o] = None
        else:
            for o in self.outputs:
                outputs_dict[o.name] = None
        for o in kwargs.get('aux_output', ()):
            outputs_dict[o] = None

        tensor_dict = self.to_input_tensor(x)
        self.init_tensor_count()
        for node in self._layers[:last + 1]:
            op = self.__dict__[node.name]
            inputs = []
            for inp in node.inputs:
                if inp in node.weight_inputs or inp in node.attr_inputs:
                    continue
                if inp in tensor_dict:
                    inputs.append(tensor_dict[inp])
                elif inp in self.states:
                    # constants are uploaded once, not at every run
                    key = (inp, id(self.dev))
                    t = self._const_cache.get(key)
                    if t is None:
                        val = np.atleast_1d(self.states[inp])
                        t = tensor.Tensor(device=self.dev, data=val)
                        if self.float_dtype is not None and \
                                val.dtype.kind == 'f':
                            # converted now, also while a step is recorded
                            prev = self.dev.graph_enabled()
                            self.dev.EnableGraph(False)
                            try:
                                t = t.as_type(self.float_dtype)
                            finally:
                                self.dev.EnableGraph(prev)
                        t.requires_grad = False
                        if self.weights is not None and \
                                val.dtype.kind == 'f' and val.size > 1:
                            # an initializer consumed as data (MatMul / Add /
                            # Gather operand, layer-norm scale): a trainable
                            # parameter of the model that owns ``weights``
                            t.requires_grad = True
                            t.stores_grad = True
                            self.weights.add(inp, t)
                        self._const_cache[key] = t
                    inputs.append(t)
                else:
                    raise KeyError("Not found the input %s for operation %s" %
                                   (inp, node.name))
            # attributes delivered as inputs
            for key, name in node.attr_inputs.items():
                ts = tensor_dict[key] if key in tensor_dict \
                    else self.states[key]
                if isinstance(ts, tensor.Tensor):
                    if name == 'indices' and ts.dtype == tensor.int32 and \
                            isinstance(op, autograd.Gather):
                        # stays on the device: no synchronisation, and a
                        # replayed step gathers with the ids of its own batch
                        op.indices = ts
                        continue
                    ts = tensor.to_numpy(ts)
                if name in ('W', 'b', 'scale', 'bias', 'running_mean',
                            'running_var'):
                    continue  # handled below as layer states
                setattr(op, name, ts)
            # layers: create parameters on first use and load the weights
            if callable(getattr(op, "initialize", None)) and \
                    not op._initialized:
                states = {name: self.states[key]
                          for key, name in node.weight_inputs.items()}
                for key, name in node.attr_inputs.items():
                    if name in ('W', 'b', 'scale', 'bias', 'running_mean',
                                'running_var'):
                        ts = tensor_dict[key]
                        states[name] = tensor.to_numpy(ts)
                prev = self.dev.graph_enabled()
                self.dev.EnableGraph(False)
                try:
                    _init_layer(op, node, inputs, states)
                finally:
                    self.dev.EnableGraph(prev)
            outputs = SingaBackend._run_node(op, inputs)
            for inp in node.inputs:
                if inp in self.tensor_count:
                    self.tensor_count[inp] -= 1
                    if self.tensor_count[inp] == 0:
                        tensor_dict.pop(inp, None)
                        del self.tensor_count[inp]
            for outp, val in zip(node.outputs, outputs):
                tensor_dict[outp] = val
                if outp in outputs_dict:
                    outputs_dict[outp] = self.to_output_tensor(val, outp)
        self.has_initialized = True
        return list(outputs_dict.values())


class _GraphWeights(layer.Layer):
    """The float initializers of an imported graph that are consumed as plain
    tensor operands (MatMul / Add / Gather inputs, layer-norm scales), kept as
    named parameters so that the optimizer updates them and checkpoints hold
    them.  Transformer exports keep all their weights in this form."""

    def __init__(self):
        super(_GraphWeights, self).__init__()
        self._initialized = True
        self.table = OrderedDict()

    def add(self, name, t):
        t.name = (self.name or "graph_weights") + layer.Layer.sep + name
        self.table[t.name] = t

    def get_params(self):
        return dict(self.table)

    def set_params(self, params):
        for k, v in params.items():
            if k in self.table:
                self.table[k].copy_from(v)

    def get_states(self):
        return self.get_params()

    def set_states(self, states):
        self.set_params(states)


class SONNXModel(model.Model):
    """A Model whose forward pass is an imported ONNX graph; subclass it and
    add ``train_one_batch`` to fine-tune (reference sonnx.py:2196-2220).
    ``trainable_weights=True`` also turns the float initializers that the
    graph feeds to MatMul / Add / Gather into parameters (the weights of a
    transformer export), not only those owned by Conv / Gemm / BatchNorm
    layers.  ``dtype=tensor.float16`` (or ``bfloat16``) runs a float32 graph
    in 16 bits: float initializers are converted when they are uploaded, float
    inputs when they are fed, and Cast-to-float nodes produce that type (the
    reference imports float32 graphs as float32 only)."""

    def __init__(self, onnx_model, trainable_weights=False, dtype=None):
        super(SONNXModel, self).__init__()
        self.sg_ir = prepare(onnx_model)
        if dtype is not None:
            self.sg_ir.set_float_dtype(dtype)
        for node, operator in self.sg_ir.layers:
            if isinstance(operator, layer.Layer):
                # registered as sub-layer: its weights become parameters of
                # this model (trainable, saved with the checkpoint)
                setattr(self, node.name, operator)
            else:
                self.__dict__[node.name] = operator
        if trainable_weights:
            self.graph_weights = _GraphWeights()
            self.sg_ir.weights = self.graph_weights
        self.sg_ir.is_graph = True

    def forward(self, *input, aux_output=(), **kwargs):
        return self.sg_ir.run(input, aux_output=aux_output, **kwargs)


# ---------------------------------------------------------------------------
# export: autograd tape -> ONNX graph
# ---------------------------------------------------------------------------
def _np_i64(v):
    return np.asarray(v, dtype=np.int64).reshape(-1)


class SingaFrontend(object):
    _target_opset_version = _OPSET

    # operator class name -> ONNX op type, for operators without attributes
    _rename_operators = {
        'ReLU': 'Relu', 'Sigmoid': 'Sigmoid', 'Add': 'Add', 'Matmul': 'MatMul',
        'Sum': 'Sum', 'Cos': 'Cos', 'Cosh': 'Cosh', 'Sin': 'Sin',
        'Sinh': 'Sinh', 'Tan': 'Tan', 'Tanh': 'Tanh', 'Acos': 'Acos',
        'Acosh': 'Acosh', 'Asin': 'Asin', 'Asinh': 'Asinh', 'Atan': 'Atan',
        'Atanh': 'Atanh', 'Equal': 'Equal', 'Less': 'Less', 'Sign': 'Sign',
        'Div': 'Div', 'Sub': 'Sub', 'Sqrt': 'Sqrt', 'Log': 'Log',
        'Greater': 'Greater', 'Identity': 'Identity', 'SoftPlus': 'Softplus',
        'SoftSign': 'Softsign', 'Mean': 'Mean', 'Pow': 'Pow', 'PRelu': 'PRelu',
        'Mul': 'Mul', 'Max': 'Max', 'Min': 'Min', 'Shape': 'Shape',
        'And': 'And', 'Or': 'Or', 'Xor': 'Xor', 'Not': 'Not',
        'Negative': 'Neg', 'Reciprocal': 'Reciprocal', 'NonZero': 'NonZero',
        'Ceil': 'Ceil', 'Floor': 'Floor', 'Abs': 'Abs', 'Exp': 'Exp',
        'Rounde': 'Round', 'Erf': 'Erf', 'AddBias': 'Add',
        'Gelu': 'Gelu', 'GlobalAveragePool': 'GlobalAveragePool',
    }

    @classmethod
    def _attrs_and_extra(cls, op):
        """(onnx op type, attribute dict, [(suffix, ndarray)] extra inputs)."""
        n = type(op).__name__
        h = getattr(op, "handle", None)
        if n in cls._rename_operators:
            return cls._rename_operators[n], {}, []
        if n == '_Conv2d':
            odd = op.odd_padding
            pads = [h.pad_h + odd[0], h.pad_w + odd[2], h.pad_h + odd[1],
                    h.pad_w + odd[3]]
            return 'Conv', dict(kernel_shape=[h.kernel_h, h.kernel_w],
                                pads=pads, strides=[h.stride_h, h.stride_w],
                                group=h.group), []
        if n == '_Pooling2d':
            odd = op.odd_padding
            pads = [h.pad_h + odd[0], h.pad_w + odd[2], h.pad_h + odd[1],
                    h.pad_w + odd[3]]
            return ('MaxPool' if h.is_max_pooling else 'AveragePool'), dict(
                kernel_shape=[h.kernel_h, h.kernel_w], pads=pads,
                strides=[h.stride_h, h.stride_w]), []
        if n == '_BatchNorm2d':
            return 'BatchNormalization', dict(momentum=float(h.factor),
                                              epsilon=1e-5), [
                ("mean", tensor.to_numpy(tensor.from_raw_tensor(
                    op.running_mean))),
                ("var", tensor.to_numpy(tensor.from_raw_tensor(
                    op.running_var)))]
        if n == 'LayerNorm':
            return 'LayerNormalization', dict(axis=-1,
                                              epsilon=float(op.eps)), []
        if n == 'Gemm':
            return 'Gemm', dict(alpha=float(op.alpha), beta=float(op.beta),
                                transA=int(op.transA),
                                transB=int(op.transB)), []
        if n == 'SoftMax':
            return 'Softmax', dict(axis=int(op.axis)), []
        if n == 'Flatten':
            return 'Flatten', dict(axis=int(op.axis)), []
        if n == 'Concat':
            return 'Concat', dict(axis=int(op.axis)), []
        if n == 'Transpose':
            return 'Transpose', dict(perm=[int(p) for p in op.perm]), []
        if n in ('Squeeze', 'Unsqueeze'):
            ax = op.axis if isinstance(op.axis, (list, tuple)) else [op.axis]
            return n, dict(axes=[int(a) for a in ax]), []
        if n in ('ReduceSum', 'ReduceMean'):
            d = dict(keepdims=int(op.keepdims))
            if op.axes is not None:
                d['axes'] = [int(a) for a in op.axes]
            return n, d, []
        if n == 'LeakyRelu':
            return 'LeakyRelu', dict(alpha=float(op.a)), []
        if n == 'Elu':
            return 'Elu', dict(alpha=float(op.alpha)), []
        if n == 'SeLU':
            return 'Selu', dict(alpha=float(op.alpha),
                                gamma=float(op.gamma)), []
        if n == 'HardSigmoid':
            return 'HardSigmoid', dict(alpha=float(op.alpha),
                                       beta=float(op.gamma)), []
        if n == 'Dropout':
            return 'Dropout', dict(ratio=float(op.ratio)), []
        if n == 'Cast':
            return 'Cast', dict(to=_singa_dtype_to_onnx(op.to)), []
        if n == 'Split':
            d = dict(axis=int(op.axis))
            if op.parts is not None:
                d['split'] = [int(p) for p in op.parts]
            return 'Split', d, []
        if n in ('DepthToSpace', 'SpaceToDepth'):
            d = dict(blocksize=int(op.blocksize))
            if n == 'DepthToSpace':
                d['mode'] = op.mode
            return n, d, []
        if n == 'ConstantOfShape':
            return n, dict(value=helper.make_tensor(
                "value", TensorProto.FLOAT, [1], [float(op.value)])), []
        if n == 'Reshape':
            return n, {}, [("shape", _np_i64(op.shape))]
        if n == 'Expand':
            return n, {}, [("shape", _np_i64(op.shape))]
        if n == 'Tile':
            return n, {}, [("repeats", _np_i64(op.repeats))]
        if n == 'Gather':
            if isinstance(op.indices, tensor.Tensor):
                # indices fed at run time: named as an input by the caller
                return n, dict(axis=int(op.axis)), []
            return n, dict(axis=int(op.axis)), [
                ("indices", np.asarray(op.indices, dtype=np.int64))]
        if n == 'Clip':
            extra = []
            for nm in ("min", "max"):
                v = getattr(op, nm)
                if v is not None:
                    extra.append((nm, np.asarray(v, dtype=np.float32)))
            return n, {}, extra
        if n == 'Slice':
            extra = [("starts", _np_i64(op.starts)), ("ends", _np_i64(op.ends))]
            axes = op.axes if op.axes is not None else list(
                range(len(op.starts)))
            extra.append(("axes", _np_i64(axes)))
            if op.steps is not None:
                extra.append(("steps", _np_i64(op.steps)))
            return n, {}, extra
        if n == 'Pad':
            return n, dict(mode=op.mode), [
                ("pads", _np_i64(op.pads)),
                ("constant", np.asarray(op.constant, dtype=np.float32))]
        if n == 'UpSample':
            return 'Upsample', dict(mode=op.mode), [
                ("scales", np.asarray(op.scales, dtype=np.float32))]
        if n == 'OneHot':
            return n, dict(axis=int(op.axis)), [
                ("depth", np.asarray([op.depth], dtype=np.int64)),
                ("values", np.asarray(op.values, dtype=np.float32))]
        if n == 'ScatterElements':
            return n, dict(axis=int(op.axis)), [
                ("indices", np.asarray(op.indices, dtype=np.int64)),
                ("updates", np.asarray(
                    tensor.to_numpy(op.updates) if isinstance(
                        op.updates, tensor.Tensor) else op.updates,
                    dtype=np.float32))]
        if n == 'Where':
            cond = op.condition
            if isinstance(cond, tensor.CTensor):
                cond = tensor.from_raw_tensor(cond)
            if isinstance(cond, tensor.Tensor):
                cond = tensor.to_numpy(cond)
            return n, {}, [("condition", np.asarray(cond) != 0)]
        raise NotImplementedError("operator %s has no ONNX translation" % n)

    @classmethod
    def _input_names(cls, op):
        names = []
        for (src_op, x_id, _, _) in op.src:
            if type(src_op).__name__ == 'Dummy':
                names.append(src_op.name)
            else:
                names.append(src_op.output_name(src_op.y_id2idx[x_id]))
        return names

    @classmethod
    def _expand_batchnorm_act(cls, op, nodes, initializers):
        """The fused normalisation operator has no ONNX counterpart: it is
        written as BatchNormalization [-> Add] -> Relu, which is also what an
        importer turns back into separate operators."""
        ins = cls._input_names(op)          # x, scale, bias[, residual]
        for suffix, raw in (("mean", op.running_mean),
                            ("var", op.running_var)):
            nm = "%s_%s" % (op.name, suffix)
            initializers.append(numpy_helper.from_array(
                tensor.to_numpy(tensor.from_raw_tensor(raw)), nm))
            ins.insert(3 + (suffix == "var"), nm)
        out = op.output_name(0)
        bn_out = out + "_bn"
        nodes.append(helper.make_node(
            'BatchNormalization', ins[:5], [bn_out], name=op.name + "_bn",
            momentum=float(op.handle.factor), epsilon=1e-5))
        if len(ins) > 5:                    # residual branch
            nodes.append(helper.make_node('Add', [bn_out, ins[5]],
                                          [out + "_sum"],
                                          name=op.name + "_add"))
            bn_out = out + "_sum"
        nodes.append(helper.make_node('Relu', [bn_out], [out],
                                      name=op.name + "_relu"))

    @classmethod
    def _expand_linear_bias(cls, op, nodes):
        """x @ W + b as MatMul -> Add, the chain the importer fuses."""
        x, w, b = cls._input_names(op)
        out = op.output_name(0)
        nodes.append(helper.make_node('MatMul', [x, w], [out + "_mm"],
                                      name=op.name + "_mm"))
        nodes.append(helper.make_node('Add', [out + "_mm", b], [out],
                                      name=op.name + "_add"))

    @classmethod
    def _expand_attention(cls, op, nodes, initializers):
        """The fused attention operator as the chain exporters write (and the
        importer recognises): MatMul -> Mul(scale) -> [Add(mask)] -> Softmax
        -> MatMul -> Transpose -> Reshape (merge heads)."""
        ins = cls._input_names(op)            # q, kt, v[, mask]
        out = op.output_name(0)

        def const(suffix, arr):
            nm = "%s_%s" % (op.name, suffix)
            initializers.append(numpy_helper.from_array(arr, nm))
            return nm

        def node(kind, inputs, suffix, **attrs):
            nm = out + "_" + suffix
            nodes.append(helper.make_node(kind, inputs, [nm],
                                          name=op.name + "_" + suffix,
                                          **attrs))
            return nm
        s_ = node('MatMul', [ins[0], ins[1]], "qk")
        s_ = node('Mul', [s_, const("scale", np.float32(op.scale))], "scaled")
        if len(ins) > 3:
            s_ = node('Add', [s_, ins[3]], "masked")
        p_ = node('Softmax', [s_], "prob", axis=-1)
        c_ = node('MatMul', [p_, ins[2]], "ctx")
        c_ = node('Transpose', [c_], "heads", perm=[0, 2, 1, 3])
        nodes.append(helper.make_node(
            'Reshape', [c_, const("merge", np.array([0, 0, -1], np.int64))],
            [out], name=op.name + "_merge"))

    @classmethod
    def singa_to_onnx_graph(cls, inputs, y, model_name="sonnx"):
        """GraphProto of the computation that produced ``y`` (a Tensor or a
        list of Tensors) from ``inputs`` (list of Tensors)."""
        ys = y if isinstance(y, (list, tuple)) else [y]
        nodes, initializers = [], []
        weights, graph_inputs = OrderedDict(), OrderedDict()
        done = set()
        for yy in ys:
            ops, ws, ins = utils.post_order_recursive(yy.creator, yy)
            weights.update(ws)
            graph_inputs.update(ins)
            for name, op in ops.items():
                if id(op) in done:
                    continue
                done.add(id(op))
                if type(op).__name__ == '_BatchNormAct2d':
                    cls._expand_batchnorm_act(op, nodes, initializers)
                    continue
                if type(op).__name__ == 'LinearBias':
                    cls._expand_linear_bias(op, nodes)
                    continue
                if type(op).__name__ == 'Attention':
                    cls._expand_attention(op, nodes, initializers)
                    continue
                op_type, attrs, extra = cls._attrs_and_extra(op)
                in_names = cls._input_names(op)
                if op_type == 'Gather' and isinstance(
                        getattr(op, "indices", None), tensor.Tensor):
                    src = op.indices.creator
                    if type(src).__name__ != 'Dummy':
                        raise NotImplementedError(
                            "Gather indices computed inside the graph cannot "
                            "be exported")
                    in_names.append(src.name)
                for suffix, arr in extra:
                    nm = "%s_%s" % (op.name, suffix)
                    initializers.append(numpy_helper.from_array(arr, nm))
                    if op_type == 'Where':
                        in_names.insert(0, nm)  # condition comes first
                    else:
                        in_names.append(nm)
                out_names = [op.output_name(i) for i in
                             range(len(op.y_id2idx))]
                nodes.append(helper.make_node(op_type, in_names, out_names,
                                              name=op.name, **attrs))
        for name, t in weights.items():
            initializers.append(numpy_helper.from_array(
                tensor.to_numpy(t), name))
        known_inputs = {x.creator.name: x for x in inputs}
        g_in = []
        for x in inputs:
            g_in.append(helper.make_tensor_value_info(
                x.creator.name, _singa_dtype_to_onnx(x.dtype), list(x.shape)))
        for name, leaf in graph_inputs.items():
            if name in known_inputs:
                continue
            if leaf is None:
                raise ValueError(
                    "graph input %s is not among the given inputs" % name)
            # a constant of the computation (e.g. of an imported graph)
            initializers.append(numpy_helper.from_array(
                tensor.to_numpy(leaf), name))
        g_out = [helper.make_tensor_value_info(
            yy.creator.output_name(yy.creator.y_id2idx[yy.uid]),
            _singa_dtype_to_onnx(yy.dtype), list(yy.shape)) for yy in ys]
        return helper.make_graph(nodes, model_name, g_in, g_out, initializers)

    @classmethod
    def singa_to_onnx_model(cls, inputs, y, model_name="sonnx"):
        graph = cls.singa_to_onnx_graph(inputs, y, model_name)
        return helper.make_model(
            graph, producer_name='sonnx',
            opset_imports=[helper.make_operatorsetid(
                "", cls._target_opset_version)], ir_version=_IR_VERSION)


run_node = SingaBackend.run_node
_run_node = SingaBackend._run_node
prepare = SingaBackend.prepare
get_op = SingaBackend._onnx_node_to_singa_op
to_onnx = SingaFrontend.singa_to_onnx_model
save = onnx.save
load = onnx.load
