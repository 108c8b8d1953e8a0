"""The ONNX wire format (IR version 6 / opset 11 subset of onnx.proto) without
the ``onnx`` package: message classes are generated at import time from a
schema written out below with protobuf's descriptor API, so files written
here are read by any ONNX tool and vice versa.

Provides what the reference's sonnx.py takes from ``onnx``: the message
classes, ``helper`` (make_node / make_tensor / make_graph / make_model /
make_tensor_value_info / get_attribute_value), ``numpy_helper`` (to_array /
from_array), ``mapping`` tables, ``load`` / ``save``
(reference python/singa/sonnx.py:22-40).
"""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_T = {"int64": _F.TYPE_INT64, "int32": _F.TYPE_INT32, "float": _F.TYPE_FLOAT,
      "double": _F.TYPE_DOUBLE, "uint64": _F.TYPE_UINT64,
      "string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES}

# message -> [(name, number, type, repeated, packed)]; a type that is not a
# scalar names another message of this schema
_SCHEMA = {
    "AttributeProto": [
        ("name", 1, "string", 0, 0), ("ref_attr_name", 21, "string", 0, 0),
        ("doc_string", 13, "string", 0, 0), ("type", 20, "int32", 0, 0),
        ("f", 2, "float", 0, 0), ("i", 3, "int64", 0, 0),
        ("s", 4, "bytes", 0, 0), ("t", 5, "TensorProto", 0, 0),
        ("g", 6, "GraphProto", 0, 0), ("floats", 7, "float", 1, 0),
        ("ints", 8, "int64", 1, 0), ("strings", 9, "bytes", 1, 0),
        ("tensors", 10, "TensorProto", 1, 0),
        ("graphs", 11, "GraphProto", 1, 0)],
    "ValueInfoProto": [
        ("name", 1, "string", 0, 0), ("type", 2, "TypeProto", 0, 0),
        ("doc_string", 3, "string", 0, 0)],
    "NodeProto": [
        ("input", 1, "string", 1, 0), ("output", 2, "string", 1, 0),
        ("name", 3, "string", 0, 0), ("op_type", 4, "string", 0, 0),
        ("domain", 7, "string", 0, 0),
        ("attribute", 5, "AttributeProto", 1, 0),
        ("doc_string", 6, "string", 0, 0)],
    "ModelProto": [
        ("ir_version", 1, "int64", 0, 0),
        ("opset_import", 8, "OperatorSetIdProto", 1, 0),
        ("producer_name", 2, "string", 0, 0),
        ("producer_version", 3, "string", 0, 0),
        ("domain", 4, "string", 0, 0), ("model_version", 5, "int64", 0, 0),
        ("doc_string", 6, "string", 0, 0), ("graph", 7, "GraphProto", 0, 0),
        ("metadata_props", 14, "StringStringEntryProto", 1, 0)],
    "StringStringEntryProto": [
        ("key", 1, "string", 0, 0), ("value", 2, "string", 0, 0)],
    "GraphProto": [
        ("node", 1, "NodeProto", 1, 0), ("name", 2, "string", 0, 0),
        ("initializer", 5, "TensorProto", 1, 0),
        ("doc_string", 10, "string", 0, 0),
        ("input", 11, "ValueInfoProto", 1, 0),
        ("output", 12, "ValueInfoProto", 1, 0),
        ("value_info", 13, "ValueInfoProto", 1, 0)],
    "TensorProto": [
        ("dims", 1, "int64", 1, 0), ("data_type", 2, "int32", 0, 0),
        ("float_data", 4, "float", 1, 1), ("int32_data", 5, "int32", 1, 1),
        ("string_data", 6, "bytes", 1, 0), ("int64_data", 7, "int64", 1, 1),
        ("name", 8, "string", 0, 0), ("doc_string", 12, "string", 0, 0),
        ("raw_data", 9, "bytes", 0, 0), ("double_data", 10, "double", 1, 1),
        ("uint64_data", 11, "uint64", 1, 1)],
    "TensorShapeProto": [("dim", 1, "TensorShapeProto.Dimension", 1, 0)],
    "TensorShapeProto.Dimension": [
        ("dim_value", 1, "int64", 0, 0), ("dim_param", 2, "string", 0, 0),
        ("denotation", 3, "string", 0, 0)],
    "TypeProto": [("tensor_type", 1, "TypeProto.Tensor", 0, 0),
                  ("denotation", 6, "string", 0, 0)],
    "TypeProto.Tensor": [("elem_type", 1, "int32", 0, 0),
                         ("shape", 2, "TensorShapeProto", 0, 0)],
    "OperatorSetIdProto": [("domain", 1, "string", 0, 0),
                           ("version", 2, "int64", 0, 0)],
}


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "singa_amd_onnx.proto"
    fd.package = "onnx"
    fd.syntax = "proto2"
    top = {}
    for full in _SCHEMA:
        if "." not in full:
            top[full] = fd.message_type.add()
            top[full].name = full
    msgs = dict(top)
    for full in _SCHEMA:
        if "." in full:
            parent, child = full.split(".")
            m = top[parent].nested_type.add()
            m.name = child
            msgs[full] = m
    for full, fields in _SCHEMA.items():
        for (name, num, typ, rep, packed) in fields:
            f = msgs[full].field.add()
            f.name = name
            f.number = num
            f.label = _F.LABEL_REPEATED if rep else _F.LABEL_OPTIONAL
            if typ in _T:
                f.type = _T[typ]
                if packed:
                    f.options.packed = True
            else:
                f.type = _F.TYPE_MESSAGE
                f.type_name = ".onnx." + typ
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    out = {}
    for full in _SCHEMA:
        desc = pool.FindMessageTypeByName("onnx." + full)
        out[full] = message_factory.GetMessageClass(desc)
    return out


_CLS = _build()
AttributeProto = _CLS["AttributeProto"]
ValueInfoProto = _CLS["ValueInfoProto"]
NodeProto = _CLS["NodeProto"]
ModelProto = _CLS["ModelProto"]
GraphProto = _CLS["GraphProto"]
TensorProto = _CLS["TensorProto"]
TensorShapeProto = _CLS["TensorShapeProto"]
TypeProto = _CLS["TypeProto"]
OperatorSetIdProto = _CLS["OperatorSetIdProto"]


class DataType(object):
    UNDEFINED, FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, \
        FLOAT16, DOUBLE, UINT32, UINT64 = range(14)
    BFLOAT16 = 16


for _n in ("UNDEFINED FLOAT UINT8 INT8 UINT16 INT16 INT32 INT64 STRING BOOL "
           "FLOAT16 DOUBLE UINT32 UINT64 BFLOAT16").split():
    setattr(TensorProto, _n, getattr(DataType, _n))


class AttrType(object):
    UNDEFINED, FLOAT, INT, STRING, TENSOR, GRAPH, FLOATS, INTS, STRINGS, \
        TENSORS, GRAPHS = range(11)


for _n in ("UNDEFINED FLOAT INT STRING TENSOR GRAPH FLOATS INTS STRINGS "
           "TENSORS GRAPHS").split():
    setattr(AttributeProto, _n, getattr(AttrType, _n))


class mapping(object):
    TENSOR_TYPE_TO_NP_TYPE = {
        DataType.FLOAT: np.dtype("float32"), DataType.UINT8: np.dtype("uint8"),
        DataType.INT8: np.dtype("int8"), DataType.UINT16: np.dtype("uint16"),
        DataType.INT16: np.dtype("int16"), DataType.INT32: np.dtype("int32"),
        DataType.INT64: np.dtype("int64"), DataType.BOOL: np.dtype("bool"),
        DataType.FLOAT16: np.dtype("float16"),
        DataType.DOUBLE: np.dtype("float64"),
        DataType.UINT32: np.dtype("uint32"),
        DataType.UINT64: np.dtype("uint64"),
    }
    NP_TYPE_TO_TENSOR_TYPE = {v: k for k, v in TENSOR_TYPE_TO_NP_TYPE.items()}


class numpy_helper(object):

    @staticmethod
    def to_array(t):
        dt = mapping.TENSOR_TYPE_TO_NP_TYPE[t.data_type]
        dims = list(t.dims)
        if t.raw_data:
            return np.frombuffer(t.raw_data, dtype=dt).reshape(dims).copy()
        if t.data_type == DataType.FLOAT:
            data = np.asarray(t.float_data, dtype=dt)
        elif t.data_type == DataType.DOUBLE:
            data = np.asarray(t.double_data, dtype=dt)
        elif t.data_type == DataType.INT64:
            data = np.asarray(t.int64_data, dtype=dt)
        elif t.data_type in (DataType.UINT32, DataType.UINT64):
            data = np.asarray(t.uint64_data, dtype=dt)
        elif t.data_type == DataType.FLOAT16:
            data = np.asarray(t.int32_data, dtype=np.uint16).view(np.float16)
        else:  # the small integer types and bool travel in int32_data
            data = np.asarray(t.int32_data, dtype=np.int32).astype(dt)
        return data.reshape(dims)

    @staticmethod
    def from_array(a, name=None):
        a = np.asarray(a)
        t = TensorProto()
        t.dims.extend(a.shape)
        if name:
            t.name = name
        t.data_type = mapping.NP_TYPE_TO_TENSOR_TYPE[a.dtype]
        t.raw_data = np.ascontiguousarray(a).tobytes()
        return t


class helper(object):

    @staticmethod
    def make_attribute(key, value):
        a = AttributeProto()
        a.name = key
        if isinstance(value, float):
            a.f, a.type = value, AttrType.FLOAT
        elif isinstance(value, (bool, int, np.integer)):
            a.i, a.type = int(value), AttrType.INT
        elif isinstance(value, (str, bytes)):
            a.s = value.encode("utf-8") if isinstance(value, str) else value
            a.type = AttrType.STRING
        elif isinstance(value, TensorProto):
            a.t.CopyFrom(value)
            a.type = AttrType.TENSOR
        elif isinstance(value, GraphProto):
            a.g.CopyFrom(value)
            a.type = AttrType.GRAPH
        elif isinstance(value, (list, tuple, np.ndarray)):
            vals = list(value)
            if all(isinstance(v, (bool, int, np.integer)) for v in vals):
                a.ints.extend(int(v) for v in vals)
                a.type = AttrType.INTS
            elif all(isinstance(v, (float, int, np.floating, np.integer))
                     for v in vals):
                a.floats.extend(float(v) for v in vals)
                a.type = AttrType.FLOATS
            elif all(isinstance(v, (str, bytes)) for v in vals):
                a.strings.extend(v.encode("utf-8") if isinstance(v, str) else v
                                 for v in vals)
                a.type = AttrType.STRINGS
            else:
                raise ValueError("unsupported attribute list %r" % (value,))
        else:
            raise ValueError("unsupported attribute value %r" % (value,))
        return a

    @staticmethod
    def get_attribute_value(a):
        t = a.type
        if t == AttrType.FLOAT:
            return a.f
        if t == AttrType.INT:
            return a.i
        if t == AttrType.STRING:
            return a.s
        if t == AttrType.TENSOR:
            return a.t
        if t == AttrType.GRAPH:
            return a.g
        if t == AttrType.FLOATS:
            return list(a.floats)
        if t == AttrType.INTS:
            return list(a.ints)
        if t == AttrType.STRINGS:
            return list(a.strings)
        if t == AttrType.TENSORS:
            return list(a.tensors)
        if t == AttrType.GRAPHS:
            return list(a.graphs)
        raise ValueError("attribute %s has no value" % a.name)

    @staticmethod
    def make_node(op_type, inputs, outputs, name=None, **attrs):
        n = NodeProto()
        n.op_type = op_type
        n.input.extend(inputs)
        n.output.extend(outputs)
        if name:
            n.name = name
        for k in sorted(attrs):
            if attrs[k] is not None:
                n.attribute.add().CopyFrom(helper.make_attribute(k, attrs[k]))
        return n

    @staticmethod
    def make_tensor(name, data_type, dims, vals, raw=False):
        t = TensorProto()
        t.name = name
        t.data_type = data_type
        t.dims.extend(dims)
        if raw:
            t.raw_data = vals
        else:
            a = np.asarray(vals, dtype=mapping.TENSOR_TYPE_TO_NP_TYPE[
                data_type]).reshape(-1)
            t.raw_data = a.tobytes()
        return t

    @staticmethod
    def make_tensor_value_info(name, elem_type, shape):
        v = ValueInfoProto()
        v.name = name
        v.type.tensor_type.elem_type = elem_type
        if shape is not None:
            v.type.tensor_type.shape.SetInParent()
            for d in shape:
                dim = v.type.tensor_type.shape.dim.add()
                if isinstance(d, str):
                    dim.dim_param = d
                elif d is not None:
                    dim.dim_value = int(d)
        return v

    @staticmethod
    def make_graph(nodes, name, inputs, outputs, initializer=None):
        g = GraphProto()
        g.name = name
        g.node.extend(nodes)
        g.input.extend(inputs)
        g.output.extend(outputs)
        if initializer:
            g.initializer.extend(initializer)
        return g

    @staticmethod
    def make_operatorsetid(domain, version):
        o = OperatorSetIdProto()
        o.domain = domain
        o.version = version
        return o

    @staticmethod
    def make_model(graph, producer_name="singa_amd", opset_imports=None,
                   ir_version=6, **kw):
        m = ModelProto()
        m.ir_version = ir_version
        m.producer_name = producer_name
        m.graph.CopyFrom(graph)
        if opset_imports is None:
            opset_imports = [helper.make_operatorsetid("", 11)]
        m.opset_import.extend(opset_imports)
        for k, v in kw.items():
            setattr(m, k, v)
        return m


def load(path_or_bytes):
    """Parse a ModelProto from a file path, a file object or bytes."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = bytes(path_or_bytes)
    elif hasattr(path_or_bytes, "read"):
        data = path_or_bytes.read()
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    m = ModelProto()
    m.ParseFromString(data)
    return m


load_model = load


def save(model, path):
    data = model.SerializeToString()
    if hasattr(path, "write"):
        path.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


save_model = save
