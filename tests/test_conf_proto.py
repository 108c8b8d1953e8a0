"""Configuration messages of the C++ model API in protobuf's wire and text
formats (csrc/cpp/src/model/conf_proto.cc), checked against the protobuf
library itself: the messages of the reference's src/proto/model.proto are
rebuilt here as dynamic google.protobuf classes from a field table typed off
that file (and compared with the file when the reference tree is present), so
the encoder / decoder under test shares no code with its oracle."""
import ctypes
import os
import re

import pytest

# SINGA_CPP_LIBRARY: another build of the library (the sanitizer job's)
LIB = os.environ.get("SINGA_CPP_LIBRARY") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "singa_amd", "lib",
    "libsinga_amd_cpp.so")
MODEL_PROTO = "/root/reference/src/proto/model.proto"
# messages that live in the reference's other schema files
ELSEWHERE = {"EncoderConf": "io.proto", "DecoderConf": "io.proto",
             "TransformerConf": "io.proto", "ImageRecord": "io.proto",
             "MemPoolConf": "core.proto"}
PACKED = {("TransformerConf", "crop_shape")}

# message -> [(field, number, label, type[, message or enum type])]
O, R = "optional", "repeated"
SCHEMA = {
    "FillerConf": [("type", 1, O, "string"), ("value", 2, O, "float"),
                   ("min", 3, O, "float"), ("max", 4, O, "float"),
                   ("mean", 5, O, "float"), ("std", 6, O, "float")],
    "RegularizerConf": [("type", 1, O, "string"),
                        ("coefficient", 2, O, "float")],
    "ConstraintConf": [("type", 1, O, "string"), ("threshold", 2, O, "float")],
    "ParamSpec": [("name", 1, O, "string"), ("lr_mult", 3, O, "float"),
                  ("decay_mult", 4, O, "float"),
                  ("filler", 20, O, "message", "FillerConf"),
                  ("constraint", 21, O, "message", "ConstraintConf"),
                  ("regularizer", 22, O, "message", "RegularizerConf")],
    "OptimizerConf": [("type", 1, O, "string"), ("rho", 2, O, "float"),
                      ("momentum", 5, O, "float"), ("delta", 6, O, "float"),
                      ("regularizer", 10, O, "message", "RegularizerConf"),
                      ("constraint", 11, O, "message", "ConstraintConf")],
    "MetricConf": [("top_k", 1, O, "uint32")],
    "ConcatConf": [("concat_dim", 1, O, "uint32"), ("axis", 2, O, "int32")],
    "ConvolutionConf": [("num_output", 1, O, "uint32"),
                        ("bias_term", 2, O, "bool"), ("pad", 3, R, "uint32"),
                        ("kernel_size", 4, R, "uint32"),
                        ("stride", 6, R, "uint32"),
                        ("weight_filler", 7, O, "message", "FillerConf"),
                        ("pad_h", 9, O, "uint32"), ("pad_w", 10, O, "uint32"),
                        ("kernel_h", 11, O, "uint32"),
                        ("kernel_w", 12, O, "uint32"),
                        ("stride_h", 13, O, "uint32"),
                        ("stride_w", 14, O, "uint32"),
                        ("prefer", 51, O, "string")],
    "DropoutConf": [("dropout_ratio", 1, O, "float")],
    "DenseConf": [("num_output", 1, O, "uint32"), ("bias_term", 2, O, "bool"),
                  ("transpose", 21, O, "bool")],
    "LRNConf": [("local_size", 1, O, "uint32"), ("alpha", 2, O, "float"),
                ("beta", 3, O, "float"), ("k", 5, O, "float")],
    "PoolingConf": [("pool", 1, O, "enum", "PoolMethod"),
                    ("kernel_size", 2, O, "uint32"), ("stride", 3, O, "uint32"),
                    ("pad", 4, O, "uint32"), ("kernel_h", 5, O, "uint32"),
                    ("kernel_w", 6, O, "uint32"), ("stride_h", 7, O, "uint32"),
                    ("stride_w", 8, O, "uint32"), ("pad_h", 9, O, "uint32"),
                    ("pad_w", 10, O, "uint32"), ("ceil", 60, O, "bool")],
    "ReLUConf": [("negative_slope", 1, O, "float")],
    "SliceConf": [("slice_dim", 1, O, "uint32"),
                  ("slice_point", 2, R, "uint32"), ("axis", 3, O, "int32")],
    "PReLUConf": [("filler", 1, O, "message", "FillerConf"),
                  ("channel_shared", 2, O, "bool"), ("format", 20, O, "string")],
    "FlattenConf": [("axis", 1, O, "int32")],
    "RNNConf": [("hidden_size", 1, O, "uint32"), ("num_stacks", 2, O, "uint32"),
                ("dropout", 3, O, "float"), ("remember_state", 4, O, "bool"),
                ("input_mode", 7, O, "string"), ("direction", 8, O, "string"),
                ("rnn_mode", 9, O, "string")],
    "BatchNormConf": [("factor", 1, O, "double")],
    "SplitConf": [("output_size", 1, O, "int32")],
    "LayerConf": [("name", 1, O, "string"), ("type", 2, O, "string"),
                  ("param", 6, R, "message", "ParamSpec"),
                  ("concat_conf", 104, O, "message", "ConcatConf"),
                  ("convolution_conf", 106, O, "message", "ConvolutionConf"),
                  ("dropout_conf", 108, O, "message", "DropoutConf"),
                  ("dense_conf", 117, O, "message", "DenseConf"),
                  ("lrn_conf", 118, O, "message", "LRNConf"),
                  ("pooling_conf", 121, O, "message", "PoolingConf"),
                  ("relu_conf", 123, O, "message", "ReLUConf"),
                  ("slice_conf", 126, O, "message", "SliceConf"),
                  ("prelu_conf", 131, O, "message", "PReLUConf"),
                  ("flatten_conf", 135, O, "message", "FlattenConf"),
                  ("rnn_conf", 140, O, "message", "RNNConf"),
                  ("batchnorm_conf", 202, O, "message", "BatchNormConf"),
                  ("split_conf", 203, O, "message", "SplitConf")],
    "EncoderConf": [("type", 1, O, "string"),
                    ("image_dim_order", 2, O, "string")],
    "DecoderConf": [("type", 1, O, "string"),
                    ("image_dim_order", 2, O, "string"),
                    ("has_label", 3, O, "bool")],
    "TransformerConf": [("featurewise_center", 1, O, "bool"),
                        ("samplewise_center", 2, O, "bool"),
                        ("featurewise_std_norm", 3, O, "bool"),
                        ("samplewise_std_norm", 4, O, "bool"),
                        ("zca_whitening", 5, O, "bool"),
                        ("rotation_range", 6, O, "int32"),
                        ("crop_shape", 7, R, "uint32"),
                        ("resize_height", 8, O, "int32"),
                        ("resize_width", 9, O, "int32"),
                        ("horizontal_mirror", 10, O, "bool"),
                        ("vertical_mirror", 11, O, "bool"),
                        ("image_dim_order", 12, O, "string"),
                        ("rescale", 13, O, "float")],
    "ImageRecord": [("shape", 1, R, "int32"), ("label", 2, R, "int32"),
                    ("pixel", 3, O, "bytes")],
    "MemPoolConf": [("type", 1, O, "string"), ("init_size", 2, O, "uint32"),
                    ("max_size", 3, O, "uint32"), ("flag", 11, O, "uint32"),
                    ("device", 12, R, "uint32")],
}


@pytest.fixture(scope="module")
def messages():
    from google.protobuf import descriptor_pb2, descriptor_pool
    try:
        from google.protobuf.message_factory import GetMessageClass
    except ImportError:                     # older protobuf
        from google.protobuf import message_factory
        GetMessageClass = message_factory.MessageFactory().GetPrototype
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(
        name="singa_model_test.proto", package="singa_conf_test",
        syntax="proto2")
    en = fd.enum_type.add(name="PoolMethod")
    for i, nm in enumerate(["MAX", "AVE", "STOCHASTIC"]):
        en.value.add(name=nm, number=i)
    types = {"string": F.TYPE_STRING, "float": F.TYPE_FLOAT,
             "double": F.TYPE_DOUBLE, "uint32": F.TYPE_UINT32,
             "int32": F.TYPE_INT32, "bool": F.TYPE_BOOL,
             "bytes": F.TYPE_BYTES,
             "message": F.TYPE_MESSAGE, "enum": F.TYPE_ENUM}
    for mname, fields in SCHEMA.items():
        msg = fd.message_type.add(name=mname)
        for spec in fields:
            f = msg.field.add(
                name=spec[0], number=spec[1], type=types[spec[3]],
                label=F.LABEL_REPEATED if spec[2] == R else F.LABEL_OPTIONAL)
            if len(spec) > 4:
                f.type_name = ".singa_conf_test." + spec[4]
            if (mname, spec[0]) in PACKED:
                f.options.packed = True
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {m: GetMessageClass(pool.FindMessageTypeByName(
        "singa_conf_test." + m)) for m in SCHEMA}


@pytest.fixture(scope="module")
def transcode():
    if not os.path.exists(LIB):
        pytest.skip("C++ library not built")
    lib = ctypes.CDLL(LIB)
    fn = lib.singa_conf_transcode
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                   ctypes.c_int64, ctypes.c_int, ctypes.c_char_p,
                   ctypes.c_int64]
    lib.singa_last_error.restype = ctypes.c_char_p

    def run(message, data, src, dst):
        """src / dst: "wire" or "text"; raises ValueError on refusal."""
        if isinstance(data, str):
            data = data.encode()
        fmt = {"wire": 0, "text": 1}
        buf = ctypes.create_string_buffer(1 << 16)
        n = fn(message.encode(), fmt[src], data, len(data), fmt[dst], buf,
               len(buf))
        if n < 0:
            raise ValueError(lib.singa_last_error().decode())
        assert n <= len(buf)
        return buf.raw[:n] if dst == "wire" else buf.raw[:n].decode()
    return run


@pytest.mark.skipif(not os.path.exists(MODEL_PROTO),
                    reason="reference tree not present")
def test_field_table_matches_the_reference_schema():
    sources = {}
    for mname, fields in SCHEMA.items():
        path = os.path.join(os.path.dirname(MODEL_PROTO),
                            ELSEWHERE.get(mname, "model.proto"))
        if path not in sources:
            sources[path] = re.sub(r"//[^\n]*", "", open(path).read())
        body = re.search(r"message %s \{(.*?)\n\}" % mname, sources[path],
                         re.S).group(1)
        for spec in fields:
            name, number, label, typ = spec[:4]
            want = spec[4] if typ in ("message", "enum") else typ
            pat = r"%s\s+%s\s+%s\s*=\s*%d\b([^;]*);" % (label, want, name,
                                                        number)
            found = re.search(pat, body)
            assert found, (mname, spec)
            assert ("packed" in found.group(1)) == ((mname, name) in PACKED)
        # and no field of these small messages is left out of the table
        if mname in ELSEWHERE:
            assert len(re.findall(r"=\s*\d+", re.sub(r"\[[^\]]*\]", "",
                                                     body))) == len(fields)


def _conv_layer(M):
    m = M["LayerConf"](name="conv1", type="singacpp_convolution")
    p = m.param.add(name="w", lr_mult=0.5)
    p.filler.type = "gaussian"
    p.filler.std = 0.01
    p.regularizer.coefficient = 5e-4
    m.param.add(name="b", decay_mult=0.0).constraint.threshold = 3.0
    c = m.convolution_conf
    c.num_output, c.bias_term = 32, False
    c.kernel_h, c.kernel_w, c.pad_h, c.pad_w, c.stride_h, c.stride_w = \
        5, 3, 2, 1, 2, 2
    return m


def _cases(M):
    out = [("LayerConf", _conv_layer(M))]
    m = M["LayerConf"](name="pool", type="singacpp_pooling")
    m.pooling_conf.pool = 1
    m.pooling_conf.kernel_h = m.pooling_conf.kernel_w = 3
    m.pooling_conf.stride_h = m.pooling_conf.stride_w = 2
    out.append(("LayerConf", m))
    m = M["LayerConf"](name="misc \"quoted\"\n\\ \x01\xc3\xa9", type="x")
    m.dense_conf.num_output, m.dense_conf.transpose = 10, True
    m.dropout_conf.dropout_ratio = 0.3
    m.lrn_conf.local_size, m.lrn_conf.alpha, m.lrn_conf.k = 3, 1e-4, 2.0
    m.relu_conf.negative_slope = 0.01
    m.slice_conf.slice_point.extend([2, 5, 7])
    m.slice_conf.axis = 2
    m.concat_conf.axis = 3
    m.flatten_conf.axis = -1
    m.prelu_conf.channel_shared = True
    m.prelu_conf.format = "NHWC"
    m.prelu_conf.filler.value = 0.25
    m.rnn_conf.hidden_size, m.rnn_conf.num_stacks = 64, 2
    m.rnn_conf.rnn_mode, m.rnn_conf.direction = "lstm", "bidirectional"
    m.rnn_conf.dropout, m.rnn_conf.remember_state = 0.1, True
    m.batchnorm_conf.factor = 0.123456789012345
    m.split_conf.output_size = 3
    out.append(("LayerConf", m))
    o = M["OptimizerConf"](type="rmsprop", rho=0.9, momentum=0.9, delta=1e-6)
    o.regularizer.coefficient = 1e-4
    o.constraint.threshold = 5.0     # a field set to its default has no
    # counterpart in the records, so none is set here
    out.append(("OptimizerConf", o))
    out.append(("FillerConf", M["FillerConf"](type="uniform", min=-0.05,
                                              max=0.05)))
    out.append(("MetricConf", M["MetricConf"](top_k=5)))
    out.append(("ParamSpec", M["ParamSpec"](name="w", lr_mult=2.0)))
    out.append(("EncoderConf", M["EncoderConf"](type="csv",
                                                image_dim_order="CHW")))
    out.append(("DecoderConf", M["DecoderConf"](image_dim_order="HWC",
                                                has_label=False)))
    t = M["TransformerConf"](
        featurewise_center=True, samplewise_std_norm=True, zca_whitening=True,
        rotation_range=-15, resize_height=256, resize_width=320,
        horizontal_mirror=True, image_dim_order="HWC", rescale=1 / 255.)
    t.crop_shape.extend([224, 227])
    out.append(("TransformerConf", t))
    r = M["ImageRecord"](pixel=bytes(range(256)) + b"\xff\xd8 \"'\\\n")
    r.shape.extend([3, 32, 32])
    r.label.extend([7, -1])
    out.append(("ImageRecord", r))
    mp = M["MemPoolConf"](type="pool", init_size=1024, max_size=4096, flag=2)
    mp.device.extend([0, 3])
    out.append(("MemPoolConf", mp))
    return out


def test_wire_bytes_equal_the_protobuf_library(messages, transcode):
    for name, msg in _cases(messages):
        blob = msg.SerializeToString()
        # decode + encode here gives the library's bytes back
        assert transcode(name, blob, "wire", "wire") == blob, name


def test_text_output_is_read_by_the_protobuf_library(messages, transcode):
    from google.protobuf import text_format
    for name, msg in _cases(messages):
        text = transcode(name, msg.SerializeToString(), "wire", "text")
        back = messages[name]()
        text_format.Parse(text, back)
        assert back == msg, (name, text)


def test_text_of_the_protobuf_library_is_read(messages, transcode):
    from google.protobuf import text_format
    for name, msg in _cases(messages):
        for one_line in (False, True):
            text = text_format.MessageToString(msg, as_one_line=one_line)
            assert transcode(name, text, "text", "wire") == \
                msg.SerializeToString(), (name, text)


def test_fields_outside_the_records_are_skipped(messages, transcode):
    m = _conv_layer(messages)
    plain = m.SerializeToString()
    m.convolution_conf.weight_filler.type = "xavier"
    m.convolution_conf.prefer = "fastest"
    assert transcode("LayerConf", m.SerializeToString(), "wire", "wire") == \
        plain
    from google.protobuf import text_format
    assert transcode("LayerConf", text_format.MessageToString(m), "text",
                     "wire") == plain


def test_legacy_extent_fields_set_both_sides(messages, transcode):
    m = messages["LayerConf"](name="c")
    m.convolution_conf.kernel_size.append(3)
    m.convolution_conf.pad.extend([1, 2])
    m.convolution_conf.stride.append(2)
    want = messages["LayerConf"](name="c")
    c = want.convolution_conf
    c.kernel_h = c.kernel_w = 3
    c.pad_h, c.pad_w = 1, 2
    c.stride_h = c.stride_w = 2
    assert transcode("LayerConf", m.SerializeToString(), "wire", "wire") == \
        want.SerializeToString()
    assert transcode("LayerConf", "name: 'c' convolution_conf { kernel_size: 3"
                     " pad: [1, 2] stride: 2 }", "text", "wire") == \
        want.SerializeToString()
    p = messages["LayerConf"]()
    p.pooling_conf.kernel_size, p.pooling_conf.stride = 2, 2
    text = transcode("LayerConf", p.SerializeToString(), "wire", "text")
    assert "kernel_h: 2" in text and "kernel_w: 2" in text and \
        "stride_w: 2" in text


@pytest.mark.parametrize("blob", [
    b"\x0a\x05ab", b"\x0a", b"\xff" * 11, b"\x0b\x00", b"\x0d\x00\x00",
    b"\x09\x00", b"\x32\x03\x0a\x05a"])
def test_malformed_wire_is_refused(transcode, blob):
    with pytest.raises(ValueError):
        transcode("LayerConf", blob, "wire", "wire")


@pytest.mark.parametrize("text,where", [
    ("name: \"open", "line 1"), ("name: 'a'\n\ndense_conf { num_output: q }",
                                 "line 3"),
    ("dense_conf { num_output: 1", "bracket"), ("name 'a'", "':'"),
    ("dense_conf: 3", "'{'"), ("} name: 'a'", "field name"),
    ("pooling_conf { pool: MEDIAN }", "pool")])
def test_malformed_text_is_refused_with_a_position(transcode, text, where):
    with pytest.raises(ValueError) as e:
        transcode("LayerConf", text, "text", "wire")
    assert where in str(e.value)


def test_unknown_message_name(transcode):
    with pytest.raises(ValueError):
        transcode("NetConf", b"", "wire", "wire")


def test_crop_shape_is_read_packed_or_not(messages, transcode):
    want = messages["TransformerConf"]()
    want.crop_shape.extend([24, 32])
    # the same two numbers as separate varint fields (an older writer)
    loose = b"\x38\x18\x38\x20"
    assert transcode("TransformerConf", loose, "wire", "wire") == \
        want.SerializeToString() == b"\x3a\x02\x18\x20"
    for text in ("crop_shape: 24 crop_shape: 32", "crop_shape: [24, 32]"):
        assert transcode("TransformerConf", text, "text", "wire") == \
            want.SerializeToString()
