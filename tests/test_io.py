"""Native runtime utilities: record files, snapshot format, channels, timer
and logging (reference test/singa/test_binfile_rw.cc, test_textfile_rw.cc,
test_snapshot.cc, test_channel.cc, test_timer.cc, test_logging.cc)."""
import os
import struct
import time

import numpy as np
import pytest

from singa_amd import _rt, snapshot, tensor


def test_binfile_roundtrip(tmp_path):
    path = os.path.join(str(tmp_path), "a.bin")
    w = _rt.BinFileWriter()
    assert w.Open(path)
    assert w.Write("key1", b"hello")
    assert w.Write("", b"\x00\x01\x02")
    assert w.Write("key3", b"x" * 100000)
    w.Flush()
    w.Close()
    r = _rt.BinFileReader()
    assert r.Open(path)
    assert r.Count() == 3
    assert r.Read() == ("key1", b"hello")
    assert r.Read() == ("", b"\x00\x01\x02")
    k, v = r.Read()
    assert k == "key3" and v == b"x" * 100000
    assert r.Read() is None
    r.SeekToFirst()
    assert r.Read()[0] == "key1"
    r.Close()
    # on-disk framing: magic 's' 'g', key flag, reserved byte, sizes as u64
    raw = open(path, "rb").read()
    assert raw[:2] == b"sg" and raw[2] == 1
    assert struct.unpack("<Q", raw[4:12])[0] == 4 and raw[12:16] == b"key1"


def test_binfile_append(tmp_path):
    path = os.path.join(str(tmp_path), "b.bin")
    w = _rt.BinFileWriter()
    w.Open(path)
    w.Write("a", b"1")
    w.Close()
    w = _rt.BinFileWriter()
    w.Open(path, True)
    w.Write("b", b"2")
    w.Close()
    r = _rt.BinFileReader()
    r.Open(path)
    assert r.Count() == 2


def test_textfile_roundtrip(tmp_path):
    path = os.path.join(str(tmp_path), "a.txt")
    w = _rt.TextFileWriter()
    assert w.Open(path)
    w.Write("", "first line")
    w.Write("", "second line")
    w.Close()
    r = _rt.TextFileReader()
    assert r.Open(path)
    assert r.Count() == 2
    assert r.Read() == ("0", "first line")
    assert r.Read() == ("1", "second line")
    assert r.Read() is None
    r.Close()


def test_snapshot_roundtrip(tmp_path):
    prefix = os.path.join(str(tmp_path), "snap")
    a = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5
    b = np.arange(5, dtype=np.int32)
    s = snapshot.Snapshot(prefix, True)
    s.write("Param_1", tensor.from_numpy(a))
    s.write("Param_2", tensor.from_numpy(b))
    s.close()
    assert os.path.isfile(prefix + ".bin") and os.path.isfile(prefix + ".desc")
    desc = open(prefix + ".desc").read()
    assert "SINGA VERSION" in desc and "Param_1" in desc
    r = snapshot.Snapshot(prefix, False)
    got = r.read()
    assert sorted(got) == ["Param_1", "Param_2"]
    np.testing.assert_array_equal(tensor.to_numpy(got["Param_1"]), a)
    np.testing.assert_array_equal(tensor.to_numpy(got["Param_2"]), b)


def test_tensor_proto_wire_format():
    """TensorProto fields (reference src/proto/core.proto:70-79, proto2):
    1 shape (repeated uint32, one varint per entry), 2 data_type, 3 stride,
    4 float_data (packed)."""
    st = _rt.SnapshotTensor()
    st.shape = [2, 1]
    st.dtype = 0
    st.data = np.array([1.0, 2.0], np.float32).tobytes()
    blob = _rt.EncodeTensorProto(st)
    assert blob[:4] == bytes([0x08, 0x02, 0x08, 0x01])
    assert bytes([0x22, 0x08]) in blob  # field 4, length-delimited, 8 bytes
    back = _rt.DecodeTensorProto(blob)
    assert list(back.shape) == [2, 1] and back.dtype == 0
    assert back.data == st.data


def _reference_tensor_proto():
    """The TensorProto message of reference src/proto/core.proto:26-34, 70-79
    (proto2) as a dynamic google.protobuf class: an encoder / decoder that
    shares no code with the runtime's hand-written wire codec."""
    from google.protobuf import descriptor_pb2, descriptor_pool
    try:
        from google.protobuf.message_factory import GetMessageClass
    except ImportError:                     # older protobuf
        from google.protobuf import message_factory
        GetMessageClass = message_factory.MessageFactory().GetPrototype
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(
        name="singa_core_test.proto", package="singa_test", syntax="proto2")
    en = fd.enum_type.add(name="DataType")
    for i, nm in enumerate(["kFloat32", "kFloat16", "kInt", "kChar", "kDouble",
                            "kUChar", "kNumDataType"]):
        en.value.add(name=nm, number=i)
    msg = fd.message_type.add(name="TensorProto")
    rep, opt = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    for name, num, label, typ, packed in [
            ("shape", 1, rep, F.TYPE_UINT32, False),
            ("data_type", 2, opt, F.TYPE_ENUM, False),
            ("stride", 3, rep, F.TYPE_INT32, False),
            ("float_data", 4, rep, F.TYPE_FLOAT, True),
            ("double_data", 5, rep, F.TYPE_DOUBLE, True),
            ("int_data", 6, rep, F.TYPE_INT32, True),
            ("bytes_data", 7, rep, F.TYPE_BYTES, False)]:
        f = msg.field.add(name=name, number=num, label=label, type=typ)
        if typ == F.TYPE_ENUM:
            f.type_name = ".singa_test.DataType"
        if packed:
            f.options.packed = True
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return GetMessageClass(pool.FindMessageTypeByName("singa_test.TensorProto"))


@pytest.mark.parametrize("dtype,np_dtype,field", [
    (0, np.float32, "float_data"), (2, np.int32, "int_data"),
    (4, np.float64, "double_data")])
def test_tensor_proto_against_protobuf_library(dtype, np_dtype, field):
    """Bytes written by the runtime parse with google.protobuf into the
    reference's message (same shape / type / values, nothing unknown left
    over), and bytes serialised by google.protobuf decode with the runtime."""
    TensorProto = _reference_tensor_proto()
    vals = (np.arange(24).reshape(2, 3, 4) * 3 - 7).astype(np_dtype)
    st = _rt.SnapshotTensor()
    st.shape = list(vals.shape)
    st.dtype = dtype
    st.data = vals.tobytes()
    blob = _rt.EncodeTensorProto(st)
    msg = TensorProto()
    msg.ParseFromString(blob)
    assert list(msg.shape) == [2, 3, 4]
    assert msg.data_type == dtype
    np.testing.assert_array_equal(
        np.array(getattr(msg, field), np_dtype), vals.reshape(-1))
    for other in ("float_data", "double_data", "int_data", "bytes_data"):
        if other != field:
            assert len(getattr(msg, other)) == 0
    # nothing the message does not know: re-serialising gives the same bytes
    assert msg.SerializeToString() == blob
    # the other direction, with strides present as the reference writes them
    ref = TensorProto()
    ref.shape.extend([4, 6])
    ref.data_type = dtype
    ref.stride.extend([6, 1])
    getattr(ref, field).extend(vals.reshape(-1).tolist())
    back = _rt.DecodeTensorProto(ref.SerializeToString())
    assert list(back.shape) == [4, 6] and back.dtype == dtype
    np.testing.assert_array_equal(np.frombuffer(back.data, np_dtype),
                                  vals.reshape(-1))


def test_channel(tmp_path):
    _rt.InitChannel("")
    _rt.SetChannelDirectory(str(tmp_path))
    ch = _rt.GetChannel("test_channel")
    assert ch.name() == "test_channel"
    ch.EnableDestFile(True)
    ch.Send("message one")
    ch.Send("message two")
    assert _rt.GetChannel("test_channel") is not None
    text = open(os.path.join(str(tmp_path), "test_channel")).read()
    assert "message one" in text and "message two" in text


def test_timer():
    t = _rt.Timer()
    time.sleep(0.05)
    assert 30 <= t.ElapsedMs() <= 500
    t.Tick()
    assert t.ElapsedMs() < 30


def test_logging(tmp_path, capfd):
    _rt.InitLogging("test")
    _rt.LogToStderr()
    _rt.Log(0, "info message", "file.py", 10)
    _rt.Log(1, "warning message", "file.py", 11)
    err = capfd.readouterr().err
    assert "info message" in err and "warning message" in err
    with pytest.raises(RuntimeError):
        _rt.Log(3, "fatal message", "file.py", 12)


# ---- LMDB environments -------------------------------------------------------
def _lmdb_walk(raw):
    """Independent parser of an LMDB data file (format version 1): returns the
    (key, value) pairs of the main database in tree order plus the database
    record of the current meta page."""
    def meta(off):
        flags, = struct.unpack_from("<H", raw, off + 10)
        assert flags & 0x08
        magic, version = struct.unpack_from("<II", raw, off + 16)
        assert magic == 0xBEEFC0DE and version == 1
        psize, = struct.unpack_from("<I", raw, off + 16 + 24)
        main = struct.unpack_from("<IHHQQQQQ", raw, off + 16 + 72)
        last_pg, txnid = struct.unpack_from("<QQ", raw, off + 16 + 120)
        return psize, main, last_pg, txnid
    psize = meta(0)[0]
    m0, m1 = meta(0), meta(psize)
    cur = m1 if m1[3] > m0[3] else m0
    _, main, last_pg, _ = cur
    assert (last_pg + 1) * psize == len(raw)
    out = []
    counts = {"branch": 0, "leaf": 0, "overflow": 0}

    def walk(pgno, level):
        off = pgno * psize
        no, _, flags, lower, upper = struct.unpack_from("<QHHHH", raw, off)
        assert no == pgno
        nkeys = (lower - 16) // 2
        ptrs = struct.unpack_from("<%dH" % nkeys, raw, off + 16)
        assert min(ptrs) == upper and lower <= upper
        depth = None
        for i, p in enumerate(ptrs):
            lo, hi, nflags, ksize = struct.unpack_from("<HHHH", raw, off + p)
            key = raw[off + p + 8: off + p + 8 + ksize]
            if flags & 0x01:
                assert (ksize == 0) == (i == 0)
                d = walk(lo | (hi << 16) | (nflags << 32), level + 1)
                assert depth in (None, d)
                depth = d
            else:
                assert flags & 0x02
                dsize = lo | (hi << 16)
                if nflags & 0x01:
                    ov, = struct.unpack_from("<Q", raw, off + p + 8 + ksize)
                    ono, _, oflags, pages = struct.unpack_from(
                        "<QHHI", raw, ov * psize)
                    assert ono == ov and oflags & 0x04
                    assert pages == (15 + dsize) // psize + 1
                    counts["overflow"] += pages
                    val = raw[ov * psize + 16: ov * psize + 16 + dsize]
                else:
                    val = raw[off + p + 8 + ksize: off + p + 8 + ksize + dsize]
                out.append((bytes(key), bytes(val)))
        if flags & 0x01:
            assert nkeys >= 2
            counts["branch"] += 1
            return depth + 1
        counts["leaf"] += 1
        return 1
    root = main[7]
    depth = walk(root, 0) if root != 2 ** 64 - 1 else 0
    assert depth == main[2]
    assert (counts["branch"], counts["leaf"], counts["overflow"]) == main[3:6]
    assert main[6] == len(out)
    return out, main


@pytest.mark.parametrize("page_size", [4096, 512])
def test_lmdb_file_format(tmp_path, page_size):
    rng = np.random.RandomState(3)
    d = os.path.join(str(tmp_path), "env")
    items = {}
    for i in rng.permutation(1500):
        key = ("k%07d" % i).encode() + b"x" * int(rng.randint(0, 40))
        items[key] = rng.bytes(int(rng.randint(1, 300)))
    items[b"big-one"] = rng.bytes(3 * page_size + 17)
    items[b"big-two"] = rng.bytes(page_size - 16)  # exactly one overflow page
    items[b"k"] = b""                                # a prefix of other keys
    w = _rt.LMDBWriter()
    assert w.Open(d, False, page_size)
    for k, v in items.items():
        assert w.Write(k, v)
    w.Close()
    raw = open(os.path.join(d, "data.mdb"), "rb").read()
    pairs, main = _lmdb_walk(raw)
    assert [k for k, _ in pairs] == sorted(items)
    assert dict(pairs) == items
    assert main[2] >= 2          # more than one level
    r = _rt.LMDBReader()
    assert r.Open(d)
    assert r.Count() == len(items) and r.page_size() == page_size
    got = []
    while True:
        kv = r.Read()
        if kv is None:
            break
        got.append(kv)
    assert got == pairs
    r.SeekToFirst()
    assert r.Read() == pairs[0]
    r.Close()


def test_lmdb_append_overwrite_and_empty(tmp_path):
    d = os.path.join(str(tmp_path), "env")
    w = _rt.LMDBWriter()
    assert w.Open(d)
    w.Close()                          # an environment without tuples
    r = _rt.LMDBReader()
    assert r.Open(d) and r.Count() == 0 and r.Read() is None
    r.Close()
    assert not _rt.LMDBWriter().Open(os.path.join(str(tmp_path), "no"), True)
    w = _rt.LMDBWriter()
    assert w.Open(d, True)
    w.Write(b"b", b"1")
    w.Write(b"a", b"2")
    w.Flush()
    w.Write(b"b", b"3")                # the last value of a key wins
    w.Close()
    w = _rt.LMDBWriter()
    assert w.Open(d, True)
    w.Write(b"c", b"4")
    w.Close()
    pairs, _ = _lmdb_walk(open(os.path.join(d, "data.mdb"), "rb").read())
    assert pairs == [(b"a", b"2"), (b"b", b"3"), (b"c", b"4")]
    with pytest.raises(RuntimeError):
        w2 = _rt.LMDBWriter()
        w2.Open(d, True)
        w2.Write(b"", b"x")
    assert not _rt.LMDBReader().Open(os.path.join(str(tmp_path), "missing"))


def test_python_io_package_round_trips(tmp_path):
    from singa_amd import io
    items = [("k%03d" % i, bytes([i % 251]) * (i + 1)) for i in range(40)]
    for name in ("a.bin", "env.lmdb"):
        path = os.path.join(str(tmp_path), name)
        with io.open_writer(path) as w:
            for k, v in reversed(items):
                assert w.write(k, v)
        with io.open_reader(path) as r:
            assert r.count() == len(items)
            got = list(r)
            # record files keep the write order, LMDB sorts by key
            assert got == (items if name.endswith("lmdb") else items[::-1])
            r.seek_to_first()
            assert r.read() == got[0]
        with io.open_writer(path, io.kAppend) as w:
            w.write("zzz", b"tail")
        assert io.open_reader(path).count() == len(items) + 1
    txt = os.path.join(str(tmp_path), "lines.txt")
    with io.open_writer(txt) as w:
        w.write("", "first")
        w.write("", b"second")
    assert list(io.open_reader(txt)) == [("0", "first"), ("1", "second")]
    with pytest.raises(IOError):
        io.open_reader(os.path.join(str(tmp_path), "missing.bin"))
