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
