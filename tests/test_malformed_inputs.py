"""Parsers of bytes that come from outside the process -- record files, LMDB
environments, snapshot blobs, JPEG files -- given damaged input: every
truncation point and a sweep of corrupted bytes.  A parser may report failure
(False / None / an exception) or return data, but it must not read or write
outside its buffers; tool/ci/sanitize.sh runs this file on an AddressSanitizer
/ UBSan build of the runtime, where any such access aborts the test.  (The
reference has neither sanitizer jobs nor malformed-input tests.)"""
import os
import subprocess

import numpy as np
import pytest

from singa_amd import _rt

RS = np.random.RandomState(11)


def _damaged(blob, n_cuts=40, n_flips=60):
    """Truncations and byte corruptions of ``blob``, deterministic."""
    size = len(blob)
    cuts = sorted(set([0, 1, 2, 3, 4, 7, 8, 9, size - 1] +
                      RS.randint(0, size, n_cuts).tolist()))
    for c in cuts:
        if 0 <= c < size:
            yield blob[:c]
    for _ in range(n_flips):
        b = bytearray(blob)
        for pos in RS.randint(0, size, RS.randint(1, 4)):
            b[pos] = RS.randint(0, 256)
        yield bytes(b)
    # length fields blown up: 0xff runs at every 8-byte boundary of the head
    for at in range(0, min(size - 8, 64), 4):
        b = bytearray(blob)
        b[at:at + 8] = b"\xff" * 8
        yield bytes(b)


def test_tensor_proto_decoder_survives_damage():
    st = _rt.SnapshotTensor()
    st.shape = [3, 5]
    st.dtype = 0
    st.data = np.arange(15, dtype=np.float32).tobytes()
    blob = _rt.EncodeTensorProto(st)
    seen = 0
    for bad in _damaged(blob):
        try:
            t = _rt.DecodeTensorProto(bad)
            n = int(np.prod(list(t.shape))) if len(t.shape) else 0
            assert len(t.data) in (0, 4 * n, 8 * n) or n >= 0
        except (RuntimeError, ValueError, OverflowError):
            pass
        seen += 1
    assert seen > 50


def _write_binfile(path):
    w = _rt.BinFileWriter()
    assert w.Open(path)
    for i in range(6):
        w.Write("key%d" % i, bytes(range(20 + i)))
    w.Write("", b"no key")
    w.Close()
    return open(path, "rb").read()


def test_binfile_reader_survives_damage(tmp_path):
    good = _write_binfile(str(tmp_path / "good.bin"))
    for i, bad in enumerate(_damaged(good)):
        p = str(tmp_path / "bad.bin")
        with open(p, "wb") as f:
            f.write(bad)
        r = _rt.BinFileReader()
        try:
            if not r.Open(p):
                continue
            for _ in range(16):
                if r.Read() is None:
                    break
            r.Count()
        except (RuntimeError, ValueError, MemoryError):
            pass
        finally:
            r.Close()


def test_snapshot_reader_survives_damage(tmp_path):
    prefix = str(tmp_path / "snap")
    s = _rt.Snapshot(prefix, True, 10)
    st = _rt.SnapshotTensor()
    st.shape = [4, 4]
    st.dtype = 0
    st.data = np.ones(16, np.float32).tobytes()
    s.Write("w", st)
    s.Close()
    good = open(prefix + ".bin", "rb").read()
    desc = open(prefix + ".desc", "rb").read()
    for bad in _damaged(good, 25, 40):
        bp = str(tmp_path / "bad")
        with open(bp + ".bin", "wb") as f:
            f.write(bad)
        with open(bp + ".desc", "wb") as f:
            f.write(desc)
        try:
            r = _rt.Snapshot(bp, False, 10)
            r.Names()
            r.Close()
        except (RuntimeError, ValueError, MemoryError):
            pass


def test_lmdb_reader_survives_damage(tmp_path):
    env = str(tmp_path / "env")
    w = _rt.LMDBWriter()
    assert w.Open(env)
    for i in range(300):
        w.Write(b"k%05d" % i, bytes([i % 251]) * (10 + 37 * (i % 9)))
    w.Write(b"big", b"x" * 20000)          # overflow pages
    w.Close()
    data = os.path.join(env, "data.mdb")
    good = open(data, "rb").read()
    page = 4096

    def variants():
        # meta pages, root / branch / leaf headers and node offsets
        for at in list(range(0, 160, 4)) + \
                [page + o for o in range(0, 160, 4)] + \
                [2 * page + o for o in range(0, 64, 2)] + \
                RS.randint(0, len(good), 60).tolist():
            for val in (0xff, 0x00, 0x7f):
                b = bytearray(good)
                if at < len(b):
                    b[at] = val
                    yield bytes(b)
        for c in (0, 10, page - 1, page, 2 * page + 17, len(good) - 5):
            yield good[:c]

    for bad in variants():
        with open(data, "wb") as f:
            f.write(bad)
        r = _rt.LMDBReader()
        try:
            if not r.Open(env):
                continue
            for _ in range(400):
                if r.Read() is None:
                    break
            r.Count()
        except (RuntimeError, ValueError, MemoryError):
            pass
        finally:
            r.Close()


def _cpp_test_exe():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.environ.get("SINGA_CPP_TEST_EXE") or os.path.join(
        root, "singa_amd", "bin", "singa_cpp_test")
    if not os.path.exists(exe):
        pytest.skip("C++ test program not built")
    return exe


def test_jpeg_decoder_survives_damage(tmp_path):
    """The C++ API's baseline JPEG decoder on damaged files: it must exit
    normally (0 = decoded something, 1 = rejected), never by a signal."""
    from PIL import Image
    exe = _cpp_test_exe()
    yy, xx = np.mgrid[0:37, 0:53]
    img = np.stack([xx * 4 % 256, yy * 6 % 256, (xx + yy) * 3 % 256],
                   -1).astype(np.uint8)
    jpg = str(tmp_path / "a.jpg")
    Image.fromarray(img).save(jpg, quality=85)
    good = open(jpg, "rb").read()
    n = 0
    for bad in _damaged(good, 30, 50):
        p = str(tmp_path / "bad.jpg")
        with open(p, "wb") as f:
            f.write(bad)
        r = subprocess.run([exe, "--jpeg-decode", p, str(tmp_path / "o.ppm")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=60)
        assert r.returncode in (0, 1), (r.returncode, r.stdout[-2000:])
        n += 1
    assert n > 50
