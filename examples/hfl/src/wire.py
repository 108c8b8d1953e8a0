"""Wire format of the federated-learning example.

A message is ``magic 'HFL1' | op (u8) | count (u32)`` followed by ``count``
named tensors, each ``name_len (u16) | name | dtype code (u8) | ndim (u8) |
dims (u32 each) | raw little-endian data``.  Nothing in a message is
executable: the receiver only ever builds numpy arrays of the three dtypes
the framework trains with.  (The reference exchanges protobuf
``WeightsExchange`` messages whose tensor payloads are pickles:
examples/hfl/src/proto/utils.py.)
"""
import socket
import struct

import numpy as np

GATHER, SCATTER = 1, 2
_MAGIC = b"HFL1"
_DTYPES = {0: np.float32, 1: np.float16, 2: np.int32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _recv_exact(conn, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        r = conn.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("peer closed the connection")
        got += r
    return bytes(buf)


def send_int(conn, value):
    conn.sendall(struct.pack("<q", int(value)))


def receive_int(conn):
    return struct.unpack("<q", _recv_exact(conn, 8))[0]


def send_weights(conn, op, weights):
    """weights: dict name -> numpy array."""
    parts = [_MAGIC, struct.pack("<BI", op, len(weights))]
    for name, a in weights.items():
        a = np.ascontiguousarray(a)
        if a.dtype not in _CODES:
            a = a.astype(np.float32)
        nb = name.encode("utf-8")
        parts.append(struct.pack("<H", len(nb)) + nb)
        parts.append(struct.pack("<BB", _CODES[a.dtype], a.ndim))
        parts.append(struct.pack("<%dI" % a.ndim, *a.shape))
        parts.append(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    conn.sendall(b"".join(parts))


def receive_weights(conn):
    """-> (op, dict name -> numpy array)."""
    if _recv_exact(conn, 4) != _MAGIC:
        raise ValueError("bad message header")
    op, count = struct.unpack("<BI", _recv_exact(conn, 5))
    out = {}
    for _ in range(count):
        (nlen,) = struct.unpack("<H", _recv_exact(conn, 2))
        name = _recv_exact(conn, nlen).decode("utf-8")
        code, ndim = struct.unpack("<BB", _recv_exact(conn, 2))
        shape = struct.unpack("<%dI" % ndim, _recv_exact(conn, 4 * ndim))
        dt = np.dtype(_DTYPES[code]).newbyteorder("<")
        n = int(np.prod(shape)) if ndim else 1
        data = _recv_exact(conn, n * dt.itemsize)
        out[name] = np.frombuffer(data, dt).reshape(shape).astype(
            _DTYPES[code])
    return op, out


def connect(host, port, retries=50, delay=0.2):
    import time
    last = None
    for _ in range(retries):
        try:
            s = socket.create_connection((host, port))
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            return s
        except OSError as e:  # server not listening yet
            last = e
            time.sleep(delay)
    raise last
