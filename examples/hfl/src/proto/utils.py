"""Socket framing, tensor serialisation and the command line of the
federated-learning example (counterpart of the reference's
examples/hfl/src/proto/utils.py:33-182).

A message on the socket is an 8-byte little-endian length followed by a
serialized ``WeightsExchange``.  A tensor travels as a numpy ``.npy`` blob
written and read with ``allow_pickle=False``: a receiver only ever builds
plain numeric arrays from what a peer sends.  (The reference pickles the
arrays, which lets a peer run code in the receiving process.)
"""
import argparse
import io
import socket
import struct
import time

import numpy as np

# messages larger than this are refused before any allocation
MAX_MESSAGE_BYTES = 1 << 31


def receive_all(conn, size):
    """Exactly ``size`` bytes from the socket."""
    buf = bytearray(size)
    view = memoryview(buf)
    got = 0
    while got < size:
        n = conn.recv_into(view[got:], size - got)
        if n == 0:
            raise ConnectionError("peer closed the connection after %d of %d "
                                  "bytes" % (got, size))
        got += n
    return bytes(buf)


def send_int(conn, i, pack_format="<q"):
    conn.sendall(struct.pack(pack_format, int(i)))


def receive_int(conn, pack_format="<q"):
    return struct.unpack(pack_format,
                         receive_all(conn, struct.calcsize(pack_format)))[0]


def send_message(conn, message, pack_format="<q"):
    data = message.SerializeToString()
    send_int(conn, len(data), pack_format)
    conn.sendall(data)


def receive_message(conn, message, pack_format="<q"):
    size = receive_int(conn, pack_format)
    if size < 0 or size > MAX_MESSAGE_BYTES:
        raise ValueError("message length %d out of range" % size)
    message.ParseFromString(receive_all(conn, size))
    return message


def serialize_array(a):
    out = io.BytesIO()
    np.lib.format.write_array(out, np.ascontiguousarray(a), version=(1, 0),
                              allow_pickle=False)
    return out.getvalue()


def deserialize_array(blob):
    return np.load(io.BytesIO(blob), allow_pickle=False)


def serialize_tensor(t):
    """singa tensor (or numpy array) -> bytes."""
    from singa_amd import tensor
    if isinstance(t, np.ndarray):
        return serialize_array(t)
    return serialize_array(tensor.to_numpy(t))


def deserialize_tensor(blob, dev=None):
    """bytes -> singa tensor on ``dev`` (the default device when None)."""
    from singa_amd import tensor
    return tensor.from_numpy(deserialize_array(blob), dev)


def connect(host, port, retries=100, delay=0.1):
    """Connect, waiting for a server that is not listening yet."""
    last = None
    for _ in range(retries):
        try:
            s = socket.create_connection((host, int(port)))
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            return s
        except OSError as e:
            last = e
            time.sleep(delay)
    raise last


def parseargs(arg=None):
    p = argparse.ArgumentParser(
        description="Horizontal federated learning with FedAvg.")
    p.add_argument("--model", choices=["mlp"], default="mlp")
    p.add_argument("--data", choices=["bank"], default="bank")
    p.add_argument("-p", choices=["float32", "float16"], default="float32",
                   dest="precision")
    p.add_argument("-m", "--max-epoch", default=10, type=int,
                   help="communication rounds (one local epoch each)",
                   dest="max_epoch")
    p.add_argument("-b", "--batch-size", default=64, type=int,
                   dest="batch_size")
    p.add_argument("-l", "--learning-rate", default=0.005, type=float,
                   dest="lr")
    p.add_argument("-i", "--device-id", default=0, type=int,
                   help="rank of this client = index of its data shard",
                   dest="device_id")
    p.add_argument("-g", "--disable-graph", default=True,
                   action="store_false", dest="graph")
    p.add_argument("-v", "--log-verbosity", default=0, type=int,
                   dest="verbosity")
    p.add_argument("-d", "--data-distribution", choices=["iid", "non-iid"],
                   default="iid", dest="data_dist")
    p.add_argument("--inner-epochs", default=1, type=int, dest="inner_epochs",
                   help="local epochs between two exchanges")
    p.add_argument("--num_clients", default=10, type=int)
    p.add_argument("--host", default="127.0.0.1")
    p.add_argument("--port", default=1234, type=int)
    p.add_argument("--data-dir", default=None, dest="data_dir",
                   help="directory of the bank_{train,test}_<i>.csv shards "
                        "(default: data/ next to src/)")
    p.add_argument("--cpu", action="store_true",
                   help="train on the host device")
    return p.parse_args(arg)
