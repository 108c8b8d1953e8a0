"""Federated-averaging server: accepts one connection per client, then every
round gathers the clients' weights, averages them and scatters the result
(counterpart of the reference's examples/hfl/src/server.py:35-162)."""
import socket
from collections import defaultdict

import numpy as np

from .proto import interface as proto
from .proto import utils
from .proto.utils import parseargs


class Server(object):
    """
    Attributes:
        num_clients: connections to wait for.
        conns / addrs: per global rank, filled by ``start``.
        weights: name -> aggregated numpy array.
        sample_counts: optional per-rank weights of the average (rows each
            client trained on); equal weights when None.
    """

    def __init__(self, num_clients=1, host="127.0.0.1", port=1234,
                 sample_counts=None):
        self.num_clients = num_clients
        self.host, self.port = host, int(port)
        self.sock = socket.socket()
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.conns = [None] * num_clients
        self.addrs = [None] * num_clients
        self.weights = {}
        self.sample_counts = sample_counts

    def __start_connection(self):
        self.sock.bind((self.host, self.port))
        self.sock.listen()
        print("Server started.", flush=True)

    def __start_rank_pairing(self):
        """Every client announces its global rank first."""
        for _ in range(self.num_clients):
            conn, addr = self.sock.accept()
            rank = utils.receive_int(conn)
            if not 0 <= rank < self.num_clients or \
                    self.conns[rank] is not None:
                conn.close()
                raise ValueError("client announced rank %d (expected one "
                                 "unused rank below %d)"
                                 % (rank, self.num_clients))
            self.conns[rank], self.addrs[rank] = conn, addr
            print("[Server] Connected by %s [global_rank %d]" % (addr, rank),
                  flush=True)
        assert None not in self.conns

    def start(self):
        self.__start_connection()
        self.__start_rank_pairing()

    def close(self):
        for c in self.conns:
            if c is not None:
                c.close()
        self.sock.close()

    def aggregate(self, weights):
        """weights: name -> list of per-client arrays (rank order).  FedAvg:
        the mean, weighted by ``sample_counts`` when given; integer tensors
        (step counters) keep their dtype."""
        w = None
        if self.sample_counts is not None:
            w = np.asarray(self.sample_counts, np.float64)
            w = w / w.sum()
        for k, v in weights.items():
            stack = np.stack([np.asarray(a, np.float64) for a in v])
            mean = stack.mean(axis=0) if w is None else \
                np.tensordot(w, stack, axes=1)
            self.weights[k] = mean.astype(v[0].dtype)
        return self.weights

    def pull(self):
        """Gather every client's weights, then aggregate."""
        collected = defaultdict(list)
        for conn in self.conns:
            message = utils.receive_message(conn, proto.WeightsExchange())
            if message.op_type != proto.GATHER:
                raise ValueError("expected a GATHER message, got op %d"
                                 % message.op_type)
            for k, blob in message.weights.items():
                collected[k].append(utils.deserialize_array(blob))
        short = [k for k, v in collected.items() if len(v) != self.num_clients]
        if short:
            raise ValueError("not every client sent %s" % sorted(short))
        self.aggregate(collected)

    def push(self):
        """Scatter the aggregated weights to every client."""
        message = proto.WeightsExchange()
        message.op_type = proto.SCATTER
        for k, v in self.weights.items():
            message.weights[k] = utils.serialize_array(v)
        for conn in self.conns:
            utils.send_message(conn, message)


def main(argv=None):
    args = parseargs(argv)
    server = Server(args.num_clients, args.host, args.port)
    server.start()
    for i in range(args.max_epoch):
        print("On epoch %d:" % i, flush=True)
        if i > 0:
            server.push()
        server.pull()
    server.close()
    return server.weights


if __name__ == "__main__":
    main()
