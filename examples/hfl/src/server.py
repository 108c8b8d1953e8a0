"""Federated-averaging server: every round it gathers the clients' weights,
averages them and scatters the result back (counterpart of the reference's
examples/hfl/src/server.py:35-162)."""
import argparse
import socket
from collections import defaultdict

import numpy as np

from . import wire


class Server(object):

    def __init__(self, num_clients=1, host="127.0.0.1", port=1234):
        self.num_clients = num_clients
        self.host, self.port = host, int(port)
        self.sock = socket.socket()
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.conns = [None] * num_clients
        self.addrs = [None] * num_clients
        self.weights = {}

    def start(self):
        """Listen, then accept one connection per client; each client
        announces its rank first."""
        self.sock.bind((self.host, self.port))
        self.sock.listen()
        print("Server started.", flush=True)
        for _ in range(self.num_clients):
            conn, addr = self.sock.accept()
            rank = wire.receive_int(conn)
            self.conns[rank], self.addrs[rank] = conn, addr
            print("[Server] Connected by %s [global_rank %d]" % (addr, rank),
                  flush=True)
        assert None not in self.conns

    def close(self):
        for c in self.conns:
            if c is not None:
                c.close()
        self.sock.close()

    def aggregate(self, weights):
        """weights: name -> list of per-client arrays; FedAvg."""
        for k, v in weights.items():
            self.weights[k] = np.mean(np.stack(v), axis=0).astype(v[0].dtype)
        return self.weights

    def pull(self):
        """Gather: receive every client's weights, then aggregate."""
        collected = defaultdict(list)
        for conn in self.conns:
            op, w = wire.receive_weights(conn)
            assert op == wire.GATHER
            for k, v in w.items():
                collected[k].append(v)
        self.aggregate(collected)

    def push(self):
        """Scatter the aggregated weights to every client."""
        for conn in self.conns:
            wire.send_weights(conn, wire.SCATTER, self.weights)


def parseargs(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_clients", type=int, default=2)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=1234)
    ap.add_argument("-m", "--max-epoch", type=int, default=3,
                    dest="max_epoch")
    ap.add_argument("-i", "--rank", type=int, default=0)
    ap.add_argument("-b", "--batch-size", type=int, default=64)
    ap.add_argument("-l", "--lr", type=float, default=0.05)
    return ap.parse_args(argv)


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
