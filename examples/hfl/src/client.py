"""Federated-learning client: trains the model on its own data shard for one
(or several) local epochs per round, pushes the weights to the server and
continues from the average it pulls back (counterpart of the reference's
examples/hfl/src/client.py:44-374)."""
import os
import sys
import time

import numpy as np

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..", "..")]

from singa_amd import device, opt, tensor  # noqa: E402

from . import bank, mlp  # noqa: E402
from .proto import interface as proto  # noqa: E402
from .proto import utils  # noqa: E402
from .proto.utils import parseargs  # noqa: E402

np_dtype = {"float16": np.float16, "float32": np.float32}
singa_dtype = {"float16": tensor.float16, "float32": tensor.float32}


class Client(object):
    """Exchanges ``WeightsExchange`` messages with the server.

    Attributes:
        global_rank: rank announced to the server after connecting.
        weights: name -> tensor, what ``push`` sends and ``pull`` fills.
    """

    def __init__(self, global_rank=0, host="127.0.0.1", port=1234, dev=None):
        self.global_rank = global_rank
        self.host, self.port = host, int(port)
        self.sock = None
        self.dev = dev
        self.weights = {}

    def start(self):
        self.sock = utils.connect(self.host, self.port)
        utils.send_int(self.sock, self.global_rank)
        print("[Client %d] Connect to %s:%s" % (self.global_rank, self.host,
                                                self.port), flush=True)

    def close(self):
        self.sock.close()

    def pull(self):
        """Receive the aggregated weights."""
        message = utils.receive_message(self.sock, proto.WeightsExchange())
        if message.op_type != proto.SCATTER:
            raise ValueError("expected a SCATTER message, got op %d"
                             % message.op_type)
        for k, blob in message.weights.items():
            self.weights[k] = utils.deserialize_tensor(blob, self.dev)

    def push(self):
        """Send the local weights."""
        message = proto.WeightsExchange()
        message.op_type = proto.GATHER
        for k, v in self.weights.items():
            message.weights[k] = utils.serialize_tensor(v)
        utils.send_message(self.sock, message)


def accuracy(pred, target):
    """Number of rows whose arg-max is the label."""
    return int((np.argmax(pred, axis=1) == target).sum())


def partition(global_rank, world_size, train_x, train_y, val_x, val_y):
    """The slice of a client's shard that one of its data-parallel ranks
    trains on."""
    per = train_x.shape[0] // world_size
    train_x = train_x[global_rank * per:(global_rank + 1) * per]
    train_y = train_y[global_rank * per:(global_rank + 1) * per]
    per = val_x.shape[0] // world_size
    val_x = val_x[global_rank * per:(global_rank + 1) * per]
    val_y = val_y[global_rank * per:(global_rank + 1) * per]
    return train_x, train_y, val_x, val_y


def reduce_variable(variable, dist_opt, reducer):
    """Sum of a host-side number over the ranks of a DistOpt."""
    reducer.copy_from_numpy(variable)
    dist_opt.all_reduce(reducer.data)
    dist_opt.wait()
    return tensor.to_numpy(reducer)


def get_data(data, data_dist="iid", device_id=None, data_dir=None):
    if data == "bank":
        # the distribution is decided when the shards are written
        # (python -m src.bank N [--non-iid])
        return bank.load(device_id, data_dir)
    raise NotImplementedError(data)


def get_model(model, num_channels=None, num_classes=None, data_size=None):
    if model == "mlp":
        return mlp.create_model(data_size=data_size, num_classes=num_classes)
    raise NotImplementedError(model)


def run(global_rank, world_size, device_id, max_epoch, batch_size, model,
        data, data_dist, sgd, graph, verbosity, dist_option="plain",
        spars=None, precision="float32", host="127.0.0.1", port=1234,
        data_dir=None, inner_epochs=1, dev=None, log=print):
    """One client.  ``global_rank`` / ``world_size`` describe the
    data-parallel group INSIDE the client (1 process unless ``sgd`` is a
    DistOpt); ``device_id`` is the client's rank at the server and the index
    of its data shard.  Returns the list of per-round evaluation accuracies.
    """
    dev = dev or device.get_default_device()
    client = Client(global_rank=device_id, host=host, port=port, dev=dev)
    client.start()
    dev.SetRandSeed(0)
    np.random.seed(0)

    train_x, train_y, val_x, val_y, num_classes = get_data(
        data, data_dist, device_id, data_dir)
    num_channels = train_x.shape[1]
    data_size = int(np.prod(train_x.shape[1:]))
    model = get_model(model, num_channels=num_channels,
                      num_classes=num_classes, data_size=data_size)

    dist = hasattr(sgd, "communicator")
    sequential = dist     # bucketed collectives want the serial order
    if dist:
        train_x, train_y, val_x, val_y = partition(
            global_rank, world_size, train_x, train_y, val_x, val_y)

    assert model.dimension == 2
    train_x = train_x.reshape(train_x.shape[0], -1)
    val_x = val_x.reshape(val_x.shape[0], -1)
    tx = tensor.Tensor((batch_size, data_size), dev, singa_dtype[precision])
    ty = tensor.Tensor((batch_size,), dev, tensor.int32)
    num_train_batch = train_x.shape[0] // batch_size
    num_val_batch = val_x.shape[0] // batch_size
    idx = np.arange(train_x.shape[0], dtype=np.int32)

    model.set_optimizer(sgd)
    model.compile([tx], is_train=True, use_graph=graph, sequential=sequential)
    dev.SetVerbosity(verbosity)

    history = []
    for epoch in range(max_epoch):
        if epoch > 0:
            client.pull()
            model.set_states(client.weights)
        if global_rank == 0:
            log("Starting Epoch %d:" % epoch)
        start_time = time.time()
        np.random.shuffle(idx)

        for inner_epoch in range(inner_epochs):
            train_correct = np.zeros(shape=[1], dtype=np.float32)
            train_loss = np.zeros(shape=[1], dtype=np.float32)
            test_correct = np.zeros(shape=[1], dtype=np.float32)

            model.train()
            for b in range(num_train_batch):
                sel = idx[b * batch_size:(b + 1) * batch_size]
                x = train_x[sel].astype(np_dtype[precision])
                y = train_y[sel]
                tx.copy_from_numpy(x)
                ty.copy_from_numpy(y)
                out, loss = model(tx, ty, dist_option, spars)
                train_correct += accuracy(tensor.to_numpy(out), y)
                train_loss += float(tensor.to_numpy(loss).reshape(-1)[0])

            if dist:
                reducer = tensor.Tensor((1,), dev, tensor.float32)
                train_correct = reduce_variable(train_correct, sgd, reducer)
                train_loss = reduce_variable(train_loss, sgd, reducer)
            if global_rank == 0:
                log("[inner epoch %d] Training loss = %f, training accuracy "
                    "= %f" % (inner_epoch, train_loss[0], train_correct[0] /
                              (num_train_batch * batch_size * world_size)))

            model.eval()
            for b in range(num_val_batch):
                x = val_x[b * batch_size:(b + 1) * batch_size]
                y = val_y[b * batch_size:(b + 1) * batch_size]
                tx.copy_from_numpy(x.astype(np_dtype[precision]))
                ty.copy_from_numpy(y)
                test_correct += accuracy(tensor.to_numpy(model(tx)), y)
            if dist:
                test_correct = reduce_variable(test_correct, sgd, reducer)
            acc = float(test_correct[0]) / max(
                num_val_batch * batch_size * world_size, 1)
            if global_rank == 0:
                log("[inner epoch %d] Evaluation accuracy = %f, Elapsed "
                    "Time = %fs" % (inner_epoch, acc,
                                    time.time() - start_time))
        history.append(acc)
        client.weights = model.get_states()
        client.push()

    dev.PrintTimeProfiling()
    client.close()
    return history


def main(argv=None):
    args = parseargs(argv)
    dev = device.get_default_device() if args.cpu else None
    if dev is None:
        try:
            dev = device.create_cuda_gpu()
        except Exception:               # no GPU on this machine
            dev = device.get_default_device()
    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                  dtype=singa_dtype[args.precision])

    def log(msg):
        print("[Client %d] %s" % (args.device_id, msg), flush=True)
    return run(0, 1, args.device_id, args.max_epoch, args.batch_size,
               args.model, args.data, args.data_dist, sgd, args.graph,
               args.verbosity, precision=args.precision, host=args.host,
               port=args.port, data_dir=args.data_dir,
               inner_epochs=args.inner_epochs, dev=dev, log=log)


if __name__ == "__main__":
    main()
