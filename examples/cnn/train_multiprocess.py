"""Data-parallel training with the ranks started from this script by
``multiprocessing`` (no external launcher):

    python train_multiprocess.py cnn mnist -w 2

The parent creates an ``NcclIdHolder`` (the rendezvous address) that every
child passes to ``opt.DistOpt`` together with its rank.  Counterpart of the
reference's examples/cnn/train_multiprocess.py.
"""
import argparse
import multiprocessing

import train_cnn
from singa_amd import opt
from singa_amd import singa_wrap as singa


def run(args, local_rank, world_size, nccl_id):
    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                  dtype=train_cnn.singa_dtype[args.precision])
    sgd = opt.DistOpt(sgd, nccl_id=nccl_id, local_rank=local_rank,
                      world_size=world_size)
    graph = args.graph and args.dist_option != "partialUpdate"
    train_cnn.run(sgd.global_rank, sgd.world_size, sgd.local_rank,
                  args.max_epoch, args.batch_size, args.model, args.data, sgd,
                  graph, args.verbosity, args.dist_option, args.spars,
                  args.precision, max_batches=args.max_batches)


if __name__ == "__main__":
    parser = train_cnn.add_common_args(argparse.ArgumentParser(
        description="Multiprocess distributed training."))
    parser.set_defaults(max_epoch=10)
    parser.add_argument("-w", "--world-size", default=2, type=int,
                        dest="world_size")
    parser.add_argument("-d", "--dist-option", default="plain",
                        choices=["plain", "half", "partialUpdate",
                                 "sparseTopK", "sparseThreshold"],
                        dest="dist_option")
    parser.add_argument("-s", "--sparsification", default=0.05, type=float,
                        dest="spars")
    args = parser.parse_args()

    # spawn: a forked child must not inherit an initialised HIP runtime
    ctx = multiprocessing.get_context("spawn")
    nccl_id = singa.NcclIdHolder()
    procs = [ctx.Process(target=run,
                         args=(args, rank, args.world_size, nccl_id))
             for rank in range(args.world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    raise SystemExit(max(p.exitcode or 0 for p in procs))
