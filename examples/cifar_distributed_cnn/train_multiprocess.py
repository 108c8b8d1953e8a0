"""Throughput harness with ranks started by multiprocessing; see train_cnn.py
in this directory."""
import multiprocessing

import train_cnn
from singa_amd import opt
from singa_amd import singa_wrap as singa


def run(args, local_rank, world_size, nccl_id):
    sgd = opt.DistOpt(opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                              dtype=train_cnn.base.singa_dtype[args.precision]),
                      nccl_id=nccl_id, local_rank=local_rank,
                      world_size=world_size)
    graph = args.graph and args.dist_option != "partialUpdate"
    train_cnn.run(sgd.global_rank, sgd.world_size, sgd.local_rank,
                  args.max_epoch, args.batch_size, args.model, args.data, sgd,
                  graph, args.verbosity, args.dist_option, args.spars,
                  args.precision, niters=args.iters)


if __name__ == "__main__":
    parser = train_cnn.parser_with_common_args(
        "Throughput harness, multiprocess data parallel.")
    parser.add_argument("-w", "--world-size", default=2, type=int,
                        dest="world_size")
    parser.add_argument("-d", "--dist-option", default="plain",
                        choices=["plain", "half", "partialUpdate",
                                 "sparseTopK", "sparseThreshold"],
                        dest="dist_option")
    parser.add_argument("-s", "--sparsification", default=0.05, type=float,
                        dest="spars")
    args = parser.parse_args()
    ctx = multiprocessing.get_context("spawn")
    nccl_id = singa.NcclIdHolder()
    procs = [ctx.Process(target=run,
                         args=(args, r, args.world_size, nccl_id))
             for r in range(args.world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    raise SystemExit(max(p.exitcode or 0 for p in procs))
