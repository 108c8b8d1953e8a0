"""Throughput harness as one rank of a launcher-started job (torchrun /
mpiexec); see train_cnn.py in this directory."""
import train_cnn
from singa_amd import opt

if __name__ == "__main__":
    parser = train_cnn.parser_with_common_args(
        "Throughput harness, data parallel.")
    parser.add_argument("-d", "--dist-option", default="plain",
                        choices=["plain", "half", "partialUpdate",
                                 "sparseTopK", "sparseThreshold"],
                        dest="dist_option")
    parser.add_argument("-s", "--sparsification", default=0.05, type=float,
                        dest="spars")
    args = parser.parse_args()
    sgd = opt.DistOpt(opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                              dtype=train_cnn.base.singa_dtype[args.precision]))
    graph = args.graph and args.dist_option != "partialUpdate"
    train_cnn.run(sgd.global_rank, sgd.world_size, sgd.local_rank,
                  args.max_epoch, args.batch_size, args.model, args.data, sgd,
                  graph, args.verbosity, args.dist_option, args.spars,
                  args.precision, niters=args.iters)
