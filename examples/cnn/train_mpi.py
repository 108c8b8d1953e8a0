"""Data-parallel training, one process per GPU, ranks started by a launcher:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        train_mpi.py resnet cifar10 -b 32 -l 0.015
    mpiexec -np 8 python train_mpi.py cnn mnist

``opt.DistOpt`` reads rank / world size from the launcher's environment
(torchrun, Open MPI, MPICH/PMI or Slurm) and builds the RCCL communicator;
MPI itself carries no tensor traffic.  Counterpart of the reference's
examples/cnn/train_mpi.py:29-91.
"""
import argparse

import train_cnn
from singa_amd import opt

if __name__ == "__main__":
    parser = train_cnn.add_common_args(argparse.ArgumentParser(
        description="Distributed training using the autograd and graph."))
    parser.set_defaults(max_epoch=10)
    parser.add_argument("-d", "--dist-option", default="plain",
                        choices=["plain", "half", "partialUpdate",
                                 "sparseTopK", "sparseThreshold"],
                        dest="dist_option")
    parser.add_argument("-s", "--sparsification", default=0.05, type=float,
                        dest="spars")
    args = parser.parse_args()

    sgd = opt.SGD(lr=args.lr, momentum=0.9, weight_decay=1e-5,
                  dtype=train_cnn.singa_dtype[args.precision])
    sgd = opt.DistOpt(sgd)
    # the partial-update scheme changes which bucket is exchanged every
    # iteration, which a frozen graph cannot express
    graph = args.graph and args.dist_option != "partialUpdate"
    train_cnn.run(sgd.global_rank, sgd.world_size, sgd.local_rank,
                  args.max_epoch, args.batch_size, args.model, args.data, sgd,
                  graph, args.verbosity, args.dist_option, args.spars,
                  args.precision, max_batches=args.max_batches,
                  checkpoint_dir=args.checkpoint_dir,
                  checkpoint_every=args.checkpoint_every)
