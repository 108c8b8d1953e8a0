#!/bin/bash
# The configurations the reference launches with run-rtx.sh (8 ranks, batch 32
# per rank, learning rate scaled by the number of ranks), one process per
# MI355X over RCCL.  N defaults to the GPUs of this machine.
N=${1:-$(python -c 'import torch; print(max(torch.cuda.device_count(), 1))')}
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533"
cd "$(dirname "$0")"
set -e
$RUN train_mpi.py resnet cifar10 -l 0.015 -b 32
$RUN train_mpi.py resnet cifar10 -l 0.015 -b 32 -d half
$RUN train_mpi.py resnet cifar10 -l 0.015 -b 32 -d partialUpdate
$RUN train_mpi.py resnet cifar10 -l 0.015 -b 32 -d sparseTopK -s 0.05
$RUN train_mpi.py resnet cifar10 -l 0.015 -b 32 -d sparseThreshold -s 0.05
$RUN benchmark.py --dist
