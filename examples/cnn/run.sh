#!/usr/bin/env bash
# Typical invocations of the scripts in this directory.
set -e
# single GPU
python train_cnn.py cnn mnist
python train_cnn.py resnet cifar10 -l 0.015 -b 32
python train_cnn.py xceptionnet cifar10 -l 0.015 -b 32
# 8 GPUs of one node, one process per GPU over RCCL
python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
    train_mpi.py resnet cifar10 -l 0.015 -b 32
# ranks started by multiprocessing
python train_multiprocess.py cnn mnist -w 2
# gradient compression variants
python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
    train_mpi.py cnn cifar10 -d half
python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
    train_mpi.py cnn cifar10 -d sparseTopK -s 0.05
