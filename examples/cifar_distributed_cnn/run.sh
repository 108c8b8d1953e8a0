#!/usr/bin/env bash
# 8 ranks, one per MI355X, ResNet-50 on CIFAR-10 shaped input, batch 32 per GPU
python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
    train_mpi.py resnet cifar10 -l 0.015 -b 32
