#!/usr/bin/env bash
# Data-parallel CIFAR-10 training inside one process (one replica per GPU):
#   ./run-parallel.sh [cnn|vgg] [replicas]
set -euo pipefail
BIN="$(dirname "$0")/../../../singa_amd/bin"
MODEL="${1:-cnn}"
REPLICAS="${2:-2}"
if [ "$MODEL" = "vgg" ]; then
  "$BIN/singa_vgg_parallel" -epoch 250 -lr 0.05 -replicas "$REPLICAS"
else
  "$BIN/singa_cnn_parallel" -epoch 140 -lr 0.01 -replicas "$REPLICAS"
fi
