#!/usr/bin/env bash
# Train AlexNet from the record files written by create_data.sh.
#   ./run.sh [data_dir]            (full ImageNet geometry)
#   SYNTHETIC=1 ./run.sh [data_dir] (the small synthetic set of create_data.sh)
set -euo pipefail
BIN="$(dirname "$0")/../../../singa_amd/bin"
DATA="${1:-imagenet_data}"
if [ "${SYNTHETIC:-0}" = "1" ]; then
  "$BIN/singa_alexnet" -data "$DATA" -ntrain 2048 -ntest 256 -filesize 1024 \
      -classes 10 -epoch 5 -batchsize 128 -pfreq 4 -nthreads 16
else
  "$BIN/singa_alexnet" -data "$DATA" -ntrain 1281167 -ntest 50000 \
      -filesize 1280 -epoch 90 -lr 0.01 -batchsize 256 -pfreq 100 -nthreads 16
fi
