#!/usr/bin/env bash
# Convert ILSVRC12 (baseline JPEG files + "<path> <label>" lists) into the
# record files singa_alexnet reads.  Without arguments a small synthetic set is
# written instead, which is enough to try the whole pipeline.
#   ./create_data.sh [imagenet_root] [output_dir]
set -euo pipefail
BIN="$(dirname "$0")/../../../singa_amd/bin"
OUT="${2:-imagenet_data}"
mkdir -p "$OUT"
if [ $# -ge 1 ]; then
  "$BIN/singa_ilsvrc12" -trainlist "$1/label/train.txt" \
      -trainfolder "$1/ILSVRC2012_img_train" -testlist "$1/label/val.txt" \
      -testfolder "$1/ILSVRC2012_img_val" -outdata "$OUT" -filesize 1280
else
  "$BIN/singa_ilsvrc12" -synthetic -outdata "$OUT" -ntrain 2048 -ntest 256 \
      -classes 10 -filesize 1024
fi
