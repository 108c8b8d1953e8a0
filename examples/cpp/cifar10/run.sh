#!/usr/bin/env bash
# CIFAR-10 CNN with the C++ API; cifar-10-batches-bin/ is read from the
# current directory (synthetic images when it is absent).
set -euo pipefail
"$(dirname "$0")/../../../singa_amd/bin/singa_cnn" -epoch 140 "$@"
