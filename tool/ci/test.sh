#!/usr/bin/env bash
# What a CI job runs (counterpart of the reference's tool/jenkins/test.sh):
# build every native component, run the host test suite and the C++ tests;
# with a GPU present also the device suite, the smoke step and a short bench.
set -euo pipefail
cd "$(dirname "$0")/../.."
python -m singa_amd.build
python -m pytest tests -x -q -m "not gpu"
singa_amd/bin/singa_cpp_test
if python -c 'import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)'; then
  python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread
  singa_amd/bin/singa_cpp_test --gpu
  python -c 'import __graft_entry__ as g; g.smoke()'
  python bench.py --steps 20 --warmup 5
fi
