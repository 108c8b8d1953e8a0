#!/usr/bin/env bash
# Python style check (same role as the reference's tool/linting/py.sh).
set -e
cd "$(dirname "$0")/../.."
python -m pyflakes singa_amd singa examples tests tools 2>/dev/null || \
  python -m compileall -q singa_amd singa examples tests tools
