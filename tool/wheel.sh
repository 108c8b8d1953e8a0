#!/usr/bin/env bash
# Build a binary wheel of singa_amd for this machine's Python and ROCm:
# compiles the runtime, the gfx950 kernels and the C++ API in-tree, then packs
# them.  The wheel lands in dist/.
# Role parity: reference tool/wheel.sh (manylinux wheels for several Pythons;
# here one wheel per ROCm image, since the kernels are built for gfx950 only).
set -euo pipefail
cd "$(dirname "$0")/.."
python -m singa_amd.build
python setup.py bdist_wheel
ls -l dist/*.whl
