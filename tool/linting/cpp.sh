#!/usr/bin/env bash
# Syntax-only compile of every C++ / HIP source with warnings on.
set -e
cd "$(dirname "$0")/../.."
ROCM=${ROCM_PATH:-/opt/rocm}
for f in $(find singa_amd/csrc/cpp/src singa_amd/csrc/runtime -name '*.cc'); do
  g++ -std=c++17 -fsyntax-only -Wall -Wextra -Wno-unused-parameter \
      -D__HIP_PLATFORM_AMD__ -I$ROCM/include -Isinga_amd/csrc/cpp/include "$f"
done
for f in singa_amd/csrc/hip/*.hip; do
  $ROCM/bin/hipcc --offload-arch=gfx950 -std=c++17 -fsyntax-only -Wall "$f"
done
echo "lint ok"
