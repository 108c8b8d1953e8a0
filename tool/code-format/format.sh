#!/usr/bin/env bash
# Format the sources in place: clang-format (Google style, as the reference's
# tool/code-format/format.sh) for C++ / HIP, autopep8 / yapf for Python when
# they are installed.
set -e
cd "$(dirname "$0")/../.."
CF=${CLANG_FORMAT:-/opt/rocm/lib/llvm/bin/clang-format}
if [ -x "$CF" ] || command -v "$CF" >/dev/null 2>&1; then
  find singa_amd/csrc examples/cpp \( -name '*.h' -o -name '*.cc' -o -name '*.hip' -o -name '*.cuh' \) \
    -print0 | xargs -0 "$CF" -i --style=Google
else
  echo "clang-format not found (set CLANG_FORMAT)" >&2
fi
if command -v yapf >/dev/null 2>&1; then
  yapf -i -r --style google singa_amd singa tests examples tools bench.py
elif command -v autopep8 >/dev/null 2>&1; then
  autopep8 -i -r singa_amd singa tests examples tools bench.py
else
  echo "no Python formatter found (yapf / autopep8)" >&2
fi
