#!/usr/bin/env bash
# Host code under AddressSanitizer + UndefinedBehaviorSanitizer: the C++
# runtime (graph scheduler, record / LMDB / snapshot files, logging), the C++
# API library (tensor, layers, JPEG / CSV codecs, epoll endpoint) and the
# Python tests that drive their parsers with files and byte strings.
#   tool/ci/sanitize.sh [build-dir]        (default /tmp/singa_amd_sanitize)
# Needs no GPU: device code is not instrumented, and every test runs on the
# host device.  Exits non-zero on the first report.
set -euo pipefail
cd "$(dirname "$0")/../.."
OUT=${1:-/tmp/singa_amd_sanitize}
python -m singa_amd.build --sanitize "$OUT"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== singa_cpp_test (host) under ASan/UBSan =="
"$OUT/bin/singa_cpp_test"
echo "== Python IO / scheduler / C-API tests on the sanitized runtime =="
RT=$(ls "$OUT"/_rt*.so)
LIBASAN=$(g++ -print-file-name=libasan.so)
LIBUBSAN=$(g++ -print-file-name=libubsan.so)
# the interpreter itself is not instrumented: the sanitizer runtimes are
# loaded ahead of it
LD_PRELOAD="$LIBASAN:$LIBUBSAN${LD_PRELOAD:+:$LD_PRELOAD}" \
SINGA_RT_LIBRARY="$RT" SINGA_CPP_LIBRARY="$OUT/lib/libsinga_amd_cpp.so" SINGA_CPP_TEST_EXE="$OUT/bin/singa_cpp_test" \
CUDA_VISIBLE_DEVICES="" \
  python -m pytest tests/test_io.py tests/test_scheduler.py tests/test_data.py \
    tests/test_malformed_inputs.py tests/test_conf_proto.py -x -q -s -m "not gpu" -p no:cacheprovider
echo "sanitize: clean"
