#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_final.log 2>&1
rc=$?; tail -4 gpurun_out/gpu_tests_final.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_final.log 2>&1 || { tail -20 gpurun_out/smoke_final.log; exit 1; }
tail -1 gpurun_out/smoke_final.log
timeout -k 10 300 python bench.py > gpurun_out/bench_final.log 2>&1 || { tail -20 gpurun_out/bench_final.log; exit 1; }
tail -1 gpurun_out/bench_final.log | cut -c1-260
SINGA_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 timeout -k 10 300 python bench.py --steps 50 > gpurun_out/bench_forced_rccl_final.log 2>&1 || { tail -20 gpurun_out/bench_forced_rccl_final.log; exit 1; }
tail -1 gpurun_out/bench_forced_rccl_final.log | cut -c1-260
