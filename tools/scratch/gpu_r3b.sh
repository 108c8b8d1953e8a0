#!/usr/bin/env bash
# GPU validation after the GEMM tile family / failure-detection hooks
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_r3b.log 2>&1
rc=$?; tail -5 gpurun_out/gpu_tests_r3b.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_bert.py --steps 20 --train-batches 32 --no-unfused --dtype fp32 > gpurun_out/bert_fp32_r3b.log 2>&1 || { tail -20 gpurun_out/bert_fp32_r3b.log; exit 1; }
tail -3 gpurun_out/bert_fp32_r3b.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_r3b.log 2>&1 || { tail -20 gpurun_out/bench_r3b.log; exit 1; }
tail -1 gpurun_out/bench_r3b.log
