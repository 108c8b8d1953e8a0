#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
export SINGA_CONV_PLAN_CACHE=0
timeout -k 10 400 python -u -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv" --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_r3d.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_tests_r3d.log
[ $rc -eq 0 ] || exit $rc
SINGA_CONV16_AUTOTUNE=0 timeout -k 10 200 python tools/bench_conv.py --dtype float16 > gpurun_out/conv_fp16_rule.txt 2>&1 || exit 1
grep "^total" gpurun_out/conv_fp16_rule.txt
timeout -k 10 300 python tools/bench_conv.py --dtype float16 > gpurun_out/conv_fp16_tuned.txt 2>&1 || exit 1
grep "^total" gpurun_out/conv_fp16_tuned.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --dtype fp16 > gpurun_out/bench_fp16_r3d.log 2>&1 || { tail -20 gpurun_out/bench_fp16_r3d.log; exit 1; }
tail -1 gpurun_out/bench_fp16_r3d.log | cut -c1-300
