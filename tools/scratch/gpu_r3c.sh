#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
export SINGA_CONV_PLAN_CACHE=0
timeout -k 10 600 python -u -m pytest tests/test_hip_kernels.py tests/test_tensor.py tests/test_attention.py tests/test_onnx.py -m gpu -x -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_r3c.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_tests_r3c.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_bert.py --steps 20 --train-batches 8,32 --no-unfused --dtype fp16 > gpurun_out/bert_fp16_r3c.log 2>&1 || { tail -20 gpurun_out/bert_fp16_r3c.log; exit 1; }
tail -1 gpurun_out/bert_fp16_r3c.log
timeout -k 10 200 python tools/bench_gemm.py --dtype float32 --vendor > gpurun_out/gemm_fp32_vs_vendor_r3.txt 2>&1 || exit 1
timeout -k 10 200 python tools/bench_gemm.py --dtype float16 --vendor > gpurun_out/gemm_fp16_vs_vendor_r3.txt 2>&1 || exit 1
cat gpurun_out/gemm_fp32_vs_vendor_r3.txt gpurun_out/gemm_fp16_vs_vendor_r3.txt
