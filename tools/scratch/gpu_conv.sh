cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 300 python -u -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python tools/bench_conv.py --check > gpurun_out/r3/bench_conv_r3a.txt 2>&1 || { tail -5 gpurun_out/r3/bench_conv_r3a.txt; exit 1; }
grep "^total" gpurun_out/r3/bench_conv_r3a.txt
timeout -k 10 200 python bench.py --steps 50 --warmup 5 2>/dev/null | tee gpurun_out/r3/bench_r3a.json | cut -c1-200
