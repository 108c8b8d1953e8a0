set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/gputests_base.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 50 --warmup 5 > gpurun_out/r3/bench_base.json 2> gpurun_out/r3/bench_base.err && \
timeout -k 10 200 python tools/bench_conv.py > gpurun_out/r3/bench_conv_base.txt 2>&1
