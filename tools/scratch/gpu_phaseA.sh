set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 500 python -u -m pytest tests/test_dist.py tests/test_rnn.py tests/test_opt_model.py -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r3/phaseA_tests.log 2>&1 || { tail -40 gpurun_out/r3/phaseA_tests.log; exit 1; }
tail -3 gpurun_out/r3/phaseA_tests.log
SINGA_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout -k 10 200 python bench.py --steps 50 --warmup 5 > gpurun_out/r3/bench_forced1rank.json 2> gpurun_out/r3/bench_forced1rank.err || { tail -20 gpurun_out/r3/bench_forced1rank.err; exit 1; }
cat gpurun_out/r3/bench_forced1rank.json
timeout -k 10 300 bash tools/run_scaling.sh --steps 30 --warmup 5 --out gpurun_out/r3/scaling > gpurun_out/r3/scaling.log 2>&1 || { tail -20 gpurun_out/r3/scaling.log; exit 1; }
tail -8 gpurun_out/r3/scaling.log
