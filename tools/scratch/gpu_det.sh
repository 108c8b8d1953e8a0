cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 500 python -u -m pytest tests/test_determinism.py -m gpu -x -q -s --timeout 280 --timeout-method thread 2>&1 | tee gpurun_out/r3/determinism.log | tail -15
