cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 200 python tools/bench_bn.py 2>&1 | tee gpurun_out/r3/bench_bn_base.txt | tail -25
