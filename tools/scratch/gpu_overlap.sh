cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3/overlap
export TMPDIR=/tmp SINGA_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3/overlap -- python3 bench.py --steps 4 --warmup 1 --bucket 2097152 > gpurun_out/r3/overlap.log 2>&1
tail -2 gpurun_out/r3/overlap.log | cut -c1-300
python tools/overlap_timeline.py gpurun_out/r3/overlap > gpurun_out/r3/overlap_timeline_1rank.md; tail -25 gpurun_out/r3/overlap_timeline_1rank.md
f=$(ls gpurun_out/r3/overlap/*/*kernel_trace.csv | head -1); head -1 $f; grep -i "nccl" $f | head -3; rm -f $f
