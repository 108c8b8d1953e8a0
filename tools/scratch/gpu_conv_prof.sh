cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
SINGA_CONV_AUTOTUNE=0 timeout -k 10 200 python tools/bench_conv.py --prof --layers 2,3,5,9,12,13,16,19,20,21,25,27 > gpurun_out/r3/bench_conv_prof.txt 2>&1
grep -v "^plan" gpurun_out/r3/bench_conv_prof.txt | tail -60
