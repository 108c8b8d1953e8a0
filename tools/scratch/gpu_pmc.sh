cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3/pmc1 gpurun_out/r3/pmc2
export TMPDIR=/tmp SINGA_CONV_AUTOTUNE=0
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3/pmc1 -- python3 tools/bench_conv.py --no-graph --iters 3 --layers 5,12,13,19 > gpurun_out/r3/pmc1.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/r3/pmc2 -- python3 tools/bench_conv.py --no-graph --iters 3 --layers 5,12,13,19 > gpurun_out/r3/pmc2.log 2>&1
ls -R gpurun_out/r3/pmc1 | head; tail -3 gpurun_out/r3/pmc2.log
