#!/usr/bin/env bash
# kernel breakdown of the fp32 and fp16 training step (graph replay)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for DT in fp32 fp16; do
  rm -rf $OUT/prof_$DT
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$DT -- python3 bench.py --steps 8 --warmup 4 --dtype $DT > $OUT/prof_$DT.log 2>&1 || { tail -5 $OUT/prof_$DT.log; exit 1; }
  python tools/step_breakdown.py $OUT/prof_$DT 3 > $OUT/resnet50_${DT}_step_breakdown_r3.md || exit 1
  head -12 $OUT/resnet50_${DT}_step_breakdown_r3.md
  rm -rf $OUT/prof_$DT
done
