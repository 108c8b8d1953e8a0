#!/usr/bin/env bash
# kernel breakdown of the fp16 training step at batch 256 (graph replay)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
DT=${1:-fp16}
B=${2:-256}
rm -rf $OUT/prof_b
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_b -- python3 bench.py --steps 6 --warmup 3 --dtype $DT --batch $B > $OUT/prof_b.log 2>&1 || { tail -5 $OUT/prof_b.log; exit 1; }
tail -1 $OUT/prof_b.log
python tools/step_breakdown.py $OUT/prof_b 3 > $OUT/resnet50_${DT}_b${B}_step_breakdown.md || exit 1
head -40 $OUT/resnet50_${DT}_b${B}_step_breakdown.md
rm -rf $OUT/prof_b
