#!/usr/bin/env bash
# kernel breakdown of the BERT-base fine-tuning step (batch 32, graph replay)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
DT=${1:-fp16}
rm -rf $OUT/prof_bert
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_bert -- python3 tools/bench_bert.py --dtype $DT --train-batches 32 --steps 8 --no-unfused --no-inference > $OUT/prof_bert.log 2>&1 || { tail -5 $OUT/prof_bert.log; exit 1; }
tail -2 $OUT/prof_bert.log | cut -c1-400
python tools/step_breakdown.py $OUT/prof_bert 3 > $OUT/bert_base_${DT}_step_breakdown.md || exit 1
head -45 $OUT/bert_base_${DT}_step_breakdown.md
rm -rf $OUT/prof_bert
