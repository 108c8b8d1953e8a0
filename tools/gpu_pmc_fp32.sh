#!/usr/bin/env bash
# Hardware counters per kernel of the fp32 training step: three rocprofv3
# --pmc passes (each a run of its own) and the table of tools/pmc_table.py.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
DT=${1:-fp32}
KIND=f32; [ "$DT" != fp32 ] && KIND=f16
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_${KIND^^} SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
B="FETCH_SIZE TCC_HIT_sum"
C="WRITE_SIZE TCC_MISS_sum"
# plans are measured without the profiler first, so the counted runs replay
timeout -k 10 200 python3 bench.py --dtype $DT --steps 2 --warmup 0 > $OUT/pmc_warm.log 2>&1 || { tail -5 $OUT/pmc_warm.log; exit 1; }
i=0
for SET in "$A" "$B" "$C"; do
  i=$((i + 1))
  rm -rf $OUT/pmc_$i
  timeout -s KILL 280 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$i -- python3 bench.py --dtype $DT --steps 2 --warmup 0 > $OUT/pmc_$i.log 2>&1 || { tail -5 $OUT/pmc_$i.log; exit 1; }
  echo "pass $i done"
done
python tools/pmc_table.py $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $KIND > $OUT/resnet50_${DT}_pmc_counters_winograd.md || exit 1
head -40 $OUT/resnet50_${DT}_pmc_counters_winograd.md
rm -rf $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3
