#!/usr/bin/env bash
# Scaling curve of the headline benchmark on the GPUs of this node: ResNet-50
# training throughput and communication time per iteration at 1, 2, 4, 8 GPUs,
# the two series of the reference's figure (doc/_static/images/benchmark.png,
# produced by examples/cifar_distributed_cnn/benchmark.py:34-92).
#
#   tools/run_scaling.sh [--steps K] [--warmup W] [--dtype fp32|fp16|bf16]
#                        [--out DIR] [--max-gpus N] [--two-phase ELEMENTS]
#                        [--xgmi-direct]
#
# --xgmi-direct sums the buckets with the kernels that read every peer's
# window over its own xGMI link (csrc/hip/xgmi.hip) instead of RCCL's ring.
# --two-phase E sums buckets of at least E elements as reduce-scatter +
# all-gather (every slice over its own xGMI link) instead of one all-reduce;
# run the script once with and once without to see which this node prefers.
#
# One process per GPU, started by torch.distributed.run BEFORE anything touches
# a GPU (the launcher itself never initialises HIP).  Every N that the node
# does not have is skipped, so the script runs unchanged on a 1-GPU box.
# Writes   DIR/scaling_curve.jsonl   one bench.py JSON line per N
#          DIR/scaling_comm_time.md  throughput, efficiency and Dist_* time
set -u
cd "$(dirname "$0")/.."
STEPS=100; WARMUP=10; DTYPE=fp32; OUT=profiles; MAXG=8; PORT=29671
while [ $# -gt 0 ]; do
  case "$1" in
    --steps) STEPS=$2; shift 2;;
    --warmup) WARMUP=$2; shift 2;;
    --dtype) DTYPE=$2; shift 2;;
    --out) OUT=$2; shift 2;;
    --max-gpus) MAXG=$2; shift 2;;
    --two-phase) export SINGA_ALLREDUCE_TWO_PHASE=$2; shift 2;;
    --xgmi-direct) export SINGA_XGMI_DIRECT=1; shift;;
    *) echo "unknown argument $1" >&2; exit 2;;
  esac
done
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# device_count() reads the topology without creating a HIP context
HAVE=$(python -c 'import torch; print(torch.cuda.device_count())')
CURVE="$OUT/scaling_curve.jsonl"; : > "$CURVE"
LOGDIR="$OUT/scaling_logs"; mkdir -p "$LOGDIR"
for N in 1 2 4 8; do
  if [ "$N" -gt "$HAVE" ] || [ "$N" -gt "$MAXG" ]; then
    echo "skipping N=$N (node has $HAVE GPUs)"; continue
  fi
  LOG="$LOGDIR/bench_n$N.log"
  echo "== N=$N =="
  if [ "$N" -eq 1 ]; then
    timeout -k 10 900 python bench.py --gpus 1 --steps "$STEPS" \
      --warmup "$WARMUP" --dtype "$DTYPE" --verbosity 3 > "$LOG" 2>&1
  else
    timeout -k 10 900 python -m torch.distributed.run --nnodes=1 \
      --nproc-per-node "$N" --master-addr 127.0.0.1 \
      --master-port $((PORT + N)) bench.py --gpus "$N" --steps "$STEPS" \
      --warmup "$WARMUP" --dtype "$DTYPE" --verbosity 3 > "$LOG" 2>&1
  fi
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "N=$N failed (exit $rc); see $LOG" >&2
    tail -5 "$LOG" >&2
    break     # nothing more on the GPUs after a failed step
  fi
  grep '^{"metric"' "$LOG" | tail -1 >> "$CURVE"
done
# all-reduce latency / bandwidth over message sizes on all GPUs of the node:
# RCCL (one call, reduce-scatter + all-gather, graph replay) next to the
# direct xGMI kernels
if [ "$HAVE" -gt 1 ]; then
  N=$HAVE; [ "$N" -gt "$MAXG" ] && N=$MAXG
  SINGA_XGMI_DIRECT=1 timeout -k 10 600 python -m torch.distributed.run \
    --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
    --master-port $((PORT + 20)) tools/bench_allreduce.py \
    > "$LOGDIR/allreduce_n$N.log" 2>&1 \
    && grep '^{' "$LOGDIR/allreduce_n$N.log" > "$OUT/allreduce_sizes_n$N.jsonl"
fi
python - "$CURVE" "$LOGDIR" "$OUT/scaling_comm_time.md" <<'EOF'
import json, os, re, sys
curve, logdir, out = sys.argv[1:4]
rows = [json.loads(l) for l in open(curve) if l.strip()]
if not rows:
    sys.exit("no successful run")
base = rows[0]["value"] / rows[0]["n_gpus"]
lines = ["# ResNet-50 DistOpt scaling on this node (%s, batch %d / GPU)" % (
             rows[0]["dtype"], rows[0]["config"]["per_gpu_batch"]), "",
         "| GPUs | images/s | ms/step | efficiency vs 1 GPU | "
         "communication s/iteration (sum of Dist_* nodes, rank 0) | "
         "collectives in hipGraph |", "|---|---|---|---|---|---|"]
for r in rows:
    n = r["n_gpus"]
    comm = 0.0
    log = os.path.join(logdir, "bench_n%d.log" % n)
    for m in re.finditer(r"^OP_ID\d+\. (\S*Dist\S*) : ([0-9.eE+-]+) sec",
                         open(log).read(), re.M):
        if "wait" not in m.group(1):
            comm += float(m.group(2))
    lines.append("| %d | %.1f | %.3f | %.1f %% | %.6f | %s |" % (
        n, r["value"], r["ms_per_step"], 100.0 * r["value"] / (base * n),
        comm, r["config"]["collectives_in_graph"]))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
EOF
