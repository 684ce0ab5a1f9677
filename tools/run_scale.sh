#!/bin/bash
# The scaling curve of the headline metric on one node: bench.py at 1, 2, 4 and 8 GPUs, one rank per GPU over RCCL (bench.py --gpus N
# starts its own ranks since round 6; this script loops over N and keeps the launcher form the driver uses)
# (weak scaling: 100000 columns on one GPU = BASELINE configs[1], 125000 per rank on several = the shard of configs[4];
# 8 GPUs = configs[4] itself, 1e6 columns).  One JSON line per run into $OUT (default gpurun_out/scale).
#   usage: bash tools/run_scale.sh [steps] [warmup]        (GPUS="1 2 4 8" overrides the list)
set -u
STEPS=${1:-20}; WARMUP=${2:-5}; OUT=${OUT:-gpurun_out/scale}; GPUS=${GPUS:-"1 2 4 8"}
cd "$(dirname "$0")/.." && mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in $GPUS; do
  if [ "$N" -gt "$NDEV" ]; then echo "{\"n_gpus\": $N, \"skipped\": \"only $NDEV device(s) visible\"}" > "$OUT/scale_$N.json"; continue; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" > "$OUT/scale_$N.json" 2> "$OUT/scale_$N.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" > "$OUT/scale_$N.json" 2> "$OUT/scale_$N.err"
  fi
  python - "$OUT/scale_$N.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{r['n_gpus']} GPU(s): {r['value'] / 1e6:.3f} M columns/s, {r['ms_per_step']:.2f} ms/step, per-rank {r.get('per_rank_ms_per_step')}, "
          f"all-reduce {r.get('allreduce_ms_per_step')}")
except Exception as e:
    print("no result:", e)
PY
done
