#!/bin/bash
# The scaling curve of bench.py on one node: N = 1, 2, 4, 8 ranks, weak (genome grows with N, per-rank queries fixed)
# and strong (the E. coli workload itself split over the ranks).  For the day an 8-GPU node is available: nothing here
# has been measured on more than one GPU (the driver's SCALE_rNN.json was skipped in every round so far).
#   tools/scale.sh [outdir] [steps] [warmup]
# torch.distributed.run is started BEFORE anything touches a GPU (one fresh launcher per N); every rank prints through
# rank 0 the one JSON line of bench.py, collected in <outdir>/scale_<mode>_<N>.json.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/scale}
STEPS=${2:-5}
WARM=${3:-1}
mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
PORT=29600
for MODE in weak strong; do
  for N in 1 2 4 8; do
    [ "$N" -le "$NGPU" ] || { echo "skip N=$N ($NGPU GPUs visible)"; continue; }
    PORT=$((PORT + 1))
    if [ "$N" -eq 1 ]; then
      python3 "$ROOT/bench.py" --gpus 1 --steps "$STEPS" --warmup "$WARM" --scaling $MODE --no-cpu > "$OUT/scale_${MODE}_$N.json"
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT \
        "$ROOT/bench.py" --gpus "$N" --steps "$STEPS" --warmup "$WARM" --scaling $MODE > "$OUT/scale_${MODE}_$N.json"
    fi
    python3 - "$OUT/scale_${MODE}_$N.json" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
j = json.loads(l)
print(f"{j['scaling']:6s} N={j['n_gpus']}: {j['value']:.3f} Gbp/s, {j['ms_per_step']:.1f} ms/step, index build {j['work']['index_build_wall_s']:.3f} s, "
      f"collectives {j['work']['index_build_collectives']}")
PY
  done
done
