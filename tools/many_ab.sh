#!/bin/bash
# A/B of the count-adaptive partition form of the hit sort's levels (FG_SORT_MANY_MIN=0: fixed thresholds)
set -e
for m in 0 4096; do
  echo "== FG_SORT_MANY_MIN=$m (bench workload)"
  FG_SORT_MANY_MIN=$m timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-assemble-stage > gpurun_out/many_ab_$m.json 2> gpurun_out/many_ab_$m.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/many_ab_$m.json").read().strip().splitlines()[-1])
k=j["work"]["kernel_ms_per_step"]
print(j["value"], j["ms_per_step"], {x:k[x] for x in k if x.startswith("k_sort")})
PY
done
for m in 0 4096; do
  echo "== FG_SORT_MANY_MIN=$m (dmel_ont30 x 0.25)"
  FG_LANES=1 FG_SORT_MANY_MIN=$m timeout -k 10 400 python tools/run_workload.py dmel_ont30 0.25 40 2>&1 | grep -E "pass 1|k_sort|identical"
done
