#!/bin/bash
set -e
for m in 10000 16384 24576; do
  echo "== FG_SORT_STREAM_MANY=$m (bench workload)"
  FG_SORT_STREAM_MANY=$m timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-assemble-stage > gpurun_out/many_ab_s$m.json 2> gpurun_out/many_ab_s$m.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/many_ab_s$m.json").read().strip().splitlines()[-1])
k=j["work"]["kernel_ms_per_step"]
print(j["value"], j["ms_per_step"], {x:k[x] for x in k if x.startswith("k_sort")})
PY
done
for m in 45000 90000 262144; do
  echo "== FG_SORT_STREAM_MANY=$m (dmel_ont30 x 0.25)"
  FG_LANES=1 FG_SORT_STREAM_MANY=$m timeout -k 10 400 python tools/run_workload.py dmel_ont30 0.25 40 2>&1 | grep -E "pass 1|identical" | cut -c1-200
  FG_LANES=1 FG_SORT_STREAM_MANY=$m timeout -k 10 400 python tools/run_workload.py dmel_ont30 0.25 40 2>&1 | grep -E "pass 1" | cut -c1-200
done
