#!/bin/bash
# A/B of the in-LDS mid tier of the hit sort (k_sort_mid): bench passes at several piece caps
set -e
for m in 0 2048 4096 8192; do
  echo "== FG_SORT_MID_MAX=$m"
  FG_SORT_MID_MAX=$m timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-assemble-stage > gpurun_out/mid_ab_$m.json 2> gpurun_out/mid_ab_$m.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/mid_ab_$m.json").read().strip().splitlines()[-1])
k=j["work"]["kernel_ms_per_step"]
print(j["value"], j["ms_per_step"], {x:k[x] for x in k if x.startswith("k_sort")})
PY
done
