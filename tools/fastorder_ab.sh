#!/bin/bash
# A/B of the tie-free score order (bitonic network) in the chaining stage's finish step: FG_ABLATE=64 switches it off
# (results stay right: the emulation is the fallback it replaces)
set -e
for a in 64 0; do
  echo "== FG_ABLATE=$a (64: fast score order off)"
  FG_ABLATE=$a timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-assemble-stage > gpurun_out/fastorder_$a.json 2> gpurun_out/fastorder_$a.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/fastorder_$a.json").read().strip().splitlines()[-1])
k=j["work"]["kernel_ms_per_step"]
print(j["value"], j["ms_per_step"], j["work"]["overlaps"], {x:k[x] for x in k if x.startswith("k_chain")}, j["roofline"]["dominant_by"][-160:])
PY
done
