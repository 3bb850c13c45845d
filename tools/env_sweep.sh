# usage: bash tools/env_sweep.sh "<kernel substring>" "VAR=val" "VAR=val VAR2=val" ...
pat=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --no-cpu --steps 4 2>/dev/null | tail -1 > gpurun_out/env_$i.json
  python -c "
import json; d=json.loads(open('gpurun_out/env_$i.json').read()); print('$e', d['value'], {k:v for k,v in d['work']['kernel_ms_per_step'].items() if '$pat' in k})"
done
