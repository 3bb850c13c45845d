# usage: bash tools/var_sweep.sh <kernel-name-substring> <variant>...   (variant libs flye_amd/lib/var_<variant>.so)
pat=$1; shift
for v in "$@"; do
  FLYE_GPU_LIB=$PWD/flye_amd/lib/var_$v.so timeout -k 10 200 python bench.py --no-cpu --steps 4 2>/dev/null | tail -1 > gpurun_out/var_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/var_$v.json').read()); print('$v', d['value'], {k:v for k,v in d['work']['kernel_ms_per_step'].items() if '$pat' in k})"
done
