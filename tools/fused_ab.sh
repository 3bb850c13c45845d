#!/bin/bash
# bench pass with the one-kernel small-group chaining path off / on at several caps
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() {
python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['work']['kernel_ms_per_step']
print('$1: %.3f Gbp/s, %.2f ms/pass, device %.2f ms, overlaps %d' % (j['value'], j['ms_per_step'], j['work']['device_ms_per_step'], j['work']['overlaps']), {n: v for n, v in k.items() if n.startswith(('k_chain','k_group_prep'))}, j['stages']['chain']['exclusive_ms'])"
}
FG_CHAIN_FUSED=0 run "three kernels"
for cap in 96 128 160 192 256 320 448; do FG_FUSED_CAP=$cap run "fused cap $cap"; done
