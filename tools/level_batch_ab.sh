#!/bin/bash
# the sort's level loop with a count read-back per level (1) and every 4 / 8 levels: bench pass and small calls
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for B in 1 4 8; do
FG_SORT_LEVEL_BATCH=$B python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('level batch $B: %.3f Gbp/s, %.2f ms/pass, device %.2f ms, overlaps %d, sort stage %.2f ms' % (j['value'], j['ms_per_step'], j['work']['device_ms_per_step'], j['work']['overlaps'], j['stages']['hit_sort']['exclusive_ms']))"
FG_SORT_LEVEL_BATCH=$B python3 $ROOT/tools/small_batch.py 2>/dev/null | cut -c1-120
done
