#!/bin/bash
# bench.py's pass with one lane and with two lanes at several cuts (FG_LANES, FG_LANE_SPLIT, FG_LANE_FRACS): ms per pass
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() {
  python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.3f Gbp/s, %.2f ms/pass, device %.2f ms, overlaps %d' % (j['value'], j['ms_per_step'], j['work']['device_ms_per_step'], j['work']['overlaps']))"
}
FG_LANES=1 run "one lane"
FG_LANES=2 FG_LANE_SPLIT=2 run "two lanes, 2 equal"
for f in "33,67" "40,60" "67,33" "25,50,25" "20,40,40" "30,40,30" "15,35,35,15" "10,30,30,30"; do
  FG_LANES=2 FG_LANE_FRACS=$f run "two lanes, $f"
done
