#!/bin/bash
# bench.py's pass with one lane and with two lanes at several cuts (FG_LANES, FG_LANE_MIN_HITS, FG_LANE_SPLIT, FG_LANE_FRACS)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() {
  python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.3f Gbp/s, %.2f ms/pass, device %.2f ms, overlaps %d' % (j['value'], j['ms_per_step'], j['work']['device_ms_per_step'], j['work']['overlaps']))"
}
FG_LANES=1 run "one lane"
FG_LANE_MIN_HITS=1 FG_LANE_SPLIT=2 run "two lanes, 2 equal"
FG_LANE_MIN_HITS=1 FG_LANE_SPLIT=3 run "two lanes, 3 equal"
FG_LANE_MIN_HITS=1 FG_LANE_FRACS=40,60 run "two lanes, 40,60"
FG_LANE_MIN_HITS=1 FG_LANE_FRACS=60,40 run "two lanes, 60,40"
