#!/bin/bash
# The rocprofv3 evidence behind bench.py's numbers, on the GPU box:  tools/profile_round.sh <tag>
#   1. --kernel-trace --stats        per-kernel time of index build + 4 overlap passes
#   2. --pmc FETCH_SIZE              fabric-side read bytes per kernel   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE              fabric-side written bytes per kernel (own pass)
#   4. --pmc SQ_*                    instruction counts and wait fractions per kernel
# Counter passes run with --kernel-trace only (no other trace domain), the program itself follows "--".
# Everything lands in gpurun_out/prof_<tag>/; the aggregated files are what gets copied to profiles/.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done" > $OUT/progress
export FLYE_BENCH_NO_SERIAL_PASS=1   # the counter passes: index build + exactly ONE overlap pass
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $OUT/fetch.err
echo "fetch done" >> $OUT/progress
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $OUT/write.err
echo "write done" >> $OUT/progress
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $OUT/sq.err
echo "sq done" >> $OUT/progress
cd $ROOT
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) around python3 bench.py --steps 1 --warmup 0 --no-cpu (index build + ONE overlap pass, E.coli PB 50x workload, detector minOverlap 1000), state $TAG; aggregated by tools/pmc_traffic.py. Counter values are KiB x 1024, summed over all launches of the kernel in that pass (bytes_per_pass) and divided by the launch count (bytes_per_launch). On gfx950 FETCH_SIZE under-counts wide coalesced reads by up to 2x (MI355X_MICROARCH.md, HBM) and is uncalibrated for the 4-8 B/lane accesses of these kernels: read the sums as lower bounds." > $OUT/traffic_summary.txt
python3 tools/pmc_sq.py $OUT/sq $OUT/sq_counters.json > $OUT/sq_summary.txt
# the per-kernel stats file of pass 1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq   # raw traces are large; the aggregates stay
echo "all done" >> $OUT/progress
