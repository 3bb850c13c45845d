#!/bin/bash
# rocprofv3 evidence for a named workload other than the bench's (tools/run_workload.py: dmel_ont30, hifi30, ...):
#   tools/profile_workload.sh <workload> <scale> <tag>
#   1. --kernel-trace --stats     per-kernel time of index build + 2 overlap passes
#   2. --pmc SQ_*                 instruction counts and wait fractions per kernel (own pass, kernel-trace only)
# The program itself follows "--"; aggregates land in gpurun_out/prof_<tag>/ (copy what is to be judged to profiles/).
set -e
WL=${1:-hifi30}; SCALE=${2:-1.0}; TAG=${3:-r03_$WL}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/run_workload.py $WL $SCALE 50 > $OUT/run_under_rocprof.txt 2> $OUT/stats.err
echo "stats done" > $OUT/progress
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/tools/run_workload.py $WL $SCALE 0 > /dev/null 2> $OUT/sq.err
echo "sq done" >> $OUT/progress
cd $ROOT
python3 tools/pmc_sq.py $OUT/sq $OUT/sq_counters.json > $OUT/sq_summary.txt
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/sq
echo "all done" >> $OUT/progress
