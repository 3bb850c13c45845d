#!/bin/bash
# SQ counters of the alignment kernels:  tools/ksw_pmc.sh <tag> [pairs] [length]   (on the GPU box)
set -e
TAG=${1:-ksw}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export KSW_BENCH_ONLY=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/tools/ksw_bench.py ${2:-2000} ${3:-10000} > $OUT/sq.out 2> $OUT/sq.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/ksw_bench.py ${2:-2000} ${3:-10000} > $OUT/sq2.out 2> $OUT/sq2.err
cd $ROOT
python3 tools/pmc_sq.py $OUT/sq $OUT/sq_counters.json > $OUT/sq_summary.txt
python3 tools/pmc_sq.py $OUT/sq2 $OUT/sq2_counters.json > $OUT/sq2_summary.txt || true
rm -rf $OUT/sq $OUT/sq2
