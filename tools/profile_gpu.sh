#!/bin/bash
# rocprofv3 passes for profiles/: kernel stats, then one --pmc pass per HBM counter
# (never combined with trace domains other than --kernel-trace).  Run on the GPU box:
#   bash tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-run}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-unmodified-caller --no-configs --no-reference-api --no-built-lattice"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps 5 --warmup 2 > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_$c.log 2>&1
done
# one SQ pass (8 SQ slots): where the waves' cycles go (quad-cycles; MI355X_MICROARCH.md "rocprofv3 PMC slots")
SQC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
timeout -s KILL 400 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d $OUT/pmc_SQ -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_SQ.log 2>&1
Q=$(find $OUT/pmc_SQ -name "*counter_collection.csv" | head -1)
[ -n "$Q" ] && python $REPO/tools/pmc_summary.py $OUT/pmc_sq.json $Q > $OUT/pmc_sq_summary.txt
cd $REPO
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats.csv
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/pmc_FETCH_SIZE.csv
[ -n "$W" ] && cp $W $OUT/pmc_WRITE_SIZE.csv
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py $OUT/pmc_hbm.json $F $W > $OUT/pmc_summary.txt
# keep the merge-back small: drop the raw rocprof trees
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ
head -12 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt $OUT/pmc_sq_summary.txt
