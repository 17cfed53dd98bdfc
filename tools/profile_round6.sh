#!/bin/bash
# round-6 evidence for profiles/: run on the GPU box (bash tools/profile_round6.sh), results under gpurun_out/prof_r06/
#   1. the C3 timed loop of bench.py: kernel stats, FETCH_SIZE / WRITE_SIZE / SQ passes (tools/profile_gpu.sh)
#   2. the reference loop (tests/native/bm_ctc_c256.cpp): kernel stats
#   3. the C5-shaped band launches (bench.py --config c5): kernel stats + FETCH_SIZE / WRITE_SIZE
#   4. the built-lattice step at B = 512 (compose_kernel, sd_forward/backward_narrow): FETCH_SIZE / WRITE_SIZE
#   5. the Viterbi kernels at C3 (tools/ubench/viterbi_bench): kernel stats + FETCH_SIZE / WRITE_SIZE
#   6. C4 (tools/bench_c4.py --steps 1): kernel stats, FETCH_SIZE / WRITE_SIZE (the transitions-gradient kernel's re-reads) + one SQ pass
# Counter passes never carry another trace domain than --kernel-trace.
set -u
REPO=$(pwd)
TAG=${1:-r06v1}   # (a second pass of the round: bash tools/profile_round6.sh r05v2)
bash tools/profile_gpu.sh $TAG > /dev/null 2>&1
OUT=$REPO/gpurun_out/prof_r06; mkdir -p $OUT
cp gpurun_out/prof_$TAG/kernel_stats.csv $OUT/c3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_hbm.json $OUT/c3_pmc_hbm.json 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_sq.json $OUT/c3_pmc_sq.json 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_FETCH_SIZE.csv $OUT/c3_pmc_FETCH_SIZE.csv 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_WRITE_SIZE.csv $OUT/c3_pmc_WRITE_SIZE.csv 2>/dev/null
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/tmp_$name
}
pmc() {  # name, counters, command...
  local name=$1; local ctr=$2; shift; shift
  timeout -s KILL 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/tmp_$name -- "$@" > $OUT/${name}_pmc.log 2>&1
  local s=$(find $OUT/tmp_$name -name "*counter_collection.csv" | head -1); [ -n "$s" ] && cp $s $OUT/${name}.csv
  rm -rf $OUT/tmp_$name
}
B="python $REPO/bench.py --no-cpu-baseline --no-unmodified-caller --no-configs --no-reference-api"
stats bm_ctc_c256 $REPO/tests/dropin/_bin/bm_ctc_c256 512 256 10 device
stats c5 $B --no-built-lattice --config c5 --steps 3 --warmup 1
pmc c5_pmc_FETCH_SIZE FETCH_SIZE $B --no-built-lattice --config c5 --steps 2 --warmup 1
pmc c5_pmc_WRITE_SIZE WRITE_SIZE $B --no-built-lattice --config c5 --steps 2 --warmup 1
pmc built_pmc_FETCH_SIZE FETCH_SIZE $B --steps 1 --warmup 1
pmc built_pmc_WRITE_SIZE WRITE_SIZE $B --steps 1 --warmup 1
stats viterbi $REPO/tools/ubench/viterbi_bench
pmc viterbi_pmc_FETCH_SIZE FETCH_SIZE $REPO/tools/ubench/viterbi_bench
pmc viterbi_pmc_WRITE_SIZE WRITE_SIZE $REPO/tools/ubench/viterbi_bench
stats c4 python $REPO/tools/bench_c4.py --steps 1 --no-cpu-baseline
pmc c4_pmc_FETCH_SIZE FETCH_SIZE python $REPO/tools/bench_c4.py --steps 1 --no-cpu-baseline
pmc c4_pmc_WRITE_SIZE WRITE_SIZE python $REPO/tools/bench_c4.py --steps 1 --no-cpu-baseline
pmc c4_pmc_SQ "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" python $REPO/tools/bench_c4.py --steps 1 --no-cpu-baseline
cd $REPO
for n in c5 built viterbi c4; do
  F=$OUT/${n}_pmc_FETCH_SIZE.csv; W=$OUT/${n}_pmc_WRITE_SIZE.csv
  [ -f $F ] && [ -f $W ] && python tools/pmc_summary.py $OUT/${n}_pmc_hbm.json $F $W > $OUT/${n}_pmc_summary.txt
done
[ -f $OUT/c4_pmc_SQ.csv ] && python tools/pmc_summary.py $OUT/c4_pmc_sq.json $OUT/c4_pmc_SQ.csv > $OUT/c4_pmc_sq_summary.txt
# keep the merge-back small: the raw counter tables of the long runs stay on the box
rm -f $OUT/c4_pmc_SQ.csv $OUT/c4_pmc_FETCH_SIZE.csv $OUT/c4_pmc_WRITE_SIZE.csv $OUT/built_pmc_*.csv $OUT/c5_pmc_*.csv
ls -la $OUT | head -40
head -8 $OUT/c3_kernel_stats.csv
cat $OUT/*_pmc_summary.txt | head -40
