#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the reference loop's step (tests/dropin/_bin/bm_ctc_c256 at C3) -> gpurun_out/prof_refloop/
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_refloop; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/tmp_$c -- $REPO/tests/dropin/_bin/bm_ctc_c256 512 256 3 device > $OUT/$c.log 2>&1
  s=$(find $OUT/tmp_$c -name "*counter_collection.csv" | head -1); [ -n "$s" ] && cp $s $OUT/pmc_$c.csv
  rm -rf $OUT/tmp_$c
done
cd $REPO
python tools/pmc_summary.py $OUT/pmc_hbm.json $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv | grep -E "band_|copy_seg"
rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
