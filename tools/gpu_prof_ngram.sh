#!/bin/bash
set -u
O=$PWD/gpurun_out/ngram; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $OLDPWD/tests/dropin/_bin/bm_ctc 8 > $O/run.log 2>&1
cat $O/run.log | tail -8
S=$(find $O/prof -name "*kernel_stats.csv" | head -1)
head -25 $S | cut -c 1-220
cp $S $O/kernel_stats.csv; rm -rf $O/prof
