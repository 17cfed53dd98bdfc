#!/bin/bash
# C4 with the beta sweep beside the alpha sweep: tests, kernel stats (gpurun_out/c4v2/)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/c4v2; mkdir -p $OUT
timeout 500 python -m pytest tests/test_lazy_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -- python $REPO/tools/bench_c4.py --steps 1 --no-cpu-baseline > $OUT/c4.log 2>&1
s=$(find $OUT/tmp -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/c4_kernel_stats.csv
rm -rf $OUT/tmp
cd $REPO
head -6 $OUT/c4_kernel_stats.csv | cut -c1-220
