#!/bin/bash
# round-3 evidence for the compose kernels: kernel stats of the reference's two benchmark programs, the C4 trace,
# and the default bench line
set -u
O=$PWD/gpurun_out/r4c; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- $R/tests/dropin/_bin/bm_ctc 8 > $O/bm_ctc.log 2>&1
S=$(find $O/p1 -name "*kernel_stats.csv" | head -1); cp $S $O/bm_ctc_kernel_stats.csv; rm -rf $O/p1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- $R/tests/dropin/_bin/bm_functions > $O/bm_functions.log 2>&1
S=$(find $O/p2 -name "*kernel_stats.csv" | head -1); cp $S $O/bm_functions_kernel_stats.csv; rm -rf $O/p2
cd $R
grep Timing $O/bm_ctc.log $O/bm_functions.log | tail -30
head -14 $O/bm_ctc_kernel_stats.csv | cut -c1-200
head -8 $O/bm_functions_kernel_stats.csv | cut -c1-200
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; head -12 $O/c4_trace.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
