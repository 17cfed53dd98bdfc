#!/bin/bash
# rocprofv3 kernel stats of the reference's two benchmark programs (unmodified, tests/dropin/_bin) and the C4 kernel
# trace: what profiles/r03_v2_bm_*_kernel_stats.csv and r03_v2_c4_kernel_trace.txt were made with.  Run on the GPU box.
set -u
O=$PWD/gpurun_out/compose_evidence; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for prog in bm_ctc bm_functions; do
  arg=""; [ $prog = bm_ctc ] && arg=8
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- $R/tests/dropin/_bin/$prog $arg > $O/$prog.log 2>&1
  S=$(find $O/p -name "*kernel_stats.csv" | head -1); cp $S $O/${prog}_kernel_stats.csv; rm -rf $O/p
  grep Timing $O/$prog.log | tail -20
  head -10 $O/${prog}_kernel_stats.csv | cut -c1-200
done
cd $R
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; head -14 $O/c4_trace.txt
