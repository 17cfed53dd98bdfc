#!/bin/bash
set -u
O=$PWD/gpurun_out/r4d; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- $R/tests/dropin/_bin/bm_functions > $O/bm_functions.log 2>&1
S=$(find $O/p2 -name "*kernel_stats.csv" | head -1); cp $S $O/bm_functions_kernel_stats.csv; rm -rf $O/p2
cd $R
head -8 $O/bm_functions_kernel_stats.csv | cut -c1-200
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dropin_gpu.py -x -q -m gpu 2>&1 | tail -3
