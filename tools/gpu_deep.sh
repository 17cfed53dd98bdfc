#!/bin/bash
set -u
O=$PWD/gpurun_out/deep; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "deep_thin or golden_shortest or rows_of_hundreds or known_answers or deep_narrow or non_layered" 2>&1 | tail -5
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- $R/tests/dropin/_bin/bm_functions > $O/bm_functions.log 2>&1
S=$(find $O/p2 -name "*kernel_stats.csv" | head -1); cp $S $O/bm_functions_kernel_stats.csv; rm -rf $O/p2
cd $R
head -8 $O/bm_functions_kernel_stats.csv | cut -c1-200
