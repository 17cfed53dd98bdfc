#!/bin/bash
set -u
O=$PWD/gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "wide or golden_compose or asg" 2>&1 | tail -4
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; grep "compose" $O/c4_trace.txt
timeout 100 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -5
timeout 100 tests/dropin/_bin/bm_functions 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
