#!/bin/bash
set -u
O=$PWD/gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "wide or golden_compose or asg or non_layered" 2>&1 | tail -15
timeout 200 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -6
timeout 300 tests/dropin/_bin/bm_functions 2>&1 | tail -4
