#!/bin/bash
set -u
O=gpurun_out/r3l; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
bash tools/gpu_r3j.sh 2>&1 | grep "vector step" | tail -4
timeout 300 tests/dropin/_bin/bm_ctc_c256 512 256 300 device 2>&1
