#!/bin/bash
set -u
O=$PWD/gpurun_out/r3s; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_lazy_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 200 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -6
timeout 600 python -m pytest tests/test_dropin_gpu.py -x -q -m gpu 2>&1 | tail -5
