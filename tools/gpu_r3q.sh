#!/bin/bash
# wide chain products: parity, then the reference's benchmarks
set -u
O=$PWD/gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "wide_chain or golden_compose or compose_linear or asg" 2>&1 | tail -15
timeout 200 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -8
timeout 200 tests/dropin/_bin/bm_functions 2>&1 | tail -12
GTNX_COMPOSE_STATS=1 timeout 100 tests/dropin/_bin/bm_ctc 8 2>&1 | grep compose | sort | uniq -c | sort -rn | head
