#!/bin/bash
set -u
O=$PWD/gpurun_out/r4g; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 100 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -5
python tools/bench_configs.py c1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1', d['value'], d['ms_per_loss'], d['kernels'].keys())"
