#!/bin/bash
set -u
O=$PWD/gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_lazy_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 100 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -5
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; head -12 $O/c4_trace.txt
timeout 300 python tools/bench_c4.py --no-cpu-baseline > $O/c4.json 2> $O/c4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4f/c4.json'))
for k in d:
    if 'ms' in k and not isinstance(d[k], dict): print(k, d[k])
PY
