#!/bin/bash
set -u
O=$PWD/gpurun_out/r3u; mkdir -p $O
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; head -30 $O/c4_trace.txt
timeout 300 python tools/bench_c4.py --no-cpu-baseline > $O/c4.json 2> $O/c4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3u/c4.json'))
for k in d:
    if 'ms' in k and not isinstance(d[k], dict): print(k, d[k])
PY
