#!/bin/bash
set -u
O=$PWD/gpurun_out/r4h; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/c4_trace.py > $O/c4_trace.txt 2>&1; head -9 $O/c4_trace.txt
timeout 300 python tools/bench_c4.py --no-cpu-baseline > $O/c4.json 2> $O/c4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h/c4.json'))
for k in d:
    if 'ms' in k and not isinstance(d[k], dict): print(k, d[k])
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-built-lattice > $O/b.json 2> $O/b.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4h/b.json') if l.startswith('{')][-1])
r=d['reference_api']; print('vector', r['vector_overloads'].get('ms_per_batch'), 'loop', r['reference_loop'].get('ctcBatched_ms'), 'host-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
print('unmodified', json.dumps(d.get('unmodified_caller'))[:700])
print(d['value'], d['ms_per_step'])
PY
