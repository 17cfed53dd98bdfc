#!/bin/bash
set -u
O=$PWD/gpurun_out/r4a; mkdir -p $O
GTN_BENCH_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-built-lattice > $O/b.json 2> $O/b.err
grep "vector step host" $O/b.err | tail -4
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4a/b.json') if l.startswith('{')][-1])
r=d['reference_api']; print('vector', r['vector_overloads'].get('ms_per_batch'), 'loop', r['reference_loop'].get('ctcBatched_ms'), 'host-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
print('unmodified', json.dumps(d.get('unmodified_caller'))[:400])
print(d['value'], d['ms_per_step'])
PY
timeout 600 python -m pytest tests/test_dropin_gpu.py tests/test_batch_gpu.py -x -q -m gpu 2>&1 | tail -3
