#!/bin/bash
set -u
O=$PWD/gpurun_out/r3x; mkdir -p $O
GTN_BENCH_TIMING=1 GTNX_HOST_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-unmodified-caller --no-built-lattice > $O/bench.json 2> $O/bench.err
grep "vector step host" $O/bench.err | tail -12
grep -i "region\.\|host timer\|\[gtnx\] host" $O/bench.err | tail -40
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3x/bench.json') if l.startswith('{')][-1])
print(json.dumps(d.get('reference_api'))[:1500])
PY
