#!/bin/bash
set -u
O=$PWD/gpurun_out/r3y; mkdir -p $O
run() {
  env "$@" GTN_BENCH_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-unmodified-caller --no-built-lattice > $O/b.json 2> $O/b.err
  echo "== $*"; grep "vector step host" $O/b.err | tail -4
  python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3y/b.json') if l.startswith('{')][-1])
r=d['reference_api']; print('vector', r['vector_overloads'].get('ms_per_batch'), 'loop', r['reference_loop'].get('ctcBatched_ms'))
PY
}
run A=1
run GTNX_RECLAIMERS=0
run GTN_AMD_THREADS=8
run GTNX_NO_MALLOPT=1
