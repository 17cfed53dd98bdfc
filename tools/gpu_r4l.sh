#!/bin/bash
set -u
O=$PWD/gpurun_out/r4l; mkdir -p $O
for i in 1 2 3 4 5 6; do
  GTNX_HOST_TIMING=1 timeout 200 python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller --no-built-lattice > $O/b$i.json 2> $O/b$i.err
  python - $O/b$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('run', sys.argv[1][-7:], round(d['value']), round(d['ms_per_step'],3), d['host_ms_last_step'])
PY
  grep "pool miss\|drain_deferred\|drain_while\|empty" $O/b$i.err | head -5
done
