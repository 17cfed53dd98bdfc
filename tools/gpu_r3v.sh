#!/bin/bash
set -u
O=$PWD/gpurun_out/r3v; mkdir -p $O
GTNX_SYNC_COMPOSE=1 GTNX_COMPOSE_STATS=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-reference-api --no-unmodified-caller > $O/bench.json 2> $O/bench.err
grep "compose:" $O/bench.err | sort | uniq -c | sort -rn | head -5
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3v/bench.json') if l.startswith('{')][-1])
print(json.dumps(d.get('built_lattice_path'), indent=0)[:1500])
PY
bash tools/gpu_prof_ngram.sh 2>&1 | grep -v "^[WE]2026" | head -22
