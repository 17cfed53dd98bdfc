#!/bin/bash
set -u
O=$PWD/gpurun_out/r3w; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_lazy_gpu.py tests/test_batch_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 200 tests/dropin/_bin/bm_ctc 8 2>&1 | tail -6
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-api --no-unmodified-caller > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3w/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
b=d.get('built_lattice_path') or {}
for k,v in b.items():
    if isinstance(v, dict) and 'ms_per_launch' in v: print(k, v['kernel'], v['ms_per_launch'], v['frac'])
    elif not isinstance(v, dict): print(k, v)
c=d.get('configs') or {}
for k,v in c.items(): print(k, v.get('value'), v.get('unit'), v.get('ms_per_loss', v.get('ms_per_batch')))
PY
