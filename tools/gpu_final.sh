#!/bin/bash
# end-of-round check: the whole GPU suite, smoke, the default bench line (timed), kept under gpurun_out/final
set -u
O=$PWD/gpurun_out/final; mkdir -p $O
[ -z "${SKIP_TESTS:-}" ] && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench wall ${SECONDS}s"; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/final/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', d['roofline']['frac'])
r=d['reference_api']; print('vector', r['vector_overloads'].get('ms_per_batch'), 'loop', r['reference_loop'].get('ctcBatched_ms'), 'host-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
u=d['unmodified_caller']; print('unmodified', u['ctcBatched_ms'], u['other_timings_ms'], u.get('functions_benchmark_ms'))
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('unit'), (v.get('cpu_baseline') or {}).get('value'))
b=d['built_lattice_path']; print('built', b['ms_per_step'], {k:(round(v['ms_per_launch'],3), round(v['frac'],3)) for k,v in b['roofline'].items()})
print('cpu', d['cpu_baseline']['value'])
PY
