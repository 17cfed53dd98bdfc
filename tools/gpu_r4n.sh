set -u
O=$PWD/gpurun_out/r4n; mkdir -p $O; rm -f $O/*
SECONDS=0; timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? wall ${SECONDS}s"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4n/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'ref_api', d.get('value_reference_api'), d.get('ms_per_step_reference_api'))
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
print('parity', d.get('parity_in_run'))
r=d['reference_api']; print('vector', r['vector_overloads'].get('ms_per_batch'), 'loop', r['reference_loop'].get('ctcBatched_ms'), 'host-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
u=d['unmodified_caller']; print('unmodified', u.get('ctcBatched_ms'), u.get('other_timings_ms'), u.get('functions_benchmark_ms'))
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('unit'), (v.get('roofline') or {}).get('frac'), (v.get('cpu_baseline') or {}).get('value'), v.get('parity_in_run'), v.get('error'))
b=d['built_lattice_path']; print('built', b['ms_per_step'], {k:(round(v['ms_per_launch'],3), round(v['frac'],3)) for k,v in b['roofline'].items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
