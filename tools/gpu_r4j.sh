set -u
O=$PWD/gpurun_out/r4j; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest.log
for i in 1 2 3; do $BM 512 256 100 device >> $O/bm.log 2>&1; done
GTN_BENCH_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-unmodified-caller --no-configs --steps 50 > $O/bench.json 2> $O/bench.err
cat $O/pytest.log; cat $O/bm.log
grep "vector step host" $O/bench.err | tail -3
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4j/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
r=d['reference_api']; print('vector', r['vector_overloads']['ms_per_batch'], '\nloop', r['reference_loop']['ctcBatched_ms'], '\nhost-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
b=d['built_lattice_path']; print('built', b['ms_per_step'], {k:(round(v['ms_per_launch'],3), round(v['frac'],3)) for k,v in b['roofline'].items()})
PY
