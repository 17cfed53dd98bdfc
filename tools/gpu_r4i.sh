set -u
O=$PWD/gpurun_out/r4i; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest.log
run() { echo "== $*" >> $O/trace.log; env "$@" GTN_AMD_POOL_TRACE=1 BM_PHASES=1 $BM 512 256 100 device >> $O/trace.log 2>&1; }
run A=1
run A=2
run A=3
GTN_BENCH_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-unmodified-caller --no-configs --steps 50 > $O/bench.json 2> $O/bench.err
cat $O/pytest.log; cat $O/trace.log
grep "vector step host" $O/bench.err | tail -6
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4i/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
r=d['reference_api']; print('vector', r['vector_overloads'], '\nloop', r['reference_loop'], '\nhost-em', r['reference_loop_host_emissions'].get('ctcBatched_ms'))
b=d['built_lattice_path']; print('built', b['ms_per_step'], {k:(round(v['ms_per_launch'],3), round(v['frac'],3)) for k,v in b['roofline'].items()})
PY
