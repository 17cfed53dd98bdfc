set -u
O=$PWD/gpurun_out/r4k; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/pytest.log
timeout 600 python tools/bench_configs.py c3v > $O/c3v.json 2> $O/c3v.err
GTNX_VITERBI_WG=1 timeout 600 python tools/bench_configs.py c3v > $O/c3v_wg.json 2> $O/c3v_wg.err
cat $O/pytest.log; tail -3 $O/c3v.err
python - <<'PY'
import json
for f in ('c3v','c3v_wg'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r4k/{f}.json') if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'no json', e); continue
    s=d['symbolic_route']; print(f, 'path ms', round(s['viterbi_path_ms_per_batch'],3), 'score ms', round(s['viterbi_score_ms_per_batch'],3), 'roof', s['roofline'] and (round(s['roofline']['ms_per_launch'],4), round(s['roofline']['frac'],3)))
    print('  kernels', {k:round(v['ms_per_launch'],4) for k,v in s['kernels_path'].items()})
    b=d.get('built_route'); print('  built', b and (b['batch'], round(b['viterbi_path_ms_per_batch'],2), {k:(round(v['ms_per_launch'],3), v.get('roofline',{}).get('frac')) for k,v in b['kernels'].items()}, b.get('labels_equal_symbolic_route')))
    print('  parity', d.get('parity_in_run'), 'cpu', d.get('cpu_baseline'))
PY
