#!/bin/bash
set -u
O=$PWD/gpurun_out/r4k; mkdir -p $O
GTNX_HOST_TIMING=1 GTN_BENCH_TIMING=1 timeout 300 python bench.py --config c5 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller --no-built-lattice > $O/c5.json 2> $O/c5.err
grep "gtnx host" $O/c5.err | sort -k6 -n -r | head -14
grep "step host\|host ms" $O/c5.err | tail -4
python -c "
import json
d=json.loads([l for l in open('$O/c5.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('host_ms_last_step'))"
GTNX_COMPOSE_STATS=1 GTNX_SYNC_COMPOSE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller > $O/b.json 2> $O/b.err
grep "compose:" $O/b.err | tail -3
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); b=d['built_lattice_path']; print(b['ms_per_step'], {k:(round(v['ms_per_launch'],3), round(v['frac'],3)) for k,v in b['roofline'].items()})"
