#!/bin/bash
# the headline line as the driver runs it (K = 20, W = 5) without the side records, twice, and rocprofv3's kernel
# statistics of the same command.   usage: bash tools/gpu_quick.sh <tag>
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
q="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-built-lattice --no-configs --no-reference-api --no-unmodified-caller"
for rep in 1 2; do
  timeout 200 python bench.py $q 2>$out/b_$rep.err | tail -n 1 > $out/b_$rep.json
  python -c "import json; d=json.load(open('$out/b_$rep.json')); print('rep $rep: value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'priming', d.get('priming_steps'), 'frac', round(d['roofline']['frac'],3), 'parity', d['parity_in_run']['ok'])"
done
root=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o c3 -- python $root/bench.py $q > $root/$out/prof.log 2>&1)
f=$(find $out/prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 12 "$f" | cut -c1-200
