#!/bin/bash
# A/B on the MI355X: an event per released pinned block (GTNX_PINNED_STAMP_EVERY=1) against one per eight, on the
# reference's loop (tests/dropin/_bin/bm_ctc_c256) and on the headline step.   usage: bash tools/gpu_ab_pinned.sh <tag>
tag=${1:-abp}
out=gpurun_out/$tag
mkdir -p $out
cpus=$(python - <<PY
import bench
c, n = bench.gpu_local_cpus(0)
print(",".join(str(x) for x in c) if c else "")
PY
)
for rep in 1 2 3; do
  for ev in 1 8; do
    if [ -n "$cpus" ]; then r=$(GTNX_PINNED_STAMP_EVERY=$ev taskset -c $cpus tests/dropin/_bin/bm_ctc_c256 512 256 300 device 2>/dev/null | tail -n 1)
    else r=$(GTNX_PINNED_STAMP_EVERY=$ev tests/dropin/_bin/bm_ctc_c256 512 256 300 device 2>/dev/null | tail -n 1); fi
    echo "loop every=$ev rep=$rep: $r" | cut -c1-200 | tee -a $out/loop.txt
  done
done
q="--gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-built-lattice --no-configs --no-reference-api --no-unmodified-caller"
for ev in 1 8; do
  GTNX_PINNED_STAMP_EVERY=$ev timeout 200 python bench.py $q 2>/dev/null | tail -n 1 > $out/b_$ev.json
  python -c "import json; d=json.load(open('$out/b_$ev.json')); print('headline every=$ev: value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4))" | tee -a $out/loop.txt
done
