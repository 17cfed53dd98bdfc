#!/bin/bash
# A/B on the MI355X: uploads of pinned tables by the runtime's copy (default: kernels up to 4 KB only) against the
# engine's own copy kernel up to 1 MB, on the headline step.   usage: bash tools/gpu_ab_h2d.sh <tag>
tag=${1:-ab}
out=gpurun_out/$tag
mkdir -p $out
q="--steps 300 --warmup 20 --no-cpu-baseline --no-built-lattice --no-configs --no-reference-api --no-unmodified-caller"
for rep in 1 2; do
  for lim in 4096 1000000; do
    GTNX_H2D_KERNEL_BYTES=$lim timeout 200 python bench.py $q 2>$out/b_${lim}_$rep.err | tail -n 1 > $out/b_${lim}_$rep.json
    python -c "import json; d=json.load(open('$out/b_${lim}_$rep.json')); print('lim $lim rep $rep: value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4))"
  done
done
