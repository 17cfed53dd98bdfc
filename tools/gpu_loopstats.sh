#!/bin/bash
# rocprofv3 kernel statistics of the reference's loop (tests/dropin/_bin/bm_ctc_c256) and of the vector-overload step.
tag=${1:-loop}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o loop -- $root/tests/dropin/_bin/bm_ctc_c256 512 256 300 device > $root/$out/loop.log 2>&1)
tail -n 1 $out/loop.log | cut -c1-200
f=$(find $out/prof -name "loop_kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 12 "$f" | cut -c1-180
