#!/bin/bash
set -u
O=gpurun_out/r3k; mkdir -p $O
B=tests/dropin/_bin
for rep in 1 2; do for r in 1 2 4 8 64; do
echo "== reclaimers $r" | tee -a $O/sweep.log
GTNX_RECLAIMERS=$r timeout 300 $B/bm_ctc_c256 512 256 300 device 2>&1 | tee -a $O/sweep.log
GTNX_RECLAIMERS=$r bash tools/gpu_r3j.sh 2>&1 | grep "vector step ms" | tee -a $O/sweep.log
done; done
