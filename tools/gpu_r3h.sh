#!/bin/bash
set -u
O=gpurun_out/r3h; mkdir -p $O
B=tests/dropin/_bin
for t in 0 8 12 16 24; do
  echo "== device C=256 threads $t (0 = default)" | tee -a $O/sweep.log
  GTN_AMD_THREADS=$t BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 300 device 2>&1 | tee -a $O/sweep.log
done
for t in 8 16 32; do
echo "== timing threads $t" | tee -a $O/timing.log
GTN_AMD_THREADS=$t GTNX_HOST_TIMING=1 BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 300 device 2>&1 | grep -v "batch\.\|gradsink" | tee -a $O/timing.log
done
