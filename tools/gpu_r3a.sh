#!/bin/bash
# round 3, first GPU pass: parity suite + the reference-API loop at C=256 / alphabet 28
set -u
mkdir -p gpurun_out/r3a
nproc > gpurun_out/r3a/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3a/pytest.log
tail -15 gpurun_out/r3a/pytest.log
B=tests/dropin/_bin
for mode in device host; do
  BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 50 $mode check > gpurun_out/r3a/bm_c256_$mode.log 2>&1
  echo "rc $?" >> gpurun_out/r3a/bm_c256_$mode.log
  cat gpurun_out/r3a/bm_c256_$mode.log
done
BM_PHASES=1 GTNX_HOST_TIMING=1 timeout 300 $B/bm_ctc_c256 512 256 50 device > gpurun_out/r3a/bm_c256_timing.log 2>&1
tail -20 gpurun_out/r3a/bm_c256_timing.log
BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 28 50 host > gpurun_out/r3a/bm_c28_host.log 2>&1; cat gpurun_out/r3a/bm_c28_host.log
timeout 600 $B/bm_ctc 512 > gpurun_out/r3a/bm_ctc_512.log 2>&1; cat gpurun_out/r3a/bm_ctc_512.log
timeout 300 $B/gather_bench 512 > gpurun_out/r3a/gather_bench.log 2>&1; cat gpurun_out/r3a/gather_bench.log
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/r3a/bench.log 2>&1; tail -3 gpurun_out/r3a/bench.log
