#!/bin/bash
set -u
O=gpurun_out/r3b; mkdir -p $O
B=tests/dropin/_bin
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 $B/bm_ctc_c256 512 256 20 device check > $O/check_device.log 2>&1; echo "rc $?" >> $O/check_device.log; cat $O/check_device.log
timeout 300 $B/bm_ctc_c256 512 28 20 host check > $O/check_host28.log 2>&1; echo "rc $?" >> $O/check_host28.log; cat $O/check_host28.log
for t in 16 32 64 128 256; do for d in 0 8; do
  echo "== device C=256 threads $t drainers $d" | tee -a $O/sweep.log
  GTN_AMD_THREADS=$t GTNX_DRAIN_THREADS=$d BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 50 device 2>&1 | tee -a $O/sweep.log
done; done
for t in 16 32 64 256; do
  echo "== host alphabet 28 threads $t" | tee -a $O/sweep.log
  GTN_AMD_THREADS=$t BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 28 50 host 2>&1 | tee -a $O/sweep.log
done
echo "== timing table, 32 threads" | tee -a $O/sweep.log
GTN_AMD_THREADS=32 GTNX_HOST_TIMING=1 BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 50 device 2>&1 | tee -a $O/sweep.log
GTN_AMD_THREADS=32 timeout 600 $B/bm_ctc 512 2>&1 | tee -a $O/sweep.log
