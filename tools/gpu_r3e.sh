#!/bin/bash
set -u
O=gpurun_out/r3e; mkdir -p $O
B=tests/dropin/_bin
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 $B/bm_ctc_c256 512 256 20 device check > $O/check_device.log 2>&1; echo "rc $?" >> $O/check_device.log; cat $O/check_device.log
for t in 0 16 24 48; do
  echo "== device C=256 threads $t (0 = default)" | tee -a $O/sweep.log
  GTN_AMD_THREADS=$t BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 200 device 2>&1 | tee -a $O/sweep.log
done
echo "== host alphabet 28" | tee -a $O/sweep.log
BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 28 200 host 2>&1 | tee -a $O/sweep.log
echo "== timing table" | tee -a $O/sweep.log
GTNX_HOST_TIMING=1 BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 200 device 2>&1 | tee -a $O/sweep.log
timeout 600 $B/bm_ctc 512 2>&1 | tee -a $O/sweep.log
