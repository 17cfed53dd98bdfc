#!/bin/bash
set -u
O=gpurun_out/r3c; mkdir -p $O
B=tests/dropin/_bin
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 $B/bm_ctc_c256 512 256 20 device check > $O/check_device.log 2>&1; echo "rc $?" >> $O/check_device.log; cat $O/check_device.log
for t in 0 16 32 64; do
  echo "== device C=256 threads $t (0 = default)" | tee -a $O/sweep.log
  GTN_AMD_THREADS=$t BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 100 device 2>&1 | tee -a $O/sweep.log
done
echo "== device C=256 no mallopt" | tee -a $O/sweep.log
GTNX_NO_MALLOPT=1 BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 100 device 2>&1 | tee -a $O/sweep.log
echo "== host alphabet 28" | tee -a $O/sweep.log
BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 28 100 host 2>&1 | tee -a $O/sweep.log
echo "== host alphabet 256" | tee -a $O/sweep.log
BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 30 host 2>&1 | tee -a $O/sweep.log
echo "== timing table" | tee -a $O/sweep.log
GTNX_HOST_TIMING=1 BM_PHASES=1 timeout 300 $B/bm_ctc_c256 512 256 100 device 2>&1 | tee -a $O/sweep.log
timeout 600 $B/bm_ctc 512 2>&1 | tee -a $O/sweep.log
python - <<'PY' 2>&1 | tee -a gpurun_out/r3c/sweep.log
import ctypes, os
lib = ctypes.CDLL("gtn_amd/lib/libgtn_amd.so")
r = ctypes.c_uint64(); u = ctypes.c_uint64()
lib.gtnx_memory_stats(ctypes.byref(r), ctypes.byref(u)); print("mem", r.value, u.value)
PY
