set -u
O=$PWD/gpurun_out/r4c; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest.log
for i in 1 2 3; do BM_PHASES=1 $BM 512 256 50 device >> $O/bm.log 2>&1; done
BM_PHASES=1 GTNX_HOST_TIMING=1 $BM 512 256 50 device > $O/bm_timing.log 2>&1
for s in 0 200 3000; do echo "spin $s" >> $O/bm_spin.log; GTN_AMD_SPIN_US=$s BM_PHASES=1 $BM 512 256 50 device >> $O/bm_spin.log 2>&1; done
for t in 8 16 64; do echo "threads $t" >> $O/bm_thr.log; GTN_AMD_THREADS=$t BM_PHASES=1 $BM 512 256 50 device >> $O/bm_thr.log 2>&1; done
cat $O/pytest.log; grep "host ms" $O/bm.log; grep -E "gtnx|host ms" $O/bm_timing.log; grep -E "spin|host ms" $O/bm_spin.log; grep -E "threads|host ms" $O/bm_thr.log
