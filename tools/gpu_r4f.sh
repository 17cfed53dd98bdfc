set -u
O=$PWD/gpurun_out/r4f; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest.log
run() { echo "== $*" >> $O/trace.log; env "$@" GTN_AMD_POOL_TRACE=1 BM_PHASES=1 $BM 512 256 100 device >> $O/trace.log 2>&1; }
run A=1
run A=2
run A=3
run GTNX_NO_ITEM_PREFETCH=1
run GTN_AMD_THREADS=64
run GTN_AMD_THREADS=16
GTNX_HOST_TIMING=1 BM_PHASES=1 $BM 512 256 100 device > $O/bm_timing.log 2>&1
$BM 512 256 20 device check > $O/check.log 2>&1
cat $O/pytest.log; cat $O/trace.log; grep -E "gtnx" $O/bm_timing.log; cat $O/check.log
