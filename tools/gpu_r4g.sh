set -u
O=$PWD/gpurun_out/r4g; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
run() { echo "== $*" >> $O/trace.log; env "$@" GTN_AMD_POOL_TRACE=1 BM_PHASES=1 $BM 512 256 100 device >> $O/trace.log 2>&1; }
run A=1
run A=2
run A=3
run GTN_AMD_THREADS=16
run GTN_AMD_THREADS=48
GTNX_HOST_TIMING=1 BM_PHASES=1 $BM 512 256 100 device > $O/bm_timing.log 2>&1
$BM 512 256 20 device check > $O/check.log 2>&1
cat $O/trace.log; grep -E "gtnx" $O/bm_timing.log; cat $O/check.log
