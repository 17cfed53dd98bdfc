set -u
O=$PWD/gpurun_out/r4e; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
run() { echo "== $*" >> $O/trace.log; env "$@" GTN_AMD_POOL_TRACE=1 BM_PHASES=1 $BM 512 256 50 device >> $O/trace.log 2>&1; }
run GTN_AMD_SPIN_US=0
run GTN_AMD_SPIN_US=0 GTN_AMD_NO_RECLAIM=1
run GTN_AMD_SPIN_US=1000 GTN_AMD_NO_RECLAIM=1
run GTN_AMD_SPIN_US=0 GTNX_RECLAIMERS=8
run GTN_AMD_SPIN_US=0 GTNX_NO_MALLOPT=1
run GTN_AMD_SPIN_US=0 MALLOC_ARENA_MAX=4
grep -vE "^\{" $O/trace.log
