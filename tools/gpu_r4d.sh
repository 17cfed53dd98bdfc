set -u
O=$PWD/gpurun_out/r4d; mkdir -p $O; rm -f $O/*
BM=tests/dropin/_bin/bm_ctc_c256
for s in 0 1000; do for t in 0 16; do echo "== spin $s threads $t" >> $O/trace.log; GTN_AMD_THREADS=$t GTN_AMD_SPIN_US=$s GTN_AMD_POOL_TRACE=1 BM_PHASES=1 $BM 512 256 50 device >> $O/trace.log 2>&1; done; done
cat $O/trace.log
