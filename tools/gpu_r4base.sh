set -u
O=$PWD/gpurun_out/r4base; mkdir -p $O
BM=tests/dropin/_bin/bm_ctc_c256
nproc > $O/nproc.txt
for i in 1 2 3; do BM_PHASES=1 $BM 512 256 50 device >> $O/bm.log 2>&1; done
BM_PHASES=1 GTNX_HOST_TIMING=1 $BM 512 256 50 device > $O/bm_timing.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $OLDPWD/$BM 512 256 10 device > $O/stats.log 2>&1
cd $OLDPWD
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $O/kernel_stats.csv; rm -rf $O/stats
cat $O/bm.log; head -8 $O/kernel_stats.csv
