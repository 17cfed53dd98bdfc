set -u
REPO=$(pwd); O=$REPO/gpurun_out/cprof; mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
for mode in 0 1; do
  GTNX_GRID_REPLICATION=$mode timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$mode -- python $REPO/tools/compose_stats.py > $O/log$mode.txt 2>&1
  s=$(find $O/t$mode -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $O/stats_grid$mode.csv
  t=$(find $O/t$mode -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && grep -E "compose_kernel|replicate" $t | awk -F, '{print $(NF-8), $0}' | cut -c1-40 > /dev/null
  rm -rf $O/t$mode
  echo "== grid=$mode"; grep -E "compose_kernel|replicate" $O/stats_grid$mode.csv | cut -c1-230
done
