set -u
O=$PWD/gpurun_out/r4a; mkdir -p $O
BM=tests/dropin/_bin/bm_ctc_c256
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
for i in 1 2 3; do BM_PHASES=1 $BM 512 256 50 device >> $O/bm.log 2>&1; done
BM_PHASES=1 $BM 512 256 50 device check >> $O/bm_check.log 2>&1
BM_PHASES=1 GTNX_HOST_TIMING=1 $BM 512 256 50 device > $O/bm_timing.log 2>&1
GTN_AMD_SPIN_US=0 BM_PHASES=1 $BM 512 256 50 device > $O/bm_nospin.log 2>&1
GTNX_REGION_NO_SLICE_PATH=1 BM_PHASES=1 $BM 512 256 50 device > $O/bm_noslice.log 2>&1
cat $O/pytest.log $O/bm.log $O/bm_check.log
