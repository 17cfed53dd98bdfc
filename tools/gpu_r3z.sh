#!/bin/bash
set -u
O=$PWD/gpurun_out/r3z; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/gtn_amd/lib:/opt/rocm/lib:${LD_LIBRARY_PATH:-}
HOST_STEP_DEVICE=1 HOST_STEP_NO_SYNC=1 HOST_STEP_FN=gtn_bench_ctc_step_vector GTN_HOST_SAMPLE=$O/s.txt timeout 120 tools/nullhip/_bin/host_step 300 2>&1 | tail -3
python tools/nullhip/report.py $O/s.txt 60 > $O/report.txt 2>&1; head -75 $O/report.txt
