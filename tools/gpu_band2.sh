#!/bin/bash
set -u
O=gpurun_out/band; mkdir -p $O
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -DGTNX_BAND_TIMING -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench_tm 2>&1 | tail -3; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench 2>&1 | tail -3 )
FUSE=1 tools/ubench/band_bench_tm 512 1000 256 100 2>&1 | tee $O/timing.txt
for c in 256 255 1024 28; do FUSE=1 tools/ubench/band_bench 512 1000 $c 100 2>&1 | head -1; done
tools/ubench/band_bench 512 1000 256 100 2>&1 | head -1
timeout 900 python -m pytest tests/test_lazy_gpu.py tests/test_batch_gpu.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
