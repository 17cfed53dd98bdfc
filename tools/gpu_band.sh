#!/bin/bash
set -u
O=gpurun_out/band; mkdir -p $O
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench 2>&1 | tail -3 )
for c in 256 255 254 29 28; do
  FUSE=1 tools/ubench/band_bench 512 1000 $c 100 2>&1 | head -1 | tee -a $O/log2.txt
done
timeout 900 python -m pytest tests/test_lazy_gpu.py tests/test_batch_gpu.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
