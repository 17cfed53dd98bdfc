#!/bin/bash
# what a backward tick costs without the helpers' jobs (tools/ubench/band_bench, -DGTNX_EXP_NO_DRAIN / -DGTNX_EXP_NO_STAGE)
set -u
cd tools/ubench
for v in "" "-DGTNX_EXP_NO_DRAIN" "-DGTNX_EXP_NO_STAGE" "-DGTNX_EXP_NO_DRAIN -DGTNX_EXP_NO_STAGE"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $v -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench_x 2>&1 | tail -3
  echo "== [$v]"
  FUSE=1 timeout 120 ./band_bench_x 512 1000 256 100 2>&1 | head -1
done
