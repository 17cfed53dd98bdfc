#!/bin/bash
set -u
cd tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench_x 2>&1 | tail -3
for c in 256 255 28 512 1024; do
  FUSE=1 timeout 120 ./band_bench_x 512 1000 $c 100 2>&1 | head -1
done
FUSE=1 timeout 120 ./band_bench_x 512 1000 256 200 2>&1 | head -1
FUSE=1 timeout 120 ./band_bench_x 512 2000 1024 200 2>&1 | head -1
