#!/bin/bash
set -u
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -I ../../gtn_amd/csrc -I ../../include band_bench.hip -o band_bench_x 2>&1 | tail -3
  for c in 256 512 255; do FUSE=1 timeout 120 ./band_bench_x 512 1000 $c 100 2>&1 | head -1; done )
timeout 1500 python -m pytest tests/test_lazy_gpu.py tests/test_batch_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 200 python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller --no-built-lattice 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
