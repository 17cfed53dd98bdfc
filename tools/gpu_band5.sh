#!/bin/bash
set -u
O=$PWD/gpurun_out/band5; mkdir -p $O
timeout 1500 python -m pytest tests/test_lazy_gpu.py tests/test_batch_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 200 python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller --no-built-lattice 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
done
timeout 200 python bench.py --config c5 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-reference-api --no-unmodified-caller --no-built-lattice 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
