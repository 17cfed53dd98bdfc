#!/bin/bash
set -u
O=gpurun_out/r3m; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_lazy_gpu.py -m gpu -q -k "golden_compose or pinned_to_the_reference or real_alphabet" > $O/pytest.log 2>&1; grep -v "^$" $O/pytest.log | tail -60 | cut -c 1-600
