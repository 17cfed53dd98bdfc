#!/bin/bash
set -u
O=gpurun_out/r3i; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time timeout 1500 python bench.py --steps 50 ) > $O/bench.log 2>&1; tail -4 $O/bench.log | cut -c 1-6000
