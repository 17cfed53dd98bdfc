#!/bin/bash
set -u
O=gpurun_out/r3n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -a -v "^$" $O/pytest.log | tail -30 | cut -c 1-400
