#!/bin/bash
set -u
O=$PWD/gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "wide_chain" 2>&1 | tail -8
bash tools/gpu_prof_ngram.sh
timeout 300 python tools/bench_c4.py > $O/c4.json 2> $O/c4.err; tail -c 1500 $O/c4.json
