#!/bin/bash
set -u
bash tools/profile_gpu.sh r03v1 > gpurun_out/prof_r03v1.log 2>&1; tail -30 gpurun_out/prof_r03v1.log
( time timeout 1500 python bench.py ) > gpurun_out/bench_r03v1.log 2>&1; tail -3 gpurun_out/bench_r03v1.log | cut -c 1-1500
timeout 300 python tools/c4_trace.py > gpurun_out/c4_trace_r03v1.txt 2>&1; tail -25 gpurun_out/c4_trace_r03v1.txt
