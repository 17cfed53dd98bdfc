#!/bin/bash
set -u
O=gpurun_out/r3g; mkdir -p $O
tools/ubench/host_task_bench 2>&1 | tee $O/task.log
for t in 4 8 16 32 64; do GTN_AMD_THREADS=$t tools/ubench/host_task_bench 2>&1 | grep empty | tee -a $O/task.log; done
