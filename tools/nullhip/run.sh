#!/bin/sh
# usage: tools/nullhip/run.sh [steps B T C U]   (env: GTNX_HOST_TIMING=1 GTN_BENCH_TIMING=1 GTNX_LAZY_COMPOSE=2 ...)
cd "$(dirname "$0")/../.." && make -s -C tools/nullhip && \
  LD_PRELOAD=tools/nullhip/_bin/libnullhip.so exec tools/nullhip/_bin/host_step "$@"
