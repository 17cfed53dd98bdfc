#!/bin/bash
set -u
O=gpurun_out/r3j; mkdir -p $O
GTN_BENCH_TIMING=1 GTNX_HOST_TIMING=1 python - > $O/vec.log 2>&1 <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gtn_amd as gtn, graphgen as gg
B,T,Cn,U = 512,1000,256,100
em, tg = gg.ctc_inputs(1234, B, T, Cn, U)
em_dev = torch.from_numpy(em).cuda(); loss = torch.empty(B, device="cuda"); grad = torch.empty(B,T,Cn, device="cuda")
n = C.CDLL("bench_native/libgtn_bench.so")
n.gtn_bench_ctc_step_vector.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int]*4 + [C.c_void_p, C.c_void_p]
for i in range(3): n.gtn_bench_ctc_step_vector(em_dev.data_ptr(), tg.ctypes.data, B,T,Cn,U, loss.data_ptr(), grad.data_ptr())
torch.cuda.synchronize(); gtn.synchronize()
t0=time.perf_counter()
for i in range(20): n.gtn_bench_ctc_step_vector(em_dev.data_ptr(), tg.ctypes.data, B,T,Cn,U, loss.data_ptr(), grad.data_ptr())
gtn.synchronize(); torch.cuda.synchronize()
print("vector step ms", (time.perf_counter()-t0)/20*1e3)
PY
cat $O/vec.log | tail -40
