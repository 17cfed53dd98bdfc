import os, sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import gtn_amd as gtn, graphgen as gg
native = C.CDLL('/root/repo/bench_native/libgtn_bench.so')
B,T,Cn,U = 512,1000,256,100
em, tg = gg.ctc_inputs(1234, B, T, Cn, U)
em_dev = torch.from_numpy(em).cuda(); tg = np.ascontiguousarray(tg, np.int32)
loss = torch.empty(B, dtype=torch.float32, device='cuda'); grad = torch.empty_like(em_dev)
native.gtn_bench_ctc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for s in range(1, 301):
    assert native.gtn_bench_ctc_step(em_dev.data_ptr(), tg.ctypes.data, B, T, Cn, U, loss.data_ptr(), grad.data_ptr()) == 0
    if s in (5, 20, 50, 100, 200, 300):
        gtn.synchronize(); m = gtn.memory_stats(); print(s, 'reserved GB', m['reserved']/1e9, 'in use GB', m['in_use']/1e9, flush=True)
