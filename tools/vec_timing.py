"""host phases of the vector-overload CTC step (bench_native: gtn_bench_ctc_step_vector) at C3 -- diagnostic"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gtn_amd as gtn
import graphgen as gg
B, T, Cn, U = 512, 1000, 256, 100
em, tg = gg.ctc_inputs(1234, B, T, Cn, U)
em_dev = torch.from_numpy(em).cuda()
tg = np.ascontiguousarray(tg, dtype=np.int32)
native = C.CDLL(os.path.join(ROOT, "bench_native", "libgtn_bench.so"))
native.gtn_bench_ctc_step_vector.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
loss = torch.empty(B, device="cuda"); grad = torch.empty_like(em_dev)
for i in range(30):
    if i == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = native.gtn_bench_ctc_step_vector(em_dev.data_ptr(), tg.ctypes.data, B, T, Cn, U, loss.data_ptr(), grad.data_ptr())
    assert rc == 0
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
