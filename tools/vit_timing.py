"""host phases of the reference-style decode loop at C3 (bench_native: gtn_bench_viterbi_reference_loop) -- diagnostic"""
import ctypes as C, os, sys
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
native.gtn_bench_viterbi_reference_loop.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
native.gtn_bench_viterbi_reference_loop.restype = C.c_double
for _ in range(2):
    ms = native.gtn_bench_viterbi_reference_loop(em_dev.data_ptr(), tg.ctypes.data, B, T, Cn, U, 30, None)
print("ms per batch", ms)
