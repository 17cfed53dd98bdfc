"""The RCCL calls of the multi-GPU path, EXECUTED -- on the one GPU a test box has, as a
process group of size one: torch.distributed's "nccl" backend is RCCL on ROCm.  world_size-2 behaviour (sharding,
ordering, the ASG shared-gradient sum) is covered over gloo in tests/test_distributed_cpu.py; what these tests
add is that init_process_group("nccl"), all_gather, all_reduce and barrier run on device tensors produced by the
engine, and that bench.py's own multi-rank branch runs end to end over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = %(port)r
os.environ["GTN_AMD_FORCE_COLLECTIVES"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import gtn_amd as gtn, graphgen as gg
from oracle_lib import ctc_loss
from gtn_amd.distributed import gather_losses, all_reduce_shared_grad, max_over_ranks
B, T, C, U = 6, 40, 10, 5
em, tg = gg.ctc_inputs(5, B, T, C, U)
em_dev = torch.from_numpy(em).cuda()
ems = gtn.linear_graph_n(B, T, C, em_dev)
ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
loss = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(gtn.intersect(ctcs, ems)))
gtn.backward(loss)
local = torch.empty(B, dtype=torch.float32, device="cuda")
gtn.items_to_device(loss, local)
gtn.synchronize()
allv = gather_losses(local, B)                      # all_gather over RCCL
want = np.array([ctc_loss(em[b], tg[b])[0] for b in range(B)], np.float32)
np.testing.assert_allclose(allv.cpu().numpy(), want, rtol=1e-4)
g = torch.arange(12, dtype=torch.float32, device="cuda")
r = all_reduce_shared_grad(g.clone())               # all_reduce(SUM) over RCCL: one rank -> unchanged
assert torch.equal(r, g)
assert max_over_ranks(0.25, torch.device("cuda", 0)) == 0.25   # all_reduce(MAX)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
"""


@pytest.mark.gpu
def test_rccl_collectives_run_on_engine_tensors():
    port = str(29500 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT, "port": port}], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "RCCL_OK True" in r.stdout


@pytest.mark.gpu
def test_bench_multi_rank_branch_over_rccl():
    """bench.py's world > 1 code (process group, all_gather of the losses every step, barrier-bracketed timing,
    max over ranks, per-rank report) forced through the nccl branch with WORLD_SIZE = 1"""
    import tempfile
    env = dict(os.environ, GTN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + os.getpid() % 2000),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", GTN_BENCH_OUT=tempfile.mkdtemp(prefix="gtn_bench_"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--batch", "64", "--no-cpu-baseline", "--no-unmodified-caller", "--no-configs", "--no-reference-api",
                        "--no-built-lattice"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    # the compact line is the LAST line of stdout (bench.py leaves the process group before it prints: RCCL writes to
    # stdout at teardown), the per-rank report is in the full record beside it
    last = r.stdout.rstrip().splitlines()[-1]
    assert last.startswith("{") and len(last) < 4096, r.stdout[-2000:]
    line = json.loads(last)
    assert line["collectives"] and "RCCL" in line["collectives"]
    assert line["value"] > 0 and line["config"]["rccl_ranks"] == 1
    full = json.load(open(os.path.join(env["GTN_BENCH_OUT"], "last_full.json")))
    assert len(full["per_rank"]) == 1 and full["value"] == pytest.approx(line["value"], rel=1e-4)


@pytest.mark.gpu
def test_c_abi_collectives_run_on_rccl_at_world_size_one():
    """gtnx_comm_* (gtn_amd/csrc/comm.cpp: what a C++ host that drives several GPUs from ONE process gathers losses and
    sums the shared ASG gradient with) on the real librccl: a communicator over this box's one GPU, forced through the
    library (GTNX_COMM_FORCE_RCCL=1 -- a one-device communicator would otherwise copy), ncclAllGather / ncclAllReduce
    enqueued on the engine's stream.  The eight-device logic is tests/test_multidevice_cpu.py's."""
    import ctypes as C
    import subprocess
    import sys
    code = r"""
import ctypes as C, os, sys
import numpy as np, torch
import gtn_amd as gtn
from gtn_amd import _capi
lib = _capi.load()
dev = (C.c_int * 1)(0)
comm = C.c_void_p()
assert lib.gtnx_comm_create(dev, 1, C.byref(comm)) == 0, lib.gtnx_last_error()
n = C.c_int()
assert lib.gtnx_comm_size(comm, C.byref(n)) == 0 and n.value == 1
send = torch.arange(512, dtype=torch.float32, device="cuda:0")
recv = torch.full((512,), -1.0, device="cuda:0")
torch.cuda.synchronize()
sp = (C.c_void_p * 1)(send.data_ptr()); rp = (C.c_void_p * 1)(recv.data_ptr())
assert lib.gtnx_comm_all_gather_f32(comm, sp, rp, 512) == 0, lib.gtnx_last_error()
g = torch.full((262656,), 2.5, device="cuda:0")
torch.cuda.synchronize()
gp = (C.c_void_p * 1)(g.data_ptr())
assert lib.gtnx_comm_all_reduce_sum_f32(comm, gp, 262656) == 0, lib.gtnx_last_error()
assert lib.gtnx_synchronize() == 0
assert torch.equal(recv, send), "all_gather over one rank is the identity"
assert float(g.min()) == 2.5 and float(g.max()) == 2.5, "all_reduce(sum) over one rank is the identity"
assert lib.gtnx_comm_destroy(comm) == 0
print("RCCL_OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTNX_COMM_FORCE_RCCL="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
