"""world_size-2 gloo test of the multi-GPU plumbing (sharding + loss gather +
max-over-ranks timing) on CPU.  The per-utterance compute is stood in by the C
oracle (tests may call it); on GPUs the same plumbing wraps the HIP engine."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, T, C, U, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import graphgen as gg
    from oracle_lib import ctc_loss
    from gtn_amd.distributed import gather_losses, max_over_ranks, shard_range
    em, tg = gg.ctc_inputs(77, B, T, C, U)              # same seed on every rank
    lo, hi = shard_range(B, rank, world)
    local = torch.tensor([ctc_loss(em[b], tg[b])[0] for b in range(lo, hi)], dtype=torch.float32)
    allv = gather_losses(local, B)
    t = max_over_ranks(0.5 + rank)
    if rank == 0:
        ret["losses"] = allv.numpy().copy()
        ret["t"] = t
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 7])
def test_sharded_losses_match_single_process(B):
    from gtn_amd.distributed import shard_range
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import graphgen as gg
    from oracle_lib import ctc_loss
    T, C, U, world = 30, 8, 4, 2
    # shards tile the batch
    cover = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        cover += list(range(lo, hi))
    assert cover == list(range(B))
    ret = mp.Manager().dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, B, T, C, U, ret), nprocs=world, join=True)
    em, tg = gg.ctc_inputs(77, B, T, C, U)
    want = np.array([ctc_loss(em[b], tg[b])[0] for b in range(B)], np.float32)
    np.testing.assert_array_equal(ret["losses"], want)
    assert ret["t"] == 1.5        # slowest rank


def _asg_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gtn_amd.distributed import all_reduce_shared_grad, shard_range
    from test_oracle import asg_utterance_grads
    from test_parity_gpu import ASG_EMISSIONS
    targets = [[2, 1, 5, 1, 3], [4, 3, 5], [3, 2, 2, 1]]
    lo, hi = shard_range(3, rank, world)
    part = np.zeros(6 + 36, np.float64)
    for b in range(lo, hi):
        part += asg_utterance_grads(5, 6, ASG_EMISSIONS[b], targets[b])[2]
    g = all_reduce_shared_grad(torch.from_numpy(part))
    if rank == 0:
        ret["grad"] = g.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def test_asg_shared_transition_grad_all_reduce():
    """the one real exchange of the ASG variant.  Two ranks own 2 + 1 utterances of the
    reference's ASG test; the all-reduced transition gradient is the batch sum it pins
    (criterion_test.cpp:289-305)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_parity_gpu import ASG_TRANS_GRAD
    ret = mp.Manager().dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_asg_worker, args=(2, port, ret), nprocs=2, join=True)
    np.testing.assert_allclose(ret["grad"][6:], ASG_TRANS_GRAD, atol=1e-4)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_branch_under_gloo(world, tmp_path):
    """bench.py's OWN world > 1 branch -- rendezvous from the torchrun environment, all_gather of the
    per-rank losses, barrier-bracketed timing, max over ranks, the per-rank report -- driven as the round
    driver launches it (python -m torch.distributed.run ... bench.py --gpus N), with --dry-run-cpu standing
    in for the step (gloo instead of RCCL, no kernel)."""
    import json
    import subprocess
    port = 29600 + (os.getpid() % 300)
    port += world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4",
           "--warmup", "1", "--batch", "8", "--dry-run-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, GTN_BENCH_OUT=str(tmp_path)))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout      # rank 0 prints ONE line
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]   # ... and it is the LAST line of stdout
    line = json.loads(lines[0])
    # the line the driver parses is short (BENCH_r05.json: a 20 KB line was not parsed); the rest is in the side file
    assert len(lines[0]) < 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "config", "full_record"):
        assert k in line, k
    import hashlib
    raw = open(os.path.join(str(tmp_path), "last_full.json"), "rb").read()  # (the digest is the FILE's: sha256sum agrees)
    assert hashlib.sha256(raw).hexdigest() == line["full_record"]["sha256"] and len(raw) == line["full_record"]["bytes"]
    text = raw.decode()
    assert line["rank_seconds"] and len(line["rank_seconds"]) == world
    out = json.loads(text)
    assert out["n_gpus"] == world and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["dry_run"] is True and out["gather_ok"] is True
    assert [p["rank"] for p in out["per_rank"]] == list(range(world))   # the 8-GPU line: eight per_rank entries
    assert out["config"]["rccl_ranks"] == world and out["config"]["global_batch"] == world * 8
    assert all(p["host_threads"] >= 1 for p in out["per_rank"])
    # max over ranks: the reported step time is no shorter than any rank's own
    assert out["ms_per_step"] * 4 >= 1e3 * max(p["seconds"] for p in out["per_rank"]) - 1e-6


def test_bench_line_is_compact():
    """bench.py's final line, made from a real full record of round 5 (20 KB: the line the driver could not parse):
    under 4 KB, parses, and carries the contract's keys, `roofline`, `cpu_baseline` and the parity verdict"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_v9_c3_bench.json")))
    full["ms_per_step_cold"], full["value_cold"] = 0.668, 766000.0
    text = bench.compact_line(full, {"path": "bench_out/last_full.json", "sha256": "0" * 64, "bytes": 20500})
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "priming_steps", "ms_per_step", "ms_per_step_cold",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
              "parity_in_run", "full_record"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BASELINE config C3") and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["parity_in_run"]["ok"] is True
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    # a record with absurdly long optional parts still fits (they are shed, the contract's keys stay)
    full["roofline_other"] = {("k%d" % i): {"kernel": "x" * 100, "frac": 0.5, "ms_per_launch": 1.0, "traffic": 1.0} for i in range(40)}
    text = bench.compact_line(full, None)
    assert len(text) < bench.LINE_LIMIT and "roofline" in json.loads(text)
