"""ONE process driving the 8 GPUs of a node through the C ABI (north_star: "a batch of independent utterance graphs
shards embarrassingly across the 8 GPUs of one node, one HIP stream per shard, RCCL only to gather scalar losses /
grads"), checked without hardware: tests/native/multidev_test.cpp -- BASELINE config C5's batch of 4096 utterances cut
into 8 blocks of 512 by gtn::parallelMapSharded -- runs against tools/nullhip with NULLHIP_DEVICES=8 (a stand-in HIP
runtime with eight fake devices: per-device streams, allocations and launch counters; kernels do not execute).  It
asserts that every device makes exactly the launches a single device makes for its 512 utterances, on its own
stream and with its own memory pool, that no launch is issued while another device is current, that results stay
on their block's device, and that the loss gather / shared-gradient sum over the RCCL entry points are right.
(The same collectives run on the real librccl at world size 1 in tests/test_distributed_gpu.py.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_fake_devices_shard_c5s_batch():
    exe = os.path.join(ROOT, "tests", "dropin", "_bin", "multidev_test")
    if not os.path.exists(exe):
        pytest.skip("tests/dropin/_bin not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools", "nullhip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    pre = os.path.join(ROOT, "tools", "nullhip", "_bin", "libnullhip.so")
    env = dict(os.environ, LD_PRELOAD=pre, NULLHIP_DEVICES="8", NULLHIP_ZERO="1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "All tests passed" in r.stdout
