"""The HOST side of the engine without a GPU: tools/nullhip's do-nothing HIP runtime LD_PRELOADed under small
drivers of BASELINE configs C1 / C2 (tools/nullhip/small_step.cpp: the per-graph functions, the vector overloads and
gtn::Batch over 256 linear chains).  Kernels do not run, so nothing here checks a RESULT -- what it pins is that the
host path (graph handles, result graphs out of per-op buffers, deferred teardown counted in graphs: DESIGN.md section
12.3) runs to the end, recycles its buffers, and stops faulting in fresh memory once warm.  Diagnostic tooling only:
nothing in the package, smoke() or bench.py loads the null runtime."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NULLHIP = os.path.join(ROOT, "tools", "nullhip")


def _build():
    if not os.path.exists(os.path.join(ROOT, "gtn_amd", "lib", "libgtn_amd.so")):
        pytest.skip("libgtn_amd.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no ROCm headers")
    r = subprocess.run(["make", "-s", "-C", NULLHIP], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        pytest.skip("tools/nullhip does not build here: " + (r.stdout + r.stderr)[-300:])


@pytest.mark.parametrize("mode", ["c1", "c2", "c2b"])
def test_host_path_runs_and_recycles_its_memory(mode):
    _build()
    env = dict(os.environ, LD_PRELOAD=os.path.join(NULLHIP, "_bin", "libnullhip.so"), GTNX_SLAB_STATS="1", NULLHIP_ZERO="1")
    r = subprocess.run([os.path.join(NULLHIP, "_bin", "small_step"), mode, "400"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    m = re.search(r"minor page faults per repetition: ([0-9.]+)", out)
    assert m, out[-1000:]
    # warm loop: the graphs' memory comes round again instead of being mapped afresh (155 faults per C2 batch before)
    assert float(m.group(1)) <= 40.0, out[-1000:]
    if mode == "c2":  # 256 result graphs per repetition out of one cached buffer
        s = re.search(r"graph slabs: (\d+) from the cache, (\d+) fresh", out)
        assert s and int(s.group(1)) >= 300 and int(s.group(1)) >= 8 * int(s.group(2)), out[-1000:]


def test_garbage_of_a_thread_that_never_synchronises_is_reachable():
    """handles destroyed on one thread go home to the list of the thread that made them; a maker that
    never comes to a reclamation point used to pin every device block behind them.  tools/nullhip/inbox_step.cpp:
    20 000 graphs made by a sleeping worker and destroyed by the main thread -- the worker's list takes at most 16 384
    (the sender destroys the rest), and gtnx_empty_cache() takes every thread's list apart: in-use bytes drop to 0."""
    _build()
    env = dict(os.environ, LD_PRELOAD=os.path.join(NULLHIP, "_bin", "libnullhip.so"))
    r = subprocess.run([os.path.join(NULLHIP, "_bin", "inbox_step"), "20000"], capture_output=True, text=True, timeout=600, env=env,
                       cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "INBOX_OK" in out, out[-2000:]
    m = re.search(r"(\d+) after destroying the handles on another thread", out)
    assert m and int(m.group(1)) == 16384 * 512, out  # bounded: the rest was destroyed by the sender


def test_default_device_is_the_threads_and_is_never_assumed():
    """with no gtnx_set_device anywhere the engine works on the device the calling thread already has
    with HIP (eight fake devices, the thread on 5 -- a rank after torch.cuda.set_device(5)), leaves the thread there,
    and keeps launching on its own device after somebody else's hipSetDevice on the same thread
    (tools/nullhip/device_step.cpp: per-device launch counters, no launch on a stream whose device is not current)."""
    _build()
    env = dict(os.environ, LD_PRELOAD=os.path.join(NULLHIP, "_bin", "libnullhip.so"), NULLHIP_DEVICES="8", NULLHIP_ZERO="1")
    r = subprocess.run([os.path.join(NULLHIP, "_bin", "device_step")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DEVICE_OK" in out, out[-2000:]


def _device_ops(cmd, extra_env=None):
    """the operations the engine puts on its stream during `cmd`, as tools/nullhip's NULLHIP_TRACE prints them"""
    _build()
    env = dict(os.environ, LD_PRELOAD=os.path.join(NULLHIP, "_bin", "libnullhip.so"), NULLHIP_TRACE="1")
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    return [ln[len("nullhip: "):] for ln in r.stderr.splitlines() if ln.startswith("nullhip: ")]


def _last_period(ops, end):
    """the operations between the last two `end` lines (one warm repetition)"""
    idx = [i for i, o in enumerate(ops) if o.startswith(end)]
    assert len(idx) >= 2, ops[-30:]
    return ops[idx[-2] + 1: idx[-1]]


def test_one_utterance_puts_eight_operations_on_the_stream_and_none_of_the_runtimes():
    """BASELINE C1: sixteen dependent device operations per loss at the start of round 5, eight
    of them copies of the HIP runtime.  Now: setWeights and the band records by the engine's copy kernel, the two
    sweeps with their record as the kernel's argument, subtract, seed + gradient function in one launch, one fill --
    and item() reads the pinned mirror: nothing but the synchronisation follows the backward sweep."""
    ops = _last_period(_device_ops([os.path.join(NULLHIP, "_bin", "small_step"), "c1", "60"]), "streamSynchronize")
    assert not [o for o in ops if o.startswith("memcpy") or o.startswith("memset")], ops
    launches = [o for o in ops if o.startswith("launch")]
    assert len(launches) == 7 and len([o for o in ops if not o.startswith("eventRecord")]) == 7, ops
    names = " | ".join(launches)
    for k in ("band_forward_one_kernel", "band_backward_one_kernel", "scalar_combine_kernel", "scalar_fan_kernel"):
        assert k in names, names
    assert names.count("copy_small_kernel") == 2 and names.rstrip().split(" | ")[-1].find("band_backward_one_kernel") >= 0, names


@pytest.mark.parametrize("prog", ["host_step", "region_step"])
def test_a_training_step_has_no_copy_of_the_runtimes_between_its_kernels(prog):
    """the tables of the headline step (labels, target arguments, the sweeps' pair tables) and
    of the reference's loop reach the device by the engine's copy kernel; a hipMemcpyAsync between two kernels cost the
    step 9 % (0.688 -> 0.630 ms).  GTNX_H2D_KERNEL_BYTES=0 brings the runtime's copies back (the switch works)."""
    exe = os.path.join(NULLHIP, "_bin", prog)
    end = "streamSynchronize" if prog == "host_step" else "eventSynchronize"
    ops = _last_period(_device_ops([exe, "6", "64", "100", "32", "10"] if prog == "host_step" else [exe, "6", "64", "32"]), end)
    assert [o for o in ops if "band_backward_kernel" in o], ops
    assert not [o for o in ops if o.startswith("memcpy") or o.startswith("memset")], ops
    assert len([o for o in ops if o.startswith("eventRecord")]) <= 3, ops
    old = _last_period(_device_ops([exe, "6", "64", "100", "32", "10"] if prog == "host_step" else [exe, "6", "64", "32"],
                                   {"GTNX_H2D_KERNEL_BYTES": "0"}), end)
    # (labels + target arguments, the forward sweep's pair table; the backward sweep of a batch record reads the forward
    #  sweep's table again -- kernels.h BandPatch -- unless GTNX_NO_BAND_PATCH brings its own upload back)
    want = 2 if prog == "host_step" else 3
    assert len([o for o in old if o.startswith("memcpyAsync kind 1")]) >= want, old
    if prog == "host_step":
        both = _last_period(_device_ops([exe, "6", "64", "100", "32", "10"], {"GTNX_H2D_KERNEL_BYTES": "0", "GTNX_NO_BAND_PATCH": "1"}), end)
        assert len([o for o in both if o.startswith("memcpyAsync kind 1")]) == len([o for o in old if o.startswith("memcpyAsync kind 1")]) + 1, both
