"""Drop-in proof: the reference's own Catch2 test programs (test/*.cpp), compiled
UNMODIFIED against include/gtn and linked to libgtn_amd.so by tests/dropin/Makefile
(run from __graft_entry__.build() where /root/reference exists), executed here on
the GPU.  Nothing is read from /root/reference at run time: the binaries are
prebuilt and travel with the snapshot."""
import os
import subprocess

import pytest

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin", "_bin")
PROGRAMS = [
    "graph_test",
    "functions_test",
    "autograd_test",
    "creations_test",
    "criterion_test",
    "parallel_test",
    "rand_test",
    "utils_test",
    "gather_test",  # own program (tests/native/gather_test.cpp): the calls of parallelMap threads vs one at a time
    "region_test",  # own program (tests/native/region_test.cpp): deferred calls behave like immediate ones
    "adjacency_refs_test",  # own program: references into out(n) / in(n) / start() / accept() stay valid across calls
]


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROGRAMS)
def test_reference_test_program(name):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build() where /root/reference is available")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, f"{name} failed:\n{tail}"
    assert "All tests passed" in r.stdout, tail


EXPECTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples_expected.json")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ex_ctc", "ex_asg"])
def test_reference_example_program(name):
    """the reference's CTC / ASG criteria (examples/ctc.cpp, examples/asg.cpp), unmodified, on the engine:
    same output as built against the reference itself (tests/golden/make_examples_expected.py)"""
    import json
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build() where /root/reference is available")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout == json.load(open(EXPECTED))[name]


@pytest.mark.gpu
def test_reference_ctc_benchmark_runs():
    """benchmarks/ctc.cpp, unmodified (per-utterance calls from parallelMap threads, gathered by the
    engine into batched launches): all five timings come out"""
    import re
    exe = os.path.join(BIN, "bm_ctc")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build() where /root/reference is available")
    r = subprocess.run([exe, "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    times = dict(re.findall(r"Timing (\w+) \.\.\.\s+([0-9.e+-]+) msec", r.stdout))
    assert set(times) == {"ctcLoss", "ctcGrad", "ngramCtcLoss", "ngramCtcGrad", "ctcBatched"}, r.stdout
    # (time_utils.h:26-44 counts whole milliseconds over 100 calls: a lambda that only ENQUEUES -- ctcLoss reads no
    # result -- may print 0 when a call costs the host less than 10 us)
    assert all(float(v) >= 0 for v in times.values()), r.stdout
    assert float(times["ctcBatched"]) > 0 and float(times["ctcGrad"]) + float(times["ctcLoss"]) > 0, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("alphabet", [256, 255, 30])  # (255, 30: the scalar staging path of the band sweeps)
def test_reference_loop_at_c3_matches_one_call_at_a_time(alphabet):
    """tests/native/bm_ctc_c256.cpp in `check` mode at BASELINE config C3's shape (64 utterances): the losses and
    emission gradients of parallelMap(fwd) / parallelMap(bwd) against the same functions called one utterance at a time
    (symbolic lattices: equal to 1e-5; built lattices: the float32 lattice recursion's own rounding), with the
    caller's emission buffer WIPED between the two maps -- setWeights copies (graph.cpp:179-181), and the engine
    makes that copy inside the region's forward sweep"""
    import json
    exe = os.path.join(BIN, "bm_ctc_c256")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build() where /root/reference is available")
    r = subprocess.run([exe, "64", str(alphabet), "3", "device", "check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    checks = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{"check"')]
    assert len(checks) == 2 and all(c["worst_rel_loss"] <= 1e-5 for c in checks), r.stdout
