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
