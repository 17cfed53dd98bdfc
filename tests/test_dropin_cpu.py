"""The host-only part of the drop-in proof, runnable without a GPU: the reference's
graph_test / creations_test / utils_test (test/*.cpp, compiled unmodified against
include/gtn by tests/dropin/Makefile) only build, inspect, copy, sort, print, save and
load graphs -- no device op -- so they must pass on the CPU-only host as well."""
import os
import subprocess

import pytest

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin", "_bin")


@pytest.mark.parametrize("name", ["graph_test", "creations_test", "utils_test"])
def test_reference_host_test_program(name):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip("tests/dropin/_bin not built (needs /root/reference: __graft_entry__.build())")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "All tests passed" in r.stdout
