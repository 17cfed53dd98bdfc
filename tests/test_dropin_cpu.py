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


ALL = ["graph_test", "functions_test", "autograd_test", "creations_test", "criterion_test", "parallel_test",
       "rand_test", "utils_test"]


@pytest.mark.parametrize("name", ALL)
def test_headers_and_abi_over_the_unmodified_reference(name, tmp_path):
    """The boundary itself, without the engine: the same eight binaries (reference test sources compiled
    against include/gtn) with oracle/_ref/libgtn_ref.so -- the UNMODIFIED reference behind the C ABI of
    include/gtn_amd.h -- substituted for libgtn_amd.so.  Every assertion must pass: the header-only mirror
    and the ABI's semantics (aliasing, exceptions, grad functions, formats) are the reference's.
    (tests/test_dropin_gpu.py runs the same binaries on the HIP engine.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(BIN, name)
    ref = os.path.join(root, "oracle", "_ref", "libgtn_ref.so")
    if not os.path.exists(exe) or not os.path.exists(ref):
        pytest.skip("needs tests/dropin/_bin and oracle/_ref (built from /root/reference by __graft_entry__.build())")
    os.symlink(ref, tmp_path / "libgtn_amd.so")  # the binaries ask for this name; LD_LIBRARY_PATH precedes RUNPATH
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "All tests passed" in r.stdout
