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


@pytest.mark.parametrize("name", ALL + ["adjacency_refs_test", "region_test", "gather_test"])
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


EXAMPLES = ["asg", "ctc"]  # the examples that call the hot path; the others are out of scope


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_examples_build_and_run_over_the_reference(name, tmp_path):
    """examples/*.cpp (and benchmarks/*.cpp, compile-only) build unmodified against include/gtn -- they lean on
    headers the reference pulls in transitively (<iostream>, <cassert>) -- and the examples run to completion,
    their own asserts included, on the reference backend behind the C ABI."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(BIN, "ex_" + name)
    ref = os.path.join(root, "oracle", "_ref", "libgtn_ref.so")
    if not os.path.exists(exe) or not os.path.exists(ref):
        pytest.skip("needs tests/dropin/_bin and oracle/_ref (built from /root/reference by __graft_entry__.build())")
    for b in ("ctc", "functions"):
        assert os.path.exists(os.path.join(BIN, "bm_" + b)), "benchmarks/%s.cpp did not build" % b
    os.symlink(ref, tmp_path / "libgtn_amd.so")
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
