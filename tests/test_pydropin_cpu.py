"""The Python side of the drop-in, on the CPU-only host: the reference's OWN Python binding tests
(bindings/python/test/test_*.py, 47 tests: graph construction, weights, formats, functions incl. the
rational ops, autograd, criteria, parallel forms) run UNMODIFIED with `import gtn` resolving to this
repo's Python mirror (gtn_amd/api.py) bound by tests/refbackend/gtn_ref.py to oracle/_ref/libgtn_ref.so -- the unmodified reference behind the C ABI
of include/gtn_amd.h.  What this pins is the Python mirror (gtn_amd/api.py + gtn_amd/hostops): same names,
overloads, keyword arguments, broadcasting and exception types as the pybind11 binding.  Needs
/root/reference (test sources are read where they lie) -- skipped elsewhere."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/bindings/python/test"


def test_reference_python_binding_tests_run_on_the_mirror(tmp_path):
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libgtn_ref.so")
    host = os.path.join(ROOT, "oracle", "_ref", "libgtn_hostops.so")
    if not (os.path.isdir(REF_TESTS) and os.path.exists(ref_lib) and os.path.exists(host)):
        pytest.skip("needs /root/reference and oracle/_ref (python -c 'import __graft_entry__ as g; g.build()')")
    pkg = tmp_path / "gtn"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(
        "import gtn_ref as _g\n"
        "globals().update({k: getattr(_g, k) for k in dir(_g) if not k.startswith('__')})\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), os.path.join(ROOT, "tests", "refbackend"), ROOT]))
    r = subprocess.run([sys.executable, "-m", "unittest", "discover", "-s", REF_TESTS], capture_output=True, text=True,
                       timeout=600, env=env, cwd=str(tmp_path))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "Ran 47 tests" in tail and "\nOK" in tail, tail


@pytest.mark.parametrize("script,expect", [("linear_crf.py", "Test: Accuracy"), ("pytorch_loss.py", "Grad has shape")])
def test_reference_python_examples_run_on_the_mirror(script, expect, tmp_path):
    """bindings/python/examples: a linear-chain CRF trained for 10 000 steps with gtn ops + autograd, and the
    PyTorch CTC loss module (pytorch_loss.py:19-102) -- unmodified, `import gtn` = gtn_amd over the reference
    backend.  (simple_graph.py and word_decompositions.py only need graphviz's `dot` on top, absent here.)"""
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libgtn_ref.so")
    src = os.path.join("/root/reference/bindings/python/examples", script)
    if not (os.path.exists(src) and os.path.exists(ref_lib)):
        pytest.skip("needs /root/reference and oracle/_ref")
    pkg = tmp_path / "gtn"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(
        "import gtn_ref as _g\n"
        "globals().update({k: getattr(_g, k) for k in dir(_g) if not k.startswith('__')})\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), os.path.join(ROOT, "tests", "refbackend"), ROOT]))
    r = subprocess.run([sys.executable, src], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert expect in r.stdout, r.stdout[-2000:]
    if script == "linear_crf.py":
        acc = float(r.stdout.strip().splitlines()[-1].split()[-1])
        assert acc > 0.9, r.stdout[-500:]


def test_reference_pybind11_binding_links_against_the_engine(tmp_path):
    """the reference's own pybind11 modules (bindings/python/gtn/_*.cpp), compiled unmodified against include/gtn and
    linked to libgtn_amd.so (tests/pydropin/Makefile): the package imports, carries the binding's public names, builds
    a graph on the host side -- and, with no GPU here, a graph FUNCTION fails loudly (no CPU fallback).  The compute
    side is tests/test_pydropin_gpu.py."""
    import glob
    ext = os.path.join(ROOT, "tests", "pydropin", "_ext")
    if not glob.glob(os.path.join(ext, "gtn", "_graph*.so")):
        pytest.skip("tests/pydropin/_ext not built (needs /root/reference: __graft_entry__.build())")
    prog = (
        "import gtn\n"
        "names = ['Graph', 'compose', 'intersect', 'forward_score', 'viterbi_score', 'viterbi_path', 'backward', 'negate',\n"
        "         'add', 'subtract', 'linear_graph', 'scalar_graph', 'parallel_for', 'closure', 'union', 'concat', 'remove',\n"
        "         'project_input', 'project_output', 'clone', 'equal', 'isomorphic', 'load', 'save', 'savetxt', 'loadtxt',\n"
        "         'write_dot', 'rand_equivalent', 'epsilon']\n"
        "missing = [n for n in names if not hasattr(gtn, n)]\n"
        "assert not missing, missing\n"
        "g = gtn.Graph(); g.add_node(True); g.add_node(False, True); g.add_arc(0, 1, 3, 4, 0.5)\n"
        "assert (g.num_nodes(), g.num_arcs(), g.labels_to_list(), g.labels_to_list(False), g.weights_to_list()) == (2, 1, [3], [4], [0.5])\n"
        "import torch\n"
        "if not torch.cuda.is_available():\n"
        "    try:\n"
        "        gtn.forward_score(g)\n"
        "        raise SystemExit('forward_score ran without a GPU')\n"
        "    except RuntimeError as e:\n"
        "        assert 'no CPU fallback' in str(e), e\n"
        "print('PYBIND_OK', gtn.__file__)\n")
    env = dict(os.environ, PYTHONPATH=ext)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "PYBIND_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
