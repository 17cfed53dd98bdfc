"""The Python side of the drop-in ON THE ENGINE: the reference's own Python binding tests
(bindings/python/test/test_*.py: graph construction, weights, formats, functions incl. the rational ops,
autograd, criteria, parallel forms) and its PyTorch CTC loss example (examples/pytorch_loss.py:19-102), unmodified,
with `import gtn` resolving to gtn_amd over libgtn_amd.so -- the MI355X engine, not the reference library
(tests/test_pydropin_cpu.py runs the same sources on the reference backend).  The GPU box has no /root/reference:
the tests are run from their bytecode (tests/pydropin/_pyc, compiled where the sources lie by
tests/pydropin/build_pyc.py: the compiled form of the reference's tests travels like tests/dropin/_bin).

Two bindings, the same tests: `ctypes` = gtn_amd/api.py (this repository's Python mirror over the C ABI), and `pybind11` =
the REFERENCE'S OWN binding (bindings/python/gtn/_*.cpp + __init__.py), compiled unmodified against include/gtn and
linked to libgtn_amd.so by tests/pydropin/Makefile -- the caller SURVEY.md section 8(b) names, recompiled, not rewritten."""
import glob
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYC = os.path.join(ROOT, "tests", "pydropin", "_pyc")

RUNNER = r'''
import importlib.machinery, importlib.util, sys, unittest
name, path = sys.argv[1], sys.argv[2]
loader = importlib.machinery.SourcelessFileLoader(name, path)
spec = importlib.util.spec_from_loader(name, loader)
mod = importlib.util.module_from_spec(spec)
sys.modules[name] = mod
loader.exec_module(mod)
suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
res = unittest.TextTestRunner(verbosity=1).run(suite)
print("RAN", res.testsRun, "FAILED", len(res.failures) + len(res.errors))
sys.exit(0 if res.wasSuccessful() else 1)
'''


EXT = os.path.join(ROOT, "tests", "pydropin", "_ext")


def _env(tmp_path, binding="ctypes"):
    if binding == "pybind11":
        if not glob.glob(os.path.join(EXT, "gtn", "_graph*.so")):
            pytest.skip("tests/pydropin/_ext not built (needs /root/reference: __graft_entry__.build())")
        pkgroot = EXT
    else:
        pkg = tmp_path / "gtn"
        pkg.mkdir()
        (pkg / "__init__.py").write_text(
            "import gtn_amd as _g\n"
            "globals().update({k: getattr(_g, k) for k in dir(_g) if not k.startswith('__')})\n")
        pkgroot = str(tmp_path)
    # the test modules import each other's helpers by name (test_helpers): sourceless modules on the path
    helpers = tmp_path / "mods"
    helpers.mkdir()
    for f in glob.glob(os.path.join(PYC, "*.pyc")):
        os.symlink(f, helpers / os.path.basename(f))
    return dict(os.environ, PYTHONPATH=os.pathsep.join([pkgroot, str(helpers), ROOT]))


MODULES = ["test_autograd", "test_bindings", "test_creations", "test_criterions", "test_functions", "test_utils"]


@pytest.mark.parametrize("binding", ["ctypes", "pybind11"])
@pytest.mark.parametrize("name", MODULES)
def test_reference_python_binding_tests_run_on_the_engine(name, binding, tmp_path):
    path = os.path.join(PYC, name + ".pyc")
    if not os.path.exists(path):
        pytest.skip("tests/pydropin/_pyc not built (needs /root/reference: __graft_entry__.build())")
    runner = tmp_path / "run_module.py"
    runner.write_text(RUNNER)
    r = subprocess.run([sys.executable, str(runner), name, path], capture_output=True, text=True, timeout=900,
                       env=_env(tmp_path, binding), cwd=str(tmp_path))
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "FAILED 0" in r.stdout, tail


@pytest.mark.parametrize("binding", ["ctypes", "pybind11"])
def test_reference_pytorch_loss_example_runs_on_the_engine(binding, tmp_path):
    path = os.path.join(PYC, "pytorch_loss.pyc")
    if not os.path.exists(path):
        pytest.skip("tests/pydropin/_pyc not built (needs /root/reference: __graft_entry__.build())")
    r = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=900, env=_env(tmp_path, binding),
                       cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "Grad has shape" in r.stdout, r.stdout[-2000:]


ROCM_POINTER = r'''
import sys
import numpy as np
import torch
import gtn
assert gtn.__file__.endswith("__init__.pyc") and "_ext" in gtn.__file__, gtn.__file__   # the reference's package
T, C, U = 40, 9, 5
g0 = torch.Generator().manual_seed(3)
em = torch.randn(T, C, generator=g0).cuda()
target = torch.randint(1, C, (U,), generator=g0).tolist()
# the emissions graph takes its weights from the ROCm tensor's data_ptr() (the reference's set_weights(uintptr_t),
# _graph.cpp:86-91 -- there a host address; pytorch_loss.py:53-61 has to go through .cpu() for it)
e = gtn.linear_graph(T, C, True)
e.set_weights(em.data_ptr())
# the target acceptor, as examples/pytorch_loss.py builds it
L = 2 * U + 1
t = gtn.Graph(False)
for l in range(L):
    t.add_node(l == 0, l == L - 1 or l == L - 2)
    label = target[l // 2] if l % 2 else 0
    t.add_arc(l, l, label)
    if l > 0:
        t.add_arc(l - 1, l, label)
    if l % 2 and l > 1 and label != target[l // 2 - 1]:
        t.add_arc(l - 2, l, label)
t.arc_sort(True)
loss = gtn.negate(gtn.forward_score(gtn.intersect(t, e)))
gtn.backward(loss)
got = loss.item()
lp = em.log_softmax(1).double().cpu()
want = torch.nn.functional.ctc_loss(lp.unsqueeze(1), torch.tensor([target]), torch.tensor([T]), torch.tensor([U]),
                                    reduction="sum").item() - float(em.double().logsumexp(1).sum())
assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (got, want)
grad = np.asarray(e.grad().weights_to_numpy()).reshape(T, C)
emc = em.double().cpu().requires_grad_(True)
ref = torch.nn.functional.ctc_loss(emc.log_softmax(1).unsqueeze(1), torch.tensor([target]), torch.tensor([T]),
                                   torch.tensor([U]), reduction="sum") - emc.logsumexp(1).sum()
ref.backward()
assert np.abs(grad - emc.grad.numpy()).max() <= 1e-4, np.abs(grad - emc.grad.numpy()).max()
print("ROCM_POINTER_OK", got, want)
'''


def test_reference_pybind11_binding_takes_a_rocm_data_ptr(tmp_path):
    """the reference's own `Graph.set_weights(uintptr_t)` (bindings/python/gtn/_graph.cpp:86-91), compiled unmodified
    over the engine, handed a ROCm tensor's data_ptr(): the weights stay on the device (gtnx_graph_set_weights sees
    a device address), and the CTC score / gradient of examples/pytorch_loss.py's graph equal torch's float64 ones"""
    env = _env(tmp_path, "pybind11")
    prog = tmp_path / "rocm_pointer.py"
    prog.write_text(ROCM_POINTER)
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "ROCM_POINTER_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
