"""The Python side of the drop-in ON THE ENGINE: the reference's own Python binding tests
(bindings/python/test/test_*.py: graph construction, weights, formats, functions incl. the rational ops,
autograd, criteria, parallel forms) and its PyTorch CTC loss example (examples/pytorch_loss.py:19-102), unmodified,
with `import gtn` resolving to gtn_amd over libgtn_amd.so -- the MI355X engine, not the reference library
(tests/test_pydropin_cpu.py runs the same sources on the reference backend).  The GPU box has no /root/reference:
the tests are run from their bytecode (tests/pydropin/_pyc, compiled where the sources lie by
tests/pydropin/build_pyc.py: the compiled form of the reference's tests travels like tests/dropin/_bin)."""
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


def _env(tmp_path):
    pkg = tmp_path / "gtn"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(
        "import gtn_amd as _g\n"
        "globals().update({k: getattr(_g, k) for k in dir(_g) if not k.startswith('__')})\n")
    # the test modules import each other's helpers by name (test_helpers): sourceless modules on the path
    helpers = tmp_path / "mods"
    helpers.mkdir()
    for f in glob.glob(os.path.join(PYC, "*.pyc")):
        os.symlink(f, helpers / os.path.basename(f))
    return dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), str(helpers), ROOT]))


MODULES = ["test_autograd", "test_bindings", "test_creations", "test_criterions", "test_functions", "test_utils"]


@pytest.mark.parametrize("name", MODULES)
def test_reference_python_binding_tests_run_on_the_engine(name, tmp_path):
    path = os.path.join(PYC, name + ".pyc")
    if not os.path.exists(path):
        pytest.skip("tests/pydropin/_pyc not built (needs /root/reference: __graft_entry__.build())")
    runner = tmp_path / "run_module.py"
    runner.write_text(RUNNER)
    r = subprocess.run([sys.executable, str(runner), name, path], capture_output=True, text=True, timeout=900, env=_env(tmp_path),
                       cwd=str(tmp_path))
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "FAILED 0" in r.stdout, tail


def test_reference_pytorch_loss_example_runs_on_the_engine(tmp_path):
    path = os.path.join(PYC, "pytorch_loss.pyc")
    if not os.path.exists(path):
        pytest.skip("tests/pydropin/_pyc not built (needs /root/reference: __graft_entry__.build())")
    r = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=900, env=_env(tmp_path), cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "Grad has shape" in r.stdout, r.stdout[-2000:]
