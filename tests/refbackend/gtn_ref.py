"""TEST INFRASTRUCTURE -- not part of the product.

The Python mirror of this repo (gtn_amd/api.py over the C ABI of include/gtn_amd.h) bound to
oracle/_ref/libgtn_ref.so: the UNMODIFIED reference compiled from /root/reference behind the same C ABI
(oracle/Makefile, oracle/ref_shim.cpp).  Same names as `gtn_amd`; `backend()` answers "reference-cpu".

Users: tests/ (the oracle's pin, the reference's own Python binding tests run against the mirror),
tests/golden/make_golden.py (fixtures), and the `cpu_baseline` legs of bench.py / tools/bench_*.py, which time
the reference on the host cores.  The product package `gtn_amd` has no way of binding to this library: the
seam lives here.

The modules gtn_amd/_capi.py and gtn_amd/api.py are loaded under THIS package name, so gtn_amd/__init__.py --
which loads libgtn_amd.so -- is not executed: a process that imports gtn_ref only never touches the HIP
library.
"""
import importlib.util
import os
import sys
import types

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_SRC = os.path.join(_ROOT, "gtn_amd")
REF_LIB = os.path.join(_ROOT, "oracle", "_ref", "libgtn_ref.so")

_pkg = types.ModuleType("_gtn_ref_pkg")
_pkg.__path__ = [_SRC]  # relative imports of api.py (`from . import _capi`) resolve inside this alias
sys.modules["_gtn_ref_pkg"] = _pkg


def _load(name):
    spec = importlib.util.spec_from_file_location("_gtn_ref_pkg." + name, os.path.join(_SRC, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_gtn_ref_pkg." + name] = mod
    spec.loader.exec_module(mod)
    setattr(_pkg, name, mod)
    return mod


_capi = _load("_capi")
_api_mod = _load("api")

if not os.path.exists(REF_LIB):
    raise ImportError("gtn_ref: %s missing -- build it where /root/reference exists "
                      "(python -c 'import __graft_entry__ as g; g.build()')" % REF_LIB)
_lib = _capi.load(REF_LIB)
_api = _api_mod.make_api(_lib)
globals().update({k: v for k, v in vars(_api).items() if not k.startswith("_")})


def load_txt(text):
    return _api_mod.load_txt(_api, text)


__version__ = _lib.gtnx_version().decode()
