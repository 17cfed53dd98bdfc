"""Expected output of the reference's example programs (examples/ctc.cpp, examples/asg.cpp), produced by
building them UNMODIFIED against the reference itself (oracle/_ref/libgtn_ref.so = /root/reference's
sources, oracle/Makefile).  Run where /root/reference exists; the result is committed as
tests/golden/examples_expected.json and compared with what the same programs print on the engine
(tests/test_dropin_gpu.py)."""
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
out = {}
with tempfile.TemporaryDirectory() as d:
    for name in ("ctc", "asg"):
        exe = os.path.join(d, name)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + REF, os.path.join(REF, "examples", name + ".cpp"),
                               os.path.join(ROOT, "oracle", "_ref", "libgtn_ref.so"),
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_ref"), "-pthread", "-o", exe])
        out["ex_" + name] = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "examples_expected.json"), "w"), indent=1)
print(out)
