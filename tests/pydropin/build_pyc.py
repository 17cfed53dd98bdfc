"""Compiles the reference's Python binding tests and its PyTorch loss example to bytecode where they lie
(/root/reference/bindings/python/{test,examples}) into tests/pydropin/_pyc/ -- git-ignored, travels to the GPU box
like tests/dropin/_bin and oracle/_ref: the reference's sources never enter the repository, its compiled tests do
(the GPU box has no /root/reference).  Run by __graft_entry__.build() where /root/reference exists."""
import glob
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/bindings/python"


def main():
    out = os.path.join(HERE, "_pyc")
    os.makedirs(out, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(REF, "test", "test_*.py"))) + [os.path.join(REF, "examples", "pytorch_loss.py")]
    for s in srcs:
        name = os.path.splitext(os.path.basename(s))[0]
        py_compile.compile(s, cfile=os.path.join(out, name + ".pyc"), dfile="reference:bindings/python/" + os.path.basename(s),
                           doraise=True)
    print("tests/pydropin/_pyc:", len(srcs), "modules")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(0)
    main()
