"""Pins tests/rational_grad_cases.py's expected numbers to the UNMODIFIED reference (oracle/_ref through
tests/refbackend/gtn_ref.py): what retained backward twice / a self-concat / a later addGrad give there."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_expected_numbers_are_the_references():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgtn_ref.so")):
        pytest.skip("needs oracle/_ref (built from /root/reference by __graft_entry__.build())")
    sys.path.insert(0, os.path.join(ROOT, "tests", "refbackend"))
    import gtn_ref
    import rational_grad_cases as rc
    got = rc.run(gtn_ref)
    for name, want in rc.EXPECTED.items():
        for k, w in enumerate(want):
            if w is not None:
                assert got[name][k] == w, (name, k, got[name])
    # the output's gradient is untouched by what happens to the input afterwards
    assert got["union_then_add_grad"][0] == got["union_then_add_grad"][1]
