"""Gradient accumulation through the rational operations on a RETAINED tape (functions.cpp:66-223; addGrad copies,
graph.cpp:91-129): the same little programs are run on the reference backend (tests/test_rational_grads_cpu.py,
which pins the expected numbers below to the unmodified reference) and on the HIP engine
(tests/test_parity_gpu.py::test_rational_ops_retained_backward_twice)."""
import numpy as np


def small(api, w):
    g = api.Graph()
    g.add_node(True)
    g.add_node(False, True)
    g.add_arc(0, 1, 0, 0, float(w[0]))
    g.add_arc(0, 1, 1, 1, float(w[1]))
    return g


def run(api):
    """returns {case: list of gradient vectors}"""
    out = {}
    # clone, backward twice with the tape retained: seed 1 -> g = 1; seed again (out.grad = 2) -> g = 1 + 2 = 3
    g = small(api, [0.5, -1.0])
    c = api.clone(g)
    api.backward(c, True)
    api.backward(c, True)
    out["clone_twice"] = [g.grad().weights_to_numpy().tolist(), c.grad().weights_to_numpy().tolist()]
    # concat of a graph with itself: both slices land in the one input
    g = small(api, [0.25, 2.0])
    cc = api.concat([g, g])
    api.backward(cc, True)
    first = g.grad().weights_to_numpy().tolist()
    api.backward(cc, True)
    out["concat_self_twice"] = [first, g.grad().weights_to_numpy().tolist(), cc.grad().weights_to_numpy().tolist()]
    # a later accumulation into the input must not show in the output's gradient
    g = small(api, [1.0, 1.0])
    u = api.union([g, small(api, [3.0, 4.0])])
    api.backward(u, True)
    before = u.grad().weights_to_numpy().tolist()
    g.add_grad(np.array([10.0, 20.0], np.float32))
    out["union_then_add_grad"] = [before, u.grad().weights_to_numpy().tolist(), g.grad().weights_to_numpy().tolist()]
    return out


EXPECTED = {
    "clone_twice": [[3.0, 3.0], [2.0, 2.0]],
    "concat_self_twice": [[2.0, 2.0], [6.0, 6.0], None],   # (the output's own gradient: checked for length only)
    "union_then_add_grad": [None, None, [11.0, 21.0]],
}
