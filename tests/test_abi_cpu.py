"""CPU-only checks: the C-ABI library loads without a GPU, exports every symbol
include/gtn_amd.h declares, host-side graph bookkeeping behaves like the
reference, and device operations fail loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gtn_amd.h")).read()
    return sorted(set(re.findall(r"\b(gtnx_[a-z0-9_]+)\s*\(", txt)) - {"gtnx_grad_fn", "gtnx_status_t", "gtnx_graph_t"})


def test_exports_every_declared_symbol():
    from gtn_amd import _capi
    lib = _capi.load()
    names = header_symbols()
    assert len(names) > 80
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_capi.ALL_SYMBOLS), set(names) ^ set(_capi.ALL_SYMBOLS)


def test_host_graph_semantics(gtn):
    # test/graph_test.cpp:35-105 (construction, counts, copies alias, deepCopy detaches)
    g = gtn.Graph()
    assert g.add_node(True) == 0 and g.add_node() == 1 and g.add_node(False, True) == 2
    assert g.add_arc(0, 1, 0) == 0 and g.add_arc(0, 2, 1, 2, 2.5) == 1
    assert (g.num_nodes(), g.num_arcs(), g.num_start(), g.num_accept()) == (3, 2, 1, 1)
    assert g.start() == [0] and g.accept() == [2]
    assert g.out(0) == [0, 1] and g.in_(2) == [1] and g.arc(1) == (0, 2, 1, 2, 2.5)
    alias = g.copy()
    alias.add_node()
    assert g.num_nodes() == 4                      # copies share structure (graph_test.cpp:56-77)
    deep = g.deep_copy()
    deep.add_node()
    assert g.num_nodes() == 4 and deep.num_nodes() == 5
    assert gtn.equal(g, g.deep_copy())
    with pytest.raises(ValueError):
        g.item()                                   # graph.cpp:70-73
    with pytest.raises(RuntimeError):
        g.grad()                                   # graph.cpp:82-87
    g.add_grad([1.0, 2.0])
    g.add_grad([1.0, 2.0])
    assert g.grad().weights_to_list() == [2.0, 4.0]
    with pytest.raises(RuntimeError):
        g.add_grad([1.0])                          # graph.cpp:93-95
    g.zero_grad()
    assert not g.is_grad_available()
    g.calc_grad = False
    g.add_grad([1.0, 2.0])
    assert not g.is_grad_available()
    with pytest.raises(IndexError):
        g.add_arc(0, 99, 0)


def test_arc_sort_and_flags(gtn):
    g = gtn.Graph()
    g.add_node(True)
    g.add_node(False, True)
    for l in (2, 0, 1):
        g.add_arc(0, 1, l, 2 - l)
    assert not g.ilabel_sorted() and not g.olabel_sorted()
    g.arc_sort()
    assert g.ilabel_sorted() and g.out(0) == [1, 2, 0]
    g.arc_sort(True)
    assert g.olabel_sorted() and not g.ilabel_sorted() and g.out(0) == [0, 2, 1]
    g.add_node()
    assert not g.olabel_sorted()                  # graph.cpp:42-43


def test_linear_graph_is_implicit_but_inspectable(gtn):
    # test/creations_test.cpp:32-53
    g = gtn.linear_graph(3, 4)
    assert (g.num_nodes(), g.num_arcs(), g.start(), g.accept()) == (4, 12, [0], [3])
    assert g.ilabel_sorted() and g.olabel_sorted()
    s, d, il, ol, w = g.arcs()
    assert s.tolist() == [0] * 4 + [1] * 4 + [2] * 4 and d.tolist() == [1] * 4 + [2] * 4 + [3] * 4
    assert il.tolist() == [0, 1, 2, 3] * 3 and w.tolist() == [0.0] * 12
    g.set_weights(np.arange(12, dtype=np.float32))
    assert g.arc(7) == (1, 2, 3, 3, 7.0) and g.out(1) == [4, 5, 6, 7] and g.in_(0) == []
    s = gtn.scalar_graph(2.5)
    assert s.item() == 2.5 and s.labels_to_list() == [-1]


def test_isomorphic(gtn):
    a = gtn.load_txt("0\n2\n0 1 0\n1 1 0\n1 2 1\n")
    b = gtn.Graph()
    for st, ac in ((False, True), (False, False), (True, False)):   # permuted node ids
        b.add_node(st, ac)
    b.add_arc(2, 1, 0)
    b.add_arc(1, 1, 0)
    b.add_arc(1, 0, 1)
    assert gtn.isomorphic(a, b) and not gtn.equal(a, b)
    b.add_arc(1, 0, 1)
    assert not gtn.isomorphic(a, b)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_device_ops_fail_loudly_without_gpu(gtn):
    g = gtn.linear_graph(2, 2)
    for fn in (gtn.forward_score, gtn.viterbi_score, gtn.viterbi_path):
        with pytest.raises(RuntimeError, match="no HIP device"):
            fn(g)
    with pytest.raises(RuntimeError, match="no HIP device"):
        gtn.compose(g, g)
    with pytest.raises(RuntimeError, match="no HIP device"):
        gtn.negate(gtn.scalar_graph(1.0))


def test_route_table_names_without_a_gpu():
    """gtnx_debug_route_name (the rows of gtn_amd/csrc/ops_symbolic.cpp) answers on a GPU-less host; a built graph has
    no symbolic route"""
    import ctypes as C
    from gtn_amd import _capi
    lib = _capi.load()
    names = []
    for r in range(6):
        buf = C.create_string_buffer(32)
        assert lib.gtnx_debug_route_name(r, buf, 32) == 0
        names.append(buf.value.decode())
    assert names == ["band", "pair", "dense_mfma", "dense", "maxplus", "walk"]
    import gtn_amd
    g = gtn_amd.Graph()
    g.add_node(True, True)
    assert gtn_amd.debug_symbolic_route(g) is None
