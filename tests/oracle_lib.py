"""ctypes binding of the plain-C parity oracle (oracle/liboracle.so).
TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = C.CDLL(path)
        vp = C.c_void_p
        L.og_new.restype = vp
        L.og_free.argtypes = [vp]
        L.og_add_node.argtypes = [vp, C.c_int, C.c_int]
        L.og_add_arc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        for f in ("og_num_nodes", "og_num_arcs", "og_num_start", "og_num_accept"):
            getattr(L, f).argtypes = [vp]
        L.og_get_nodes.argtypes = [vp, vp, vp]
        L.og_get_arcs.argtypes = [vp, vp, vp, vp, vp, vp]
        L.og_set_weights.argtypes = [vp, vp]
        L.og_arc_sort.argtypes = [vp, C.c_int]
        L.og_mark_sorted.argtypes = [vp, C.c_int]
        L.og_linear_graph.argtypes = [C.c_int, C.c_int]
        L.og_linear_graph.restype = vp
        L.og_shortest_distance.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.og_shortest_distance_grad.argtypes = [vp, C.c_int, C.c_float, vp]
        L.og_shortest_path.argtypes = [vp, vp, vp, vp]
        L.og_compose.argtypes = [vp, vp, C.c_int]
        L.og_compose.restype = vp
        L.og_grad_info.argtypes = [vp]
        L.og_grad_info.restype = C.POINTER(C.c_int)
        L.og_compose_grad.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.og_ctc_graph.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.og_ctc_graph.restype = vp
        L.og_ctc_loss.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
        _LIB = L
    return _LIB


class OGraph:
    def __init__(self, h=None):
        self.h = h if h is not None else lib().og_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().og_free(self.h)
            self.h = None

    @classmethod
    def from_dict(cls, d):
        g = cls()
        L = lib()
        for s, a in zip(d["start"], d["accept"]):
            L.og_add_node(g.h, int(s), int(a))
        for s, t, i, o, w in zip(d["src"], d["dst"], d["il"], d["ol"], d["w"]):
            L.og_add_arc(g.h, s, t, i, o, w)
        if d.get("sort") == "i":
            L.og_arc_sort(g.h, 0)
        elif d.get("sort") == "o":
            L.og_arc_sort(g.h, 1)
        return g

    @classmethod
    def linear(cls, T, Cn, weights=None):
        g = cls(lib().og_linear_graph(T, Cn))
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            lib().og_set_weights(g.h, w.ctypes.data)
        return g

    @property
    def N(self):
        return lib().og_num_nodes(self.h)

    @property
    def A(self):
        return lib().og_num_arcs(self.h)

    def to_dict(self):
        N, A = self.N, self.A
        st, ac = np.zeros(N, np.uint8), np.zeros(N, np.uint8)
        s, d, i, o = (np.zeros(A, np.int32) for _ in range(4))
        w = np.zeros(A, np.float32)
        lib().og_get_nodes(self.h, st.ctypes.data, ac.ctypes.data)
        lib().og_get_arcs(self.h, s.ctypes.data, d.ctypes.data, i.ctypes.data, o.ctypes.data,
                          w.ctypes.data)
        return {"start": st.tolist(), "accept": ac.tolist(), "src": s.tolist(), "dst": d.tolist(),
                "il": i.tolist(), "ol": o.tolist(), "w": [float(x) for x in w], "sort": None}

    def shortest_distance(self, tropical=False):
        """-> score (float) or None on 'cycle / disconnected'"""
        out = C.c_float()
        err = lib().og_shortest_distance(self.h, int(tropical), C.byref(out), None, None, None)
        return None if err else out.value

    def shortest_distance_grad(self, tropical=False, delta=1.0):
        g = np.zeros(self.A, np.float32)
        err = lib().og_shortest_distance_grad(self.h, int(tropical), delta, g.ctypes.data)
        return None if err else g

    def shortest_path(self):
        arcs = np.zeros(self.N + 1, np.int32)
        n, has = C.c_int(), C.c_int()
        err = lib().og_shortest_path(self.h, arcs.ctypes.data, C.byref(n), C.byref(has))
        if err:
            return None
        return arcs[: n.value].tolist(), bool(has.value)

    def compose(self, other, mode="compose"):
        return OGraph(lib().og_compose(self.h, other.h, 1 if mode == "intersect" else 0))

    def grad_info(self):
        p = lib().og_grad_info(self.h)
        return np.ctypeslib.as_array(p, shape=(self.A, 2)).copy() if self.A else np.zeros((0, 2), np.int32)

    def compose_grad(self, deltas, A1, A2):
        g1, g2 = np.zeros(A1, np.float32), np.zeros(A2, np.float32)
        d = np.ascontiguousarray(deltas, dtype=np.float32)
        lib().og_compose_grad(self.h, d.ctypes.data, A1, A2, g1.ctypes.data, g2.ctypes.data)
        return g1, g2


def ctc_loss(emissions, target):
    """-> (loss, grad[T,C]) via the oracle (benchmarks/ctc.cpp:150-160 semantics)"""
    em = np.ascontiguousarray(emissions, dtype=np.float32)
    tg = np.ascontiguousarray(target, dtype=np.int32)
    T, Cn = em.shape
    loss = C.c_float()
    grad = np.zeros_like(em)
    err = lib().og_ctc_loss(em.ctypes.data, T, Cn, tg.ctypes.data, tg.size, C.byref(loss),
                            grad.ctypes.data)
    if err:
        raise ValueError("Graph has a cycle, self-loop or is disconnected!")
    return loss.value, grad
