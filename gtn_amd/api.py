"""Host-side mirror of the reference's Python interface over the C ABI.

Names, argument meaning and error behaviour follow bindings/python/gtn
(`_graph.cpp:20-110`, `_functions.cpp:18-200`, `_autograd.cpp`, `_creations.cpp`,
`_utils.cpp`): `Graph.add_node/add_arc/...`, `compose`, `intersect`,
`forward_score`, `viterbi_score`, `viterbi_path`, `negate/add/subtract`,
`backward`, `linear_graph`, `scalar_graph`, `equal`, `isomorphic`.  As in the
reference binding every graph function also accepts lists of graphs
(`_functions.cpp:84-135`); here a list runs as ONE batched device launch per
kernel (`gtnx_*_n`) instead of `parallelMap` over CPU threads.

`make_api(lib)` binds this interface to one loaded library, so the tests can
hold the product (libgtn_amd.so) and the reference shim (libgtn_ref.so) side by
side.
"""
import ctypes as C
import types

import numpy as np

from . import _capi

_EXC = {
    _capi.INVALID_ARGUMENT: ValueError,   # pybind11 maps std::invalid_argument -> ValueError
    _capi.LOGIC_ERROR: RuntimeError,      # std::logic_error -> RuntimeError
    _capi.RUNTIME_ERROR: RuntimeError,
    _capi.OUT_OF_RANGE: IndexError,
    _capi.DEVICE_ERROR: RuntimeError,
}


class GtnError(RuntimeError):
    pass


def _is_seq(x):
    return isinstance(x, (list, tuple))


def make_api(lib):
    ns = types.SimpleNamespace()
    ns.lib = lib
    ns.epsilon = -1

    def check(status):
        if status != 0:
            msg = lib.gtnx_last_error().decode("utf-8", "replace")
            raise _EXC.get(status, GtnError)(msg)

    ns.check = check

    def _harr(graphs):
        arr = (C.c_void_p * len(graphs))(*[g._h for g in graphs])
        return arr

    def _wrap_many(arr, n):
        return [Graph._from_handle(arr[i]) for i in range(n)]

    def _as_dev_ptr(x):
        """torch ROCm tensor / object with data_ptr() / int address -> int"""
        if hasattr(x, "data_ptr"):
            return int(x.data_ptr())
        return int(x)

    class Graph:
        """gtn.Graph (bindings/python/gtn/_graph.cpp:23-108)."""

        __slots__ = ("_h", "__weakref__")

        def __init__(self, calc_grad=True):
            h = C.c_void_p()
            check(lib.gtnx_graph_create(1 if calc_grad else 0, C.byref(h)))
            self._h = h.value

        @classmethod
        def _from_handle(cls, h):
            g = object.__new__(cls)
            g._h = h
            return g

        def __del__(self):
            h = getattr(self, "_h", None)
            if h:
                try:
                    lib.gtnx_graph_destroy(h)
                except Exception:
                    pass
                self._h = None

        # -- construction ---------------------------------------------------
        def add_node(self, start=False, accept=False):
            i = C.c_int()
            check(lib.gtnx_graph_add_node(self._h, int(bool(start)), int(bool(accept)), C.byref(i)))
            return i.value

        def add_arc(self, src_node, dst_node, ilabel=None, olabel=None, weight=0.0, label=None):
            """add_arc(src_node, dst_node, label) / add_arc(src_node, dst_node, ilabel, olabel, weight=0.0)
            (the binding's two overloads, _graph.cpp:30-43)"""
            if label is not None:
                if ilabel is not None or olabel is not None:
                    raise TypeError("add_arc: give either label or ilabel / olabel")
                ilabel = label
            if ilabel is None:
                raise TypeError("add_arc: missing label")
            if olabel is None:
                olabel = ilabel
            i = C.c_int()
            check(lib.gtnx_graph_add_arc(self._h, int(src_node), int(dst_node), int(ilabel),
                                         int(olabel), float(weight), C.byref(i)))
            return i.value

        def add_nodes(self, start, accept):
            """bulk add_node (extension; same order as successive calls)"""
            s = np.ascontiguousarray(start, dtype=np.uint8)
            a = np.ascontiguousarray(accept, dtype=np.uint8)
            assert s.shape == a.shape
            check(lib.gtnx_graph_add_nodes(self._h, s.size, s.ctypes.data, a.ctypes.data))

        def add_arcs(self, src, dst, ilabel, olabel=None, weight=None):
            """bulk add_arc (extension)"""
            src = np.ascontiguousarray(src, dtype=np.int32)
            dst = np.ascontiguousarray(dst, dtype=np.int32)
            il = np.ascontiguousarray(ilabel, dtype=np.int32)
            ol = il if olabel is None else np.ascontiguousarray(olabel, dtype=np.int32)
            w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
            check(lib.gtnx_graph_add_arcs(self._h, src.size, src.ctypes.data, dst.ctypes.data,
                                          il.ctypes.data, ol.ctypes.data,
                                          None if w is None else w.ctypes.data))

        # -- counts -----------------------------------------------------------
        def _count(self, fn):
            v = C.c_int64()
            check(fn(self._h, C.byref(v)))
            return v.value

        def num_arcs(self):
            return self._count(lib.gtnx_graph_num_arcs)

        def num_nodes(self):
            return self._count(lib.gtnx_graph_num_nodes)

        def num_start(self):
            return self._count(lib.gtnx_graph_num_start)

        def num_accept(self):
            return self._count(lib.gtnx_graph_num_accept)

        def item(self):
            v = C.c_float()
            check(lib.gtnx_graph_item(self._h, C.byref(v)))
            return v.value

        # -- sorting ------------------------------------------------------------
        def arc_sort(self, olabel=False):
            check(lib.gtnx_graph_arc_sort(self._h, int(bool(olabel))))

        def mark_arc_sorted(self, olabel=False):
            check(lib.gtnx_graph_mark_arc_sorted(self._h, int(bool(olabel))))

        def ilabel_sorted(self):
            v = C.c_int()
            check(lib.gtnx_graph_ilabel_sorted(self._h, C.byref(v)))
            return bool(v.value)

        def olabel_sorted(self):
            v = C.c_int()
            check(lib.gtnx_graph_olabel_sorted(self._h, C.byref(v)))
            return bool(v.value)

        # -- weights ------------------------------------------------------------
        def weights(self):
            """address of the live host weight buffer (as the reference returns)"""
            p = _capi.c_f32_p()
            check(lib.gtnx_graph_weights(self._h, 1, C.byref(p)))
            return C.cast(p, C.c_void_p).value or 0

        def weights_to_numpy(self):
            out = np.empty(self.num_arcs(), dtype=np.float32)
            check(lib.gtnx_graph_get_weights(self._h, out.ctypes.data))
            return out

        def weights_to_list(self):
            return self.weights_to_numpy().tolist()

        def weights_device(self):
            """device address of the weight buffer (extension)"""
            p = C.c_void_p()
            check(lib.gtnx_graph_weights_device(self._h, C.byref(p)))
            return p.value or 0

        def set_weights(self, weights):
            """list / numpy array / host address (reference overloads), or a
            ROCm torch tensor, which is copied device-to-device."""
            if hasattr(weights, "is_cuda") and weights.is_cuda:
                if weights.numel() != self.num_arcs() or str(weights.dtype) != "torch.float32":
                    raise ValueError("set_weights: need numArcs float32 values")
                w = weights.contiguous()
                check(lib.gtnx_graph_set_weights_device(self._h, w.data_ptr()))
                return
            if hasattr(weights, "detach"):
                weights = weights.detach().cpu().numpy()
            if isinstance(weights, int):
                check(lib.gtnx_graph_set_weights(self._h, weights))
                return
            w = np.ascontiguousarray(weights, dtype=np.float32)
            if w.size != self.num_arcs():
                raise ValueError("set_weights: need numArcs values")
            check(lib.gtnx_graph_set_weights(self._h, w.ctypes.data))

        def set_weights_device(self, ptr):
            check(lib.gtnx_graph_set_weights_device(self._h, _as_dev_ptr(ptr)))

        def labels_to_list(self, ilabel=True):
            out = np.empty(self.num_arcs(), dtype=np.int32)
            check(lib.gtnx_graph_labels_to_array(self._h, out.ctypes.data, int(bool(ilabel))))
            return out.tolist()

        # -- grad ----------------------------------------------------------------
        @property
        def calc_grad(self):
            v = C.c_int()
            check(lib.gtnx_graph_calc_grad(self._h, C.byref(v)))
            return bool(v.value)

        @calc_grad.setter
        def calc_grad(self, value):
            check(lib.gtnx_graph_set_calc_grad(self._h, int(bool(value))))

        def is_grad_available(self):
            v = C.c_int()
            check(lib.gtnx_graph_is_grad_available(self._h, C.byref(v)))
            return bool(v.value)

        def grad(self):
            h = C.c_void_p()
            check(lib.gtnx_graph_grad(self._h, C.byref(h)))
            return Graph._from_handle(h.value)

        def zero_grad(self):
            check(lib.gtnx_graph_zero_grad(self._h))

        def add_grad(self, other):
            if isinstance(other, Graph):
                check(lib.gtnx_graph_add_grad_graph(self._h, other._h))
            else:
                v = np.ascontiguousarray(other, dtype=np.float32)
                check(lib.gtnx_graph_add_grad(self._h, v.ctypes.data, v.size))

        def id(self):
            v = C.c_size_t()
            check(lib.gtnx_graph_id(self._h, C.byref(v)))
            return v.value

        # -- structure inspection (C++ accessors, graph.h:330-414) -----------------
        def start(self):
            out = np.empty(self.num_start(), dtype=np.int32)
            check(lib.gtnx_graph_get_start(self._h, out.ctypes.data))
            return out.tolist()

        def accept(self):
            out = np.empty(self.num_accept(), dtype=np.int32)
            check(lib.gtnx_graph_get_accept(self._h, out.ctypes.data))
            return out.tolist()

        def is_start(self, n):
            v = C.c_int()
            check(lib.gtnx_graph_is_start(self._h, n, C.byref(v)))
            return bool(v.value)

        def is_accept(self, n):
            v = C.c_int()
            check(lib.gtnx_graph_is_accept(self._h, n, C.byref(v)))
            return bool(v.value)

        def make_accept(self, n):
            check(lib.gtnx_graph_make_accept(self._h, n))

        def out(self, n):
            c = C.c_int64()
            check(lib.gtnx_graph_num_out(self._h, n, C.byref(c)))
            out = np.empty(c.value, dtype=np.int32)
            check(lib.gtnx_graph_get_out(self._h, n, out.ctypes.data))
            return out.tolist()

        def in_(self, n):
            c = C.c_int64()
            check(lib.gtnx_graph_num_in(self._h, n, C.byref(c)))
            out = np.empty(c.value, dtype=np.int32)
            check(lib.gtnx_graph_get_in(self._h, n, out.ctypes.data))
            return out.tolist()

        def arcs(self):
            """(src, dst, ilabel, olabel, weight) numpy arrays in arc-id order"""
            A = self.num_arcs()
            s, d, i, o = (np.empty(A, dtype=np.int32) for _ in range(4))
            check(lib.gtnx_graph_get_arcs(self._h, s.ctypes.data, d.ctypes.data,
                                          i.ctypes.data, o.ctypes.data))
            return s, d, i, o, self.weights_to_numpy()

        def arc(self, a):
            s, d, i, o, w = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_float()
            check(lib.gtnx_graph_get_arc(self._h, a, C.byref(s), C.byref(d), C.byref(i),
                                         C.byref(o), C.byref(w)))
            return s.value, d.value, i.value, o.value, w.value

        def set_weight(self, a, w):
            check(lib.gtnx_graph_set_weight(self._h, a, float(w)))

        def copy(self):
            h = C.c_void_p()
            check(lib.gtnx_graph_copy(self._h, C.byref(h)))
            return Graph._from_handle(h.value)

        def deep_copy(self):
            h = C.c_void_p()
            check(lib.gtnx_graph_deep_copy(self._h, C.byref(h)))
            return Graph._from_handle(h.value)

        def __repr__(self):
            host = _host(required=False)
            if host is not None:  # operator<< of utils.h (what the binding prints)
                need = C.c_size_t()
                hcheck(host.gtnh_repr(self._h, None, 0, C.byref(need)))
                buf = C.create_string_buffer(need.value)
                hcheck(host.gtnh_repr(self._h, buf, need.value, None))
                return buf.value.decode()
            s, d, i, o, w = self.arcs()
            lines = [" ".join(map(str, self.start())), " ".join(map(str, self.accept()))]
            lines += [f"{a} {b} {c} {e} {f:g}" for a, b, c, e, f in zip(s, d, i, o, w)]
            return "\n".join(lines)

    ns.Graph = Graph

    # ---------------------------------------------------------------- host-side builders / formats
    # gtn_amd/hostops: C entry points over include/gtn's header-only clone / project / concat / closure /
    # union / remove / sample / randEquivalent / load / save / draw.  One copy sits next to (and is linked
    # against) each C-ABI library; RTLD_DEEPBIND keeps its gtnx_* calls on that library even when the
    # product and the reference shim are loaded side by side.
    _hostlib = []

    def _host(required=True):
        if not _hostlib:
            import os
            path = os.path.join(os.path.dirname(os.path.abspath(getattr(lib, "_path", ""))), "libgtn_hostops.so")
            h = None
            if os.path.exists(path):
                h = C.CDLL(path, mode=getattr(os, "RTLD_DEEPBIND", 0) | getattr(os, "RTLD_NOW", 2))
                vp, vpp, ip = C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)
                for name, args in {
                    "gtnh_clone": [vp, C.c_int, vpp], "gtnh_concat": [vpp, C.c_int, vpp], "gtnh_closure": [vp, vpp],
                    "gtnh_union": [vpp, C.c_int, vpp], "gtnh_remove": [vp, C.c_int, C.c_int, vpp],
                    "gtnh_sample": [vp, C.c_size_t, vpp],
                    "gtnh_rand_equivalent": [vp, vp, C.c_size_t, C.c_double, C.c_size_t, ip],
                    "gtnh_load": [C.c_char_p, vpp], "gtnh_save": [C.c_char_p, vp],
                    "gtnh_loadtxt": [C.c_char_p, vpp], "gtnh_savetxt": [C.c_char_p, vp],
                    "gtnh_write_dot": [vp, C.c_char_p, ip, C.POINTER(C.c_char_p), C.c_int, ip, C.POINTER(C.c_char_p),
                                       C.c_int],
                    "gtnh_repr": [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)],
                }.items():
                    fn = getattr(h, name)
                    fn.argtypes = args
                    fn.restype = C.c_int
                h.gtnh_last_error.restype = C.c_char_p
            _hostlib.append(h)
        if _hostlib[0] is None and required:
            raise ImportError("gtn_amd: libgtn_hostops.so not found next to the C-ABI library "
                              "(python -c 'import __graft_entry__ as g; g.build()')")
        return _hostlib[0]

    def hcheck(status):
        if status != 0:
            msg = _host().gtnh_last_error().decode("utf-8", "replace")
            raise _EXC.get(status, GtnError)(msg)

    def _host_unary(name, *extra):
        def one(g, *args):
            h = C.c_void_p()
            hcheck(getattr(_host(), name)(g._h, *args, C.byref(h)))
            return Graph._from_handle(h.value)
        return one

    def _host_nary(name):
        def one(graphs):
            graphs = list(graphs)
            arr = (C.c_void_p * max(len(graphs), 1))(*[g._h for g in graphs])
            h = C.c_void_p()
            hcheck(getattr(_host(), name)(arr, len(graphs), C.byref(h)))
            return Graph._from_handle(h.value)
        return one

    _clone1 = _host_unary("gtnh_clone")
    _closure1 = _host_unary("gtnh_closure")
    _remove1 = _host_unary("gtnh_remove")
    _concat1 = _host_nary("gtnh_concat")
    _union1 = _host_nary("gtnh_union")

    def clone(g, projection=0):
        """clone(g) (_functions.cpp:78-83); lists map element-wise"""
        if _is_seq(g):
            return [_clone1(x, int(projection)) for x in g]
        return _clone1(g, int(projection))

    def project_input(g):
        return [_clone1(x, 1) for x in g] if _is_seq(g) else _clone1(g, 1)

    def project_output(g):
        return [_clone1(x, 2) for x in g] if _is_seq(g) else _clone1(g, 2)

    def closure(g):
        return [_closure1(x) for x in g] if _is_seq(g) else _closure1(g)

    def _bcast(seq, n):
        seq = list(seq)
        if len(seq) == n:
            return seq
        if len(seq) == 1:
            return seq * n
        raise RuntimeError("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector")

    def concat(a, b=None):
        """concat(g1, g2) / concat(graphs1, graphs2) / concat(graphs) / concat(list of lists)
        (_functions.cpp:36-68)"""
        if b is not None:
            if _is_seq(a) or _is_seq(b):
                la = list(a) if _is_seq(a) else [a]
                lb = list(b) if _is_seq(b) else [b]
                n = max(len(la), len(lb))
                return [_concat1([x, y]) for x, y in zip(_bcast(la, n), _bcast(lb, n))]
            return _concat1([a, b])
        if len(a) and _is_seq(a[0]):
            return [_concat1(x) for x in a]
        return _concat1(a)

    def union(graphs):
        """union(graphs) / union(list of lists) (_functions.cpp:221-233)"""
        if len(graphs) and _is_seq(graphs[0]):
            return [_union1(x) for x in graphs]
        return _union1(graphs)

    def remove(g, ilabel=None, olabel=None, label=None, labels=None):
        """remove(g, label=epsilon) / remove(g, ilabel, olabel) / remove(graphs, labels=[epsilon])
        (_functions.cpp:179-203)"""
        if _is_seq(g):
            lab = labels if labels is not None else (ilabel if ilabel is not None else [ns.epsilon])
            lab = list(lab) if _is_seq(lab) else [lab]
            n = max(len(g), len(lab))
            return [_remove1(x, int(l), int(l)) for x, l in zip(_bcast(g, n), _bcast(lab, n))]
        if label is not None:
            ilabel = olabel = label
        if ilabel is None:
            ilabel = ns.epsilon
        if olabel is None:
            olabel = ilabel
        return _remove1(g, int(ilabel), int(olabel))

    def sample(g, max_length=1000):
        h = C.c_void_p()
        hcheck(_host().gtnh_sample(g._h, int(max_length), C.byref(h)))
        return Graph._from_handle(h.value)

    def rand_equivalent(g1, g2, num_samples=1000, tol=1e-4, max_length=1000):
        v = C.c_int()
        hcheck(_host().gtnh_rand_equivalent(g1._h, g2._h, int(num_samples), float(tol), int(max_length), C.byref(v)))
        return bool(v.value)

    def load(file_name):
        h = C.c_void_p()
        hcheck(_host().gtnh_load(str(file_name).encode(), C.byref(h)))
        return Graph._from_handle(h.value)

    def save(file_name, graph):
        hcheck(_host().gtnh_save(str(file_name).encode(), graph._h))

    def loadtxt(file_name):
        h = C.c_void_p()
        hcheck(_host().gtnh_loadtxt(str(file_name).encode(), C.byref(h)))
        return Graph._from_handle(h.value)

    def savetxt(file_name, graph):
        hcheck(_host().gtnh_savetxt(str(file_name).encode(), graph._h))

    def write_dot(g, file_name, isymbols={}, osymbols={}):
        def table(m):
            keys = (C.c_int * max(len(m), 1))(*[int(k) for k in m])
            names = (C.c_char_p * max(len(m), 1))(*[str(v).encode() for v in m.values()])
            return keys, names, len(m)
        ik, inames, ni = table(isymbols)
        ok, onames, no = table(osymbols)
        hcheck(_host().gtnh_write_dot(g._h, str(file_name).encode(), ik, inames, ni, ok, onames, no))

    def draw(graph, file_name, isymbols={}, osymbols={}):
        """bindings/python/gtn/__init__.py:22-26: dot -> picture through graphviz"""
        import os
        import subprocess
        import tempfile
        ext = os.path.splitext(file_name)[1]
        with tempfile.NamedTemporaryFile() as tmpf:
            write_dot(graph, tmpf.name, isymbols, osymbols)
            subprocess.check_call(["dot", "-T" + ext[1:], tmpf.name, "-o", file_name])

    for _n, _f in (("clone", clone), ("project_input", project_input), ("project_output", project_output),
                   ("closure", closure), ("concat", concat), ("union", union), ("remove", remove),
                   ("sample", sample), ("rand_equivalent", rand_equivalent), ("load", load), ("save", save),
                   ("loadtxt", loadtxt), ("savetxt", savetxt), ("write_dot", write_dot), ("draw", draw)):
        setattr(ns, _n, _f)

    # ---------------------------------------------------------------- functions
    def _unary(single, batched):
        def fn(g):
            if _is_seq(g):
                n = len(g)
                out = (C.c_void_p * n)()
                if n:
                    check(batched(_harr(g), n, out))
                return _wrap_many(out, n)
            h = C.c_void_p()
            check(single(g._h, C.byref(h)))
            return Graph._from_handle(h.value)
        return fn

    def _binary(single, batched):
        def fn(a, b):
            if _is_seq(a) or _is_seq(b):
                la = list(a) if _is_seq(a) else [a]
                lb = list(b) if _is_seq(b) else [b]
                n = max(len(la), len(lb))
                out = (C.c_void_p * n)()
                if n:
                    check(batched(_harr(la), len(la), _harr(lb), len(lb), out))
                return _wrap_many(out, n)
            h = C.c_void_p()
            check(single(a._h, b._h, C.byref(h)))
            return Graph._from_handle(h.value)
        return fn

    ns.negate = _unary(lib.gtnx_negate, lib.gtnx_negate_n)
    ns.add = _binary(lib.gtnx_add, lib.gtnx_add_n)
    ns.subtract = _binary(lib.gtnx_subtract, lib.gtnx_subtract_n)
    ns.compose = _binary(lib.gtnx_compose, lib.gtnx_compose_n)
    ns.intersect = _binary(lib.gtnx_intersect, lib.gtnx_intersect_n)
    ns.forward_score = _unary(lib.gtnx_forward_score, lib.gtnx_forward_score_n)
    ns.viterbi_score = _unary(lib.gtnx_viterbi_score, lib.gtnx_viterbi_score_n)
    ns.viterbi_path = _unary(lib.gtnx_viterbi_path, lib.gtnx_viterbi_path_n)

    def backward(g, grad_or_retain=None, retain_graph=False):
        """backward(g, retain_graph=False) / backward(g, grad, retain_graph=False);
        lists run batched (bindings/python/gtn/_autograd.cpp)."""
        grad = None
        if isinstance(grad_or_retain, Graph):
            grad = grad_or_retain
        elif grad_or_retain is not None:
            retain_graph = grad_or_retain
        if _is_seq(g):
            # backward(graphs, retain_graphs=[0]) / backward(graphs, grads, retain_graphs=[0]): parallelMap
            # with size-1 lists broadcast (_autograd.cpp:27-60)
            grads = None
            if _is_seq(grad_or_retain) and grad_or_retain and isinstance(grad_or_retain[0], Graph):
                grads = list(grad_or_retain)
            elif _is_seq(grad_or_retain):
                retain_graph = grad_or_retain
            if grads is not None:
                rl = list(retain_graph) if _is_seq(retain_graph) else [retain_graph]
                n = max(len(g), len(grads), len(rl))
                for x, gr, r in zip(_bcast(g, n), _bcast(grads, n), _bcast(rl, n)):
                    check(lib.gtnx_backward_with_grad(x._h, gr._h, int(bool(r))))
                return
            if _is_seq(retain_graph):
                retain_graph = retain_graph[0] if retain_graph else False
            if len(g):
                check(lib.gtnx_backward_n(_harr(g), len(g), int(bool(retain_graph))))
            return
        if grad is not None:
            check(lib.gtnx_backward_with_grad(g._h, grad._h, int(bool(retain_graph))))
        else:
            check(lib.gtnx_backward(g._h, int(bool(retain_graph))))

    ns.backward = backward

    def scalar_graph(val, calc_grad=True):
        h = C.c_void_p()
        check(lib.gtnx_scalar_graph(float(val), int(bool(calc_grad)), C.byref(h)))
        return Graph._from_handle(h.value)

    def linear_graph(M, N, calc_grad=True):
        h = C.c_void_p()
        check(lib.gtnx_linear_graph(int(M), int(N), int(bool(calc_grad)), C.byref(h)))
        return Graph._from_handle(h.value)

    def linear_graph_n(B, M, N, device_weights, calc_grad=True):
        """B linear graphs with weights copied from one device tensor [B, M, N]"""
        out = (C.c_void_p * B)()
        check(lib.gtnx_linear_graph_n(B, M, N, int(bool(calc_grad)),
                                      _as_dev_ptr(device_weights), out))
        return _wrap_many(out, B)

    ns.scalar_graph = scalar_graph
    ns.linear_graph = linear_graph
    ns.linear_graph_n = linear_graph_n

    def equal(a, b):
        v = C.c_int()
        check(lib.gtnx_equal(a._h, b._h, C.byref(v)))
        return bool(v.value)

    def isomorphic(a, b):
        v = C.c_int()
        check(lib.gtnx_isomorphic(a._h, b._h, C.byref(v)))
        return bool(v.value)

    ns.equal = equal
    ns.isomorphic = isomorphic

    def items(graphs):
        """item() of many scalar graphs with one device->host copy"""
        n = len(graphs)
        out = np.empty(n, dtype=np.float32)
        if n:
            check(lib.gtnx_items_n(_harr(graphs), n, out.ctypes.data))
        return out

    def items_to_device(graphs, device_out):
        check(lib.gtnx_items_device_n(_harr(graphs), len(graphs), _as_dev_ptr(device_out)))

    def grads_to_device(graphs, device_out, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        check(lib.gtnx_grads_device_n(_harr(graphs), len(graphs), _as_dev_ptr(device_out),
                                      off.ctypes.data))

    ns.items = items
    ns.items_to_device = items_to_device
    ns.grads_to_device = grads_to_device

    # ---------------------------------------------------------------- batch records (gtnx_batch_*)
    class Batch:
        """B graphs held as one record: what the list forms of the functions return when
        nobody looks at the elements (one object and one tape node per call).  `b[i]` is
        element i as an ordinary Graph.  The module-level functions (intersect, forward_score,
        subtract, backward, ...) accept Batch arguments."""

        __slots__ = ("_h", "_keep", "__weakref__")  # (_keep: a caller's tensor the record borrows, subtract_into)

        def __init__(self, graphs=None):
            self._h = None
            if graphs is not None:
                graphs = list(graphs)
                h = C.c_void_p()
                check(lib.gtnx_batch_from_graphs(_harr(graphs), len(graphs), C.byref(h)))
                self._h = h.value

        @classmethod
        def _from_handle(cls, h):
            b = cls.__new__(cls)
            b._h = h
            return b

        @classmethod
        def ctc_targets(cls, targets, blank=0, calc_grad=True):
            """CTC target acceptors (benchmarks/ctc.cpp:40-58) of label sequences, built on the device"""
            lens = np.asarray([len(t) for t in targets], dtype=np.int32)
            flat = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in targets])
                                        if len(targets) else np.zeros(0, np.int32), dtype=np.int32)
            h = C.c_void_p()
            check(lib.gtnx_batch_ctc_targets(flat.ctypes.data, lens.ctypes.data, len(lens), int(blank),
                                             int(bool(calc_grad)), C.byref(h)))
            return cls._from_handle(h.value)

        @classmethod
        def asg_force_align(cls, targets, transitions, n_labels):
            """compose(forceAlign(target), transitions) of examples/asg.cpp:50-68 for every label sequence, built on
            the device; `transitions`: a Graph in the arc layout of examples/asg.cpp:36-47 over n_labels labels"""
            lens = np.asarray([len(t) for t in targets], dtype=np.int32)
            flat = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in targets])
                                        if len(targets) else np.zeros(0, np.int32), dtype=np.int32)
            h = C.c_void_p()
            check(lib.gtnx_batch_asg_force_align(flat.ctypes.data, lens.ctypes.data, len(lens), transitions._h,
                                                 int(n_labels), C.byref(h)))
            return cls._from_handle(h.value)

        @classmethod
        def linear(cls, B, M, N, device_weights, calc_grad=True, borrow=False):
            """B linear graphs over one device tensor [B, M, N]; borrow: read in place"""
            h = C.c_void_p()
            check(lib.gtnx_batch_linear(int(B), int(M), int(N), int(bool(calc_grad)), _as_dev_ptr(device_weights),
                                        int(bool(borrow)), C.byref(h)))
            return cls._from_handle(h.value)

        def __del__(self):
            h, self._h = getattr(self, "_h", None), None
            if h:
                try:
                    lib.gtnx_batch_destroy(h)
                except Exception:
                    pass

        def __len__(self):
            v = C.c_int()
            check(lib.gtnx_batch_size(self._h, C.byref(v)))
            return v.value

        def __getitem__(self, i):
            h = C.c_void_p()
            check(lib.gtnx_batch_get(self._h, int(i), C.byref(h)))
            return Graph._from_handle(h.value)

        def items(self):
            out = np.empty(len(self), dtype=np.float32)
            if len(out):
                check(lib.gtnx_batch_items(self._h, out.ctypes.data))
            return out

        def items_to_device(self, device_out):
            check(lib.gtnx_batch_items_device(self._h, _as_dev_ptr(device_out)))

        def bind_grads(self, device_out, offsets):
            off = np.ascontiguousarray(offsets, dtype=np.int64)
            check(lib.gtnx_batch_grads_bind_device(self._h, _as_dev_ptr(device_out), off.ctypes.data))

        def grads_to_device(self, device_out, offsets):
            off = np.ascontiguousarray(offsets, dtype=np.int64)
            check(lib.gtnx_batch_grads_device(self._h, _as_dev_ptr(device_out), off.ctypes.data))

    ns.Batch = Batch

    def _batch_fn(cfn, *args):
        h = C.c_void_p()
        check(cfn(*[a._h for a in args], C.byref(h)))
        return Batch._from_handle(h.value)

    def _with_batches(plain, cfn):
        def f(*args, **kw):
            if args and all(isinstance(a, Batch) for a in args) and not kw:
                return _batch_fn(cfn, *args)
            return plain(*args, **kw)
        f.__doc__ = plain.__doc__
        f.__name__ = getattr(plain, "__name__", "f")
        return f

    if hasattr(lib, "gtnx_batch_negate"):  # (not in the reference-backed shim the CPU tests load)
        for _name in ("negate", "add", "subtract", "compose", "intersect", "forward_score", "viterbi_score",
                      "viterbi_path"):
            setattr(ns, _name, _with_batches(getattr(ns, _name), getattr(lib, "gtnx_batch_" + _name)))

    def subtract_into(a, b, items_device):
        """subtract(a, b) of two batches with the values written straight into `items_device` (a torch CUDA tensor or
        a device address, borrowed: it must outlive the result) -- gtnx_batch_subtract_into"""
        h = C.c_void_p()
        check(lib.gtnx_batch_subtract_into(a._h, b._h, _as_dev_ptr(items_device), C.byref(h)))
        r = Batch._from_handle(h.value)
        r._keep = items_device
        return r

    if hasattr(lib, "gtnx_batch_subtract_into"):
        ns.subtract_into = subtract_into

    _plain_backward = ns.backward

    def backward(g, grad_or_retain=None, retain_graph=False):
        if isinstance(g, Batch):
            r = grad_or_retain if isinstance(grad_or_retain, bool) else retain_graph
            check(lib.gtnx_batch_backward(g._h, int(bool(r))))
            return
        return _plain_backward(g, grad_or_retain, retain_graph)

    ns.backward = backward

    def parallel_for(fn, iterable):
        """gtn.parallel_for (_parallel.cpp:20-26).  The device engine batches
        through list arguments instead; this runs the callable serially."""
        for i in iterable:
            fn(i)

    ns.parallel_for = parallel_for

    # ---------------------------------------------------------------- runtime
    def backend():
        return lib.gtnx_backend().decode()

    def device_count():
        return lib.gtnx_device_count()

    def synchronize():
        check(lib.gtnx_synchronize())

    def set_device(i):
        check(lib.gtnx_set_device(int(i)))

    def set_stream(stream):
        """stream: hipStream_t address (torch.cuda.Stream.cuda_stream) or None"""
        check(lib.gtnx_set_stream(int(stream) if stream else None))

    def compose_mode(mode):
        """0: build compositions; 1: keep chain compositions symbolic whenever eligible; 2: when the
        per-utterance sweep kernels apply; -1 (the default): as 2 for partners built on the host, else 0.
        Returns the previous mode (gtn_amd.h)."""
        prev = C.c_int()
        check(lib.gtnx_compose_mode(int(mode), C.byref(prev)))
        return prev.value

    def memory_stats():
        r, u = C.c_uint64(), C.c_uint64()
        check(lib.gtnx_memory_stats(C.byref(r), C.byref(u)))
        return {"reserved": r.value, "in_use": u.value}

    def prof_enable(on=True):
        check(lib.gtnx_prof_enable(int(bool(on))))

    def prof_reset():
        check(lib.gtnx_prof_reset())

    def prof_get(name):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        check(lib.gtnx_prof_get(name.encode(), C.byref(ms), C.byref(n), C.byref(b)))
        return {"total_ms": ms.value, "launches": n.value, "algorithmic_bytes": b.value}

    def prof_names():
        buf = C.create_string_buffer(4096)
        check(lib.gtnx_prof_names(buf, 4096))
        return [s for s in buf.value.decode().split("\n") if s]

    def debug_symbolic_route(g, tropical=False):
        """which kernel family scores this symbolic product (gtn_amd/csrc/ops_symbolic.cpp); None if `g` is built"""
        r = C.c_int(-1)
        check(lib.gtnx_debug_symbolic_route(g._h, int(bool(tropical)), C.byref(r)))
        if r.value < 0:
            return None
        buf = C.create_string_buffer(32)
        check(lib.gtnx_debug_route_name(r.value, buf, 32))
        return buf.value.decode()

    def debug_viterbi_ties():
        """(seen, unresolved): best paths of symbolic dense-partner products that ran through an exact tie / of those,
        the ones whose product was too large to rebuild and replay the reference's queue on (include/gtn_amd.h)"""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib.gtnx_debug_viterbi_ties(C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    ns.debug_symbolic_route = debug_symbolic_route
    ns.debug_viterbi_ties = debug_viterbi_ties

    def debug_tie_ranks(g):
        """(queue_rank, creation_rank) of a CTC-shaped target's nodes, or None (gtn_amd.h: gtnx_debug_tie_ranks)"""
        n = g.num_nodes()
        k, c = np.zeros(n, np.int32), np.zeros(n, np.int32)
        ok = C.c_int()
        check(lib.gtnx_debug_tie_ranks(g._h, k.ctypes.data, c.ctypes.data, C.byref(ok)))
        return (k.tolist(), c.tolist()) if ok.value else None

    ns.debug_tie_ranks = debug_tie_ranks
    ns.backend = backend
    ns.device_count = device_count
    ns.synchronize = synchronize
    ns.set_device = set_device
    ns.set_stream = set_stream
    ns.compose_mode = compose_mode
    ns.memory_stats = memory_stats
    ns.empty_cache = lambda: check(lib.gtnx_empty_cache())
    ns.prof_enable = prof_enable
    ns.prof_reset = prof_reset
    ns.prof_get = prof_get
    ns.prof_names = prof_names
    return ns


def load_txt(api, text):
    """gtn.loadTxt (gtn/utils.cpp:283-345) for the text fixtures of the tests:
    line 1 start nodes, line 2 accept nodes, then `src dst ilabel [olabel [w]]`."""
    lines = [l for l in text.strip("\n").split("\n")]
    starts = [int(x) for x in lines[0].split()]
    accepts = [int(x) for x in lines[1].split()]
    arcs = []
    nmax = max(starts + accepts + [-1])
    for l in lines[2:]:
        f = l.split()
        if not f:
            continue
        s, d, il = int(f[0]), int(f[1]), int(f[2])
        ol = int(f[3]) if len(f) > 3 else il
        w = float(f[4]) if len(f) > 4 else 0.0
        arcs.append((s, d, il, ol, w))
        nmax = max(nmax, s, d)
    g = api.Graph()
    for n in range(nmax + 1):
        g.add_node(n in starts, n in accepts)
    for s, d, il, ol, w in arcs:
        g.add_arc(s, d, il, ol, w)
    return g
