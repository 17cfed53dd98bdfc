"""Seeded synthetic graph generators shared by the golden-fixture script and
the parity tests.  A graph is a plain dict of lists:

  {"start": [0/1]*N, "accept": [0/1]*N, "src": [...], "dst": [...],
   "il": [...], "ol": [...], "w": [...], "sort": None | "i" | "o"}
"""
import numpy as np

EPS = -1


def to_api(api, d, calc_grad=True):
    g = api.Graph(calc_grad)
    if len(d["start"]):
        g.add_nodes(d["start"], d["accept"])
    if len(d["src"]):
        g.add_arcs(d["src"], d["dst"], d["il"], d["ol"], d["w"])
    if d.get("sort") == "i":
        g.arc_sort(False)
    elif d.get("sort") == "o":
        g.arc_sort(True)
    return g


def from_api(g):
    s, d, il, ol, w = g.arcs()
    N = g.num_nodes()
    start = [0] * N
    accept = [0] * N
    for n in g.start():
        start[n] = 1
    for n in g.accept():
        accept[n] = 1
    return {
        "start": start, "accept": accept, "src": s.tolist(), "dst": d.tolist(),
        "il": il.tolist(), "ol": ol.tolist(), "w": [float(x) for x in w], "sort": None,
    }


def _f32(x):
    return [float(np.float32(v)) for v in x]


def random_dag(rng, N, avg_deg=2.5, nlabels=4, n_start=1, n_accept=1, orphan_ok=False,
               wscale=2.0):
    """topologically indexed DAG; every non-start node gets >= 1 incoming arc
    unless orphan_ok."""
    start = [0] * N
    accept = [0] * N
    for n in range(min(n_start, N)):
        start[n] = 1
    for n in range(max(0, N - n_accept), N):
        accept[n] = 1
    src, dst = [], []
    for n in range(1, N):
        if start[n] and rng.random() < 0.5:
            continue
        if orphan_ok and rng.random() < 0.1:
            continue
        k = 1 + rng.poisson(max(avg_deg - 1, 0))
        for _ in range(k):
            src.append(int(rng.integers(0, n)))
            dst.append(n)
    perm = rng.permutation(len(src))
    src = [src[i] for i in perm]
    dst = [dst[i] for i in perm]
    A = len(src)
    il = rng.integers(0, nlabels, A).tolist()
    ol = rng.integers(0, nlabels, A).tolist()
    w = _f32(rng.normal(0, wscale, A))
    return {"start": start, "accept": accept, "src": src, "dst": dst, "il": il, "ol": ol,
            "w": w, "sort": None}


def random_graph(rng, N, A, nlabels=3, eps_prob=0.0, acceptor=False, p_start=0.3, p_accept=0.3):
    """arbitrary (possibly cyclic) graph for composition tests"""
    start = (rng.random(N) < p_start).astype(int).tolist()
    accept = (rng.random(N) < p_accept).astype(int).tolist()
    if N and not any(start):
        start[0] = 1
    if N and not any(accept):
        accept[N - 1] = 1
    src = rng.integers(0, N, A).tolist() if N else []
    dst = rng.integers(0, N, A).tolist() if N else []
    il = rng.integers(0, nlabels, A)
    ol = il.copy() if acceptor else rng.integers(0, nlabels, A)
    if eps_prob > 0:
        il = np.where(rng.random(A) < eps_prob, EPS, il)
        ol = np.where(rng.random(A) < eps_prob, EPS, ol)
        if acceptor:
            ol = il.copy()
    w = _f32(rng.normal(0, 1.0, A))
    return {"start": start, "accept": accept, "src": src, "dst": dst, "il": il.tolist(),
            "ol": ol.tolist(), "w": w, "sort": None}


def ctc_target_graph(target, blank=0):
    """benchmarks/ctc.cpp:40-58 (node/arc order preserved)"""
    L = 2 * len(target) + 1
    start, accept, src, dst, lab = [], [], [], [], []
    for l in range(L):
        idx = (l - 1) // 2
        start.append(int(l == 0))
        accept.append(int(l == L - 1 or l == L - 2))
        label = target[idx] if l % 2 else blank
        src.append(l); dst.append(l); lab.append(label)
        if l > 0:
            src.append(l - 1); dst.append(l); lab.append(label)
        if l % 2 and l > 1 and label != target[idx - 1]:
            src.append(l - 2); dst.append(l); lab.append(label)
    return {"start": start, "accept": accept, "src": src, "dst": dst, "il": lab, "ol": list(lab),
            "w": [0.0] * len(src), "sort": "i"}


def linear(T, C, weights):
    """gtn/creations.cpp:20-33 as an explicit dict (arc id = t*C + c)"""
    start = [1] + [0] * T
    accept = [0] * T + [1] if T > 0 else [0]
    src = np.repeat(np.arange(T), C).tolist()
    dst = (np.repeat(np.arange(T), C) + 1).tolist()
    lab = np.tile(np.arange(C), T).tolist()
    return {"start": start, "accept": accept, "src": src, "dst": dst, "il": lab, "ol": list(lab),
            "w": _f32(np.asarray(weights).reshape(-1)), "sort": "both"}


def ctc_inputs(seed, B, T, C, U):
    """benchmarks/ctc.cpp:16-38: emissions uniform [-5,5), targets uniform in
    [1, C-1]; seeded instead of std::rand."""
    rng = np.random.default_rng(seed)
    em = (rng.random((B, T, C), dtype=np.float32) * 10 - 5).astype(np.float32)
    tg = rng.integers(1, C, size=(B, U)).astype(np.int32)
    return em, tg
