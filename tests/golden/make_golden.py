#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the UNMODIFIED reference
(oracle/_ref/libgtn_ref.so = /root/reference sources + oracle/ref_shim.cpp)
on seeded synthetic inputs.  Run in the build container only (needs
/root/reference to build the shim):

    make -C oracle && python tests/golden/make_golden.py

The fixtures pin (a) the C oracle (tests/test_oracle.py, CPU) and (b) the HIP
engine (tests/test_parity_gpu.py, GPU).
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "refbackend"))

import gtn_ref as ref  # noqa: E402  (the Python mirror bound to the reference shim: tests/refbackend/gtn_ref.py)
import graphgen as gg  # noqa: E402

assert ref.backend() == "reference-cpu"
INF = float("inf")


def fl(x):
    return [float(v) for v in np.asarray(x, dtype=np.float32).reshape(-1)]


def shortest_case(name, d):
    case = {"name": name, "graph": d}
    g = gg.to_api(ref, d)
    for key, fn in (("forward", ref.forward_score), ("viterbi", ref.viterbi_score)):
        try:
            s = fn(g)
            case[key] = s.item()
            g.zero_grad()
            ref.backward(s)
            case[key + "_grad"] = fl(g.grad().weights_to_numpy()) if g.num_arcs() else []
        except ValueError as e:
            case[key] = "error"
    try:
        g.zero_grad()
        p = ref.viterbi_path(g)
        case["path"] = gg.from_api(p)
        if p.num_arcs() > 0 or True:
            ref.backward(p)
            case["path_grad"] = fl(g.grad().weights_to_numpy()) if g.num_arcs() else []
    except ValueError:
        case["path"] = "error"
    return case


def compose_case(name, d1, d2, mode):
    g1, g2 = gg.to_api(ref, d1), gg.to_api(ref, d2)
    fn = ref.compose if mode == "compose" else ref.intersect
    out = fn(g1, g2)
    case = {"name": name, "g1": d1, "g2": d2, "mode": mode, "out": gg.from_api(out)}
    if out.num_arcs() > 0:
        ref.backward(out)
        case["grad1"] = fl(g1.grad().weights_to_numpy())
        case["grad2"] = fl(g2.grad().weights_to_numpy())
    # chain: forwardScore(compose) when it is a DAG
    g1.zero_grad()
    g2.zero_grad()
    try:
        s = ref.forward_score(fn(g1, g2))
        case["forward"] = s.item()
        if math.isfinite(s.item()):
            ref.backward(s)
            case["fgrad1"] = fl(g1.grad().weights_to_numpy())
            case["fgrad2"] = fl(g2.grad().weights_to_numpy())
    except ValueError:
        case["forward"] = "error"
    return case


def ctc_case(name, seed, T, C, U):
    em, tg = gg.ctc_inputs(seed, 1, T, C, U)
    em, tg = em[0], tg[0]
    ctc = gg.to_api(ref, gg.ctc_target_graph(tg.tolist()))
    e = ref.linear_graph(T, C)
    e.set_weights(em)
    comp = ref.intersect(ctc, e)
    loss = ref.subtract(ref.forward_score(e), ref.forward_score(comp))
    ref.backward(loss)
    vit = ref.viterbi_path(ref.intersect(ctc, e))
    return {"name": name, "seed": seed, "T": T, "C": C, "U": U, "emissions": fl(em),
            "target": tg.tolist(), "loss": loss.item(), "grad": fl(e.grad().weights_to_numpy()),
            "comp_nodes": comp.num_nodes(), "comp_arcs": comp.num_arcs(),
            "viterbi_labels": vit.labels_to_list()}


def asg_case(name, seed, T, N, target):
    """examples/asg.cpp:30-68 with random emissions / transition / start scores: loss, emission gradient,
    transition gradient (arc order of the transitions graph: N start arcs, then arc N + i*N + j = j -> i)
    and the Viterbi decode of compose(emissions, transitions)"""
    rng = np.random.default_rng(seed)
    em = rng.normal(0, 1, (T, N)).astype(np.float32)
    tw = rng.normal(0, 1, N + N * N).astype(np.float32)
    trans = ref.Graph()
    trans.add_node(True)
    for i in range(N):
        trans.add_node(False, True)
        trans.add_arc(0, i + 1, i, i, float(tw[i]))
    for i in range(N):
        for j in range(N):
            trans.add_arc(j + 1, i + 1, i, i, float(tw[N + i * N + j]))
    fal = ref.Graph()
    fal.add_node(True, len(target) == 0)
    for l in range(1, len(target) + 1):
        fal.add_node(False, l == len(target))
        fal.add_arc(l - 1, l, target[l - 1])
        fal.add_arc(l, l, target[l - 1])
    e = ref.linear_graph(T, N)
    e.set_weights(em)
    loss = ref.subtract(ref.forward_score(ref.compose(e, trans)),
                        ref.forward_score(ref.compose(ref.compose(fal, trans), e)))
    ref.backward(loss)
    vit = ref.viterbi_path(ref.compose(e, trans))
    return {"name": name, "seed": seed, "T": T, "N": N, "target": list(target), "emissions": fl(em),
            "transitions": fl(tw), "loss": loss.item(), "grad_emissions": fl(e.grad().weights_to_numpy()),
            "grad_transitions": fl(trans.grad().weights_to_numpy()), "viterbi_labels": vit.labels_to_list()}


def main():
    rng = np.random.default_rng(20240925)
    out = {"shortest": [], "compose": [], "ctc": [], "asg": []}

    # ---- shortest distance / path
    empty = {"start": [], "accept": [], "src": [], "dst": [], "il": [], "ol": [], "w": [], "sort": None}
    out["shortest"].append(shortest_case("empty", empty))
    single = dict(empty, start=[1], accept=[1])
    out["shortest"].append(shortest_case("single", single))
    k = 0
    for N in (3, 5, 8, 13, 21, 34, 60, 120, 200):
        for rep in range(4):
            d = gg.random_dag(rng, N, avg_deg=1.5 + rep, nlabels=4,
                              n_start=1 + rep % 3, n_accept=1 + (rep + 1) % 3,
                              orphan_ok=(rep == 3))
            if rep == 2 and len(d["w"]) > 2:  # +-inf weights
                d["w"][0] = -INF
                d["w"][len(d["w"]) // 2] = -INF
            if rep == 1:
                d["sort"] = "i"
            out["shortest"].append(shortest_case(f"dag{k}_N{N}", d))
            k += 1
    # integer weights => ties in the tropical semiring
    for rep in range(6):
        d = gg.random_dag(rng, 10 + 5 * rep, avg_deg=3, nlabels=3)
        d["w"] = [float(int(x)) for x in rng.integers(-2, 3, len(d["w"]))]
        out["shortest"].append(shortest_case(f"ties{rep}", d))

    # ---- composition
    k = 0
    for eps in (0.0, 0.35):
        for rep in range(36):
            N1, N2 = int(rng.integers(1, 8)), int(rng.integers(1, 8))
            A1, A2 = int(rng.integers(0, 22)), int(rng.integers(0, 22))
            mode = "intersect" if rep % 3 == 2 else "compose"
            acc = mode == "intersect"
            d1 = gg.random_graph(rng, N1, A1, nlabels=3, eps_prob=eps, acceptor=acc)
            d2 = gg.random_graph(rng, N2, A2, nlabels=3, eps_prob=eps, acceptor=acc)
            sv = rep % 4
            d1["sort"] = "o" if sv in (1, 3) else None
            d2["sort"] = "i" if sv in (2, 3) else None
            out["compose"].append(compose_case(f"rand{k}_eps{eps}", d1, d2, mode))
            k += 1
    # DAG x DAG (acyclic products, exercised by forwardScore + backward)
    for rep in range(12):
        d1 = gg.random_dag(rng, int(rng.integers(3, 14)), avg_deg=2.5, nlabels=3)
        d2 = gg.random_dag(rng, int(rng.integers(3, 14)), avg_deg=2.5, nlabels=3)
        d1["sort"] = "o" if rep % 2 else None
        d2["sort"] = "i" if rep % 3 == 0 else None
        out["compose"].append(compose_case(f"dag{rep}", d1, d2, "compose"))
    # acceptor x linear chain (the CTC / ASG shapes)
    for rep in range(6):
        T, C = int(rng.integers(2, 9)), int(rng.integers(2, 5))
        d1 = gg.random_graph(rng, int(rng.integers(2, 6)), int(rng.integers(3, 14)),
                             nlabels=C, acceptor=True)
        d1["sort"] = "i" if rep % 2 else None
        d2 = gg.linear(T, C, rng.normal(0, 1, T * C))
        d2["sort"] = "i"
        out["compose"].append(compose_case(f"lin{rep}", d1, d2, "intersect"))
        out["compose"].append(compose_case(f"linrev{rep}", d2, d1, "compose"))

    # ---- CTC
    for i, (T, C, U) in enumerate([(5, 4, 2), (12, 5, 4), (20, 6, 5), (50, 10, 8), (100, 28, 20)]):
        out["ctc"].append(ctc_case(f"ctc_T{T}_C{C}_U{U}", 1234 + i, T, C, U))

    # ---- ASG (own seeds: the sections above keep their random stream)
    for i, (T, N, target) in enumerate([(4, 3, [1]), (6, 4, [2, 2, 0]), (12, 6, [5, 1, 1, 3]), (25, 9, [0, 8, 3, 3, 4, 1, 7]),
                                        (40, 28, [3, 20, 20, 7, 7, 7, 1, 27, 0, 12])]):
        out["asg"].append(asg_case(f"asg_T{T}_N{N}_U{len(target)}", 777 + i, T, N, target))

    path = os.path.join(HERE, "golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;",
          {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
