"""Pins the plain-C oracle (oracle/gtn_oracle.c) against
 (1) the known-answer values of the reference's own tests, and
 (2) tests/golden/golden.json -- outputs of the unmodified reference.
CPU only."""
import math

import numpy as np
import pytest

import graphgen as gg
from oracle_lib import OGraph, ctc_loss

INF = float("inf")


def G(start, accept, arcs):
    N = max([a[0] for a in arcs] + [a[1] for a in arcs] + start + accept + [-1]) + 1
    d = {"start": [int(n in start) for n in range(N)], "accept": [int(n in accept) for n in range(N)],
         "src": [a[0] for a in arcs], "dst": [a[1] for a in arcs], "il": [a[2] for a in arcs],
         "ol": [a[3] for a in arcs], "w": [float(a[4]) for a in arcs], "sort": None}
    return OGraph.from_dict(d)


COMPLEX = ([0, 1], [3, 4], [(0, 1, 0, 0, 2), (0, 2, 1, 1, 1), (1, 2, 0, 0, 2), (2, 3, 0, 0, 1),
                            (2, 3, 1, 1, 1), (1, 4, 0, 0, 2), (2, 4, 1, 1, 3), (3, 4, 0, 0, 2)])
SIMPLE = ([0], [2], [(0, 1, 0, 0, 1), (0, 1, 1, 1, 2), (0, 1, 2, 2, 3), (1, 2, 0, 0, 1),
                     (1, 2, 1, 1, 2), (1, 2, 2, 2, 3)])


# ---- reference test/functions_test.cpp:231-389 ---------------------------------
def test_forward_known_answers():
    assert OGraph().shortest_distance() == -INF                       # :233-236
    assert G([0], [0], [(0, 0, 1, 1, 0)]).shortest_distance() is None  # :238-244 self loop
    assert G([0], [2], [(0, 1, 0, 0, 0), (1, 2, 0, 0, 0), (1, 1, 0, 0, 0)]).shortest_distance() is None
    assert G([0], [2], [(0, 1, 0, 0, 0), (1, 2, 0, 0, 0), (2, 2, 0, 0, 0)]).shortest_distance() is None
    assert G([0], [2], [(0, 1, 0, 0, 0), (1, 2, 0, 0, 0), (2, 0, 0, 0, 0)]).shortest_distance() is None
    assert G([0], [2], [(0, 2, 0, 0, 0), (1, 2, 0, 0, 0)]).shortest_distance() is None  # :281-291
    assert G([0], [1], [(0, 1, 0, 0, -INF), (0, 1, 1, 1, -INF)]).shortest_distance() == -INF
    assert G([0], [1], [(0, 1, 0, 0, INF), (0, 1, 1, 1, 0)]).shortest_distance() == INF
    assert G([0], [0], []).shortest_distance() == 0.0                  # :314-319
    assert G(*SIMPLE).shortest_distance() == pytest.approx(6.8152, rel=1e-4)   # :321-334
    e = math.log(math.exp(1) + math.exp(-5 + 2) + math.exp(2))
    assert G([0, 1], [2], [(0, 1, 0, 0, -5), (0, 2, 0, 0, 1), (1, 2, 0, 0, 2)]).shortest_distance() == pytest.approx(e)
    e = math.log(2 * math.exp(2) + math.exp(4))
    assert G([0], [1, 2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2), (1, 2, 0, 0, 2)]).shortest_distance() == pytest.approx(e)
    assert G([0], [2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2)]).shortest_distance() == 2.0
    assert G(*COMPLEX).shortest_distance() == pytest.approx(8.36931, rel=1e-5)  # :373-388


# ---- test/functions_test.cpp:391-453 ---------------------------------------------
def test_viterbi_known_answers():
    assert OGraph().shortest_distance(True) == -INF
    assert G(*SIMPLE).shortest_distance(True) == 6.0
    assert G([0, 1], [2], [(0, 1, 0, 0, -5), (0, 2, 0, 0, 1), (1, 2, 0, 0, 2)]).shortest_distance(True) == 2.0
    assert G([0], [1, 2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2), (1, 2, 0, 0, 2)]).shortest_distance(True) == 4.0
    assert G(*COMPLEX).shortest_distance(True) == 7.0


# ---- test/autograd_test.cpp:270-480 ------------------------------------------------
def test_grad_known_answers():
    g = G([0, 1], [2], [(0, 1, 0, 0, -5), (0, 2, 0, 0, 1), (1, 2, 0, 0, 2)])
    den = 1 / (math.exp(-3) + math.exp(1) + math.exp(2))
    np.testing.assert_allclose(g.shortest_distance_grad(),
                               [den * math.exp(-3), den * math.exp(1), den * (math.exp(-3) + math.exp(2))], rtol=1e-5)
    g = G([0], [1, 2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2), (1, 2, 0, 0, 2)])
    den = 1 / (2 * math.exp(2) + math.exp(4))
    np.testing.assert_allclose(g.shortest_distance_grad(),
                               [den * (math.exp(2) + math.exp(4)), den * math.exp(2), den * math.exp(4)], rtol=1e-5)
    np.testing.assert_allclose(G([0], [2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2)]).shortest_distance_grad(), [0, 1])
    assert np.isnan(G([0], [1], [(0, 1, 0, 0, -INF), (0, 1, 1, 1, -INF)]).shortest_distance_grad()).all()
    np.testing.assert_allclose(G([0], [1], [(0, 1, 0, 0, -INF), (0, 1, 1, 1, 1.0)]).shortest_distance_grad(), [0, 1])
    assert np.isnan(G([0], [1], [(0, 1, 0, 0, INF), (0, 1, 1, 1, INF)]).shortest_distance_grad()).all()
    assert np.isnan(G([0], [1], [(0, 1, 0, 0, INF), (0, 1, 1, 1, 1.0)]).shortest_distance_grad()).all()
    # viterbi one-hot grads (:407-480)
    assert G(*SIMPLE).shortest_distance_grad(True).tolist() == [0, 0, 1, 0, 0, 1]
    assert G([0, 1], [2], [(0, 1, 0, 0, -5), (0, 2, 0, 0, 1), (1, 2, 0, 0, 2)]).shortest_distance_grad(True).tolist() == [0, 0, 1]
    assert G([0], [1, 2], [(0, 1, 0, 0, 2), (0, 2, 0, 0, 2), (1, 2, 0, 0, 2)]).shortest_distance_grad(True).tolist() == [1, 0, 1]
    assert G(*COMPLEX).shortest_distance_grad(True).tolist() in ([1, 0, 1, 0, 0, 0, 1, 0], [1, 0, 1, 0, 1, 0, 0, 1])


# ---- test/autograd_test.cpp:148-188 (compose grads) ----------------------------------
def test_compose_grad_known_answer():
    first = G([0], [3], [(0, 0, 0, 0, 0), (0, 1, 1, 1, 0), (1, 2, 2, 2, 0), (2, 3, 0, 0, 0)])  # shape only
    # the exact fixture of the reference:
    first = G([0], [4], [(0, 1, 0, 0, 0), (0, 1, 1, 1, 0), (0, 1, 2, 2, 0), (1, 2, 0, 0, 0), (1, 2, 1, 1, 0),
                         (1, 2, 2, 2, 0), (2, 3, 0, 0, 0), (2, 3, 1, 1, 0), (2, 3, 2, 2, 0), (3, 4, 0, 0, 0),
                         (3, 4, 1, 1, 0), (3, 4, 2, 2, 0)])
    second = G([0], [2], [(0, 1, 0, 0, 3.5), (1, 1, 0, 0, 2.5), (1, 2, 1, 1, 1.5), (2, 2, 1, 1, 4.5)])
    comp = first.compose(second)
    g1, g2 = comp.compose_grad(np.ones(comp.A), first.A, second.A)
    assert g1.tolist() == [1, 0, 0, 1, 1, 0, 1, 2, 0, 0, 2, 0]
    assert g2.tolist() == [1, 2, 3, 2]


# ---- test/criterion_test.cpp:56-180 ----------------------------------------------------
def ctc_graph_dict(target, blank):
    d = gg.ctc_target_graph(target, blank)
    d["sort"] = None
    return d


def test_ctc_criterion_known_answers():
    # case 1 (:60-70): log(0) = -inf emissions
    with np.errstate(divide="ignore"):
        em = np.log(np.array([1.0, 0.0, 0.0, 1.0, 1.0, 0.0], np.float32))
    ctc = OGraph.from_dict(ctc_graph_dict([0, 0], 1))
    e = OGraph.linear(3, 2, em)
    assert ctc.compose(e).shortest_distance() == 0.0
    assert e.shortest_distance() == 0.0
    # case 2 (:72-84)
    T, N = 3, 4
    ctc = OGraph.from_dict(ctc_graph_dict([1, 2], N - 1))
    e = OGraph.linear(T, N, np.zeros(T * N, np.float32))
    loss = ctc.compose(e).shortest_distance() - e.shortest_distance()
    assert -loss == pytest.approx(-math.log(0.25 ** 3 * 5), rel=1e-5)
    # case 3 (:88-130) TensorFlow vector
    em = np.array([0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553,
                   0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436,
                   0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688,
                   0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533,
                   0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107], np.float32)
    expected_grad = np.array([-0.366234, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553,
                              0.111121, -0.411608, 0.278779, 0.0055756, 0.00569609, 0.010436,
                              0.0357786, 0.633813, -0.678582, 0.00249248, 0.00272882, 0.0037688,
                              0.0663296, -0.356151, 0.280111, 0.00283995, 0.0035545, 0.00331533,
                              -0.541765, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107], np.float32)
    ctc = OGraph.from_dict(ctc_graph_dict([0, 1, 2, 1, 0], 5))
    e = OGraph.linear(5, 6, np.log(em))
    z = e.shortest_distance()
    assert abs(z) < 1e-5
    comp = ctc.compose(e)
    assert z - comp.shortest_distance() == pytest.approx(3.34211, rel=1e-5)
    gz = e.shortest_distance_grad(delta=1.0)
    gc = comp.shortest_distance_grad(delta=-1.0)
    _, g2 = comp.compose_grad(gc, ctc.A, e.A)
    np.testing.assert_allclose(gz + g2, expected_grad, atol=1e-5)


# ---- golden fixtures from the real reference ------------------------------------------------
def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_golden_shortest(golden):
    for c in golden["shortest"]:
        g = OGraph.from_dict(c["graph"])
        for key, trop in (("forward", False), ("viterbi", True)):
            s = g.shortest_distance(trop)
            if c[key] == "error":
                assert s is None, c["name"]
                continue
            assert close(s, c[key]), (c["name"], key, s, c[key])
            if key + "_grad" in c:
                gr = g.shortest_distance_grad(trop)
                if trop:
                    assert gr.tolist() == c[key + "_grad"], (c["name"], key)
                else:
                    assert close(gr, c[key + "_grad"], 1e-4, 1e-6), (c["name"], key)
        p = g.shortest_path()
        if c["path"] == "error":
            assert p is None
        else:
            arcs, has = p
            d = c["graph"]
            assert [d["il"][a] for a in arcs] == c["path"]["il"], c["name"]
            assert [d["ol"][a] for a in arcs] == c["path"]["ol"], c["name"]
            assert [d["w"][a] for a in arcs] == c["path"]["w"], c["name"]
            assert len(c["path"]["start"]) == (len(arcs) + 1 if has else 0)
            pg = np.zeros(len(d["src"]), np.float32)
            for a in arcs:
                pg[a] += 1
            assert pg.tolist() == c.get("path_grad", pg.tolist()), c["name"]


def sorted_arcs_by_src(d):
    """reference equal(): same node ids, per-node multiset of out arcs"""
    return sorted(zip(d["src"], d["dst"], d["il"], d["ol"], d["w"]))


def test_golden_compose(golden):
    n_exact = 0
    for c in golden["compose"]:
        g1, g2 = OGraph.from_dict(c["g1"]), OGraph.from_dict(c["g2"])
        out = g1.compose(g2, c["mode"])
        d, e = out.to_dict(), c["out"]
        assert d["start"] == e["start"] and d["accept"] == e["accept"], c["name"]
        # node numbering must be the reference's; arc ORDER may differ only among
        # equal labels (std::sort instability), so compare as equal() does ...
        assert sorted_arcs_by_src(d) == sorted_arcs_by_src(e), c["name"]
        # ... and count how often even the arc order is identical
        same_order = (d["src"], d["dst"], d["il"], d["ol"]) == (e["src"], e["dst"], e["il"], e["ol"])
        n_exact += same_order
        if "grad1" in c and same_order:
            a, b = out.compose_grad(np.ones(out.A), g1.A, g2.A)
            assert a.tolist() == c["grad1"] and b.tolist() == c["grad2"], c["name"]
        s = out.shortest_distance()
        if c["forward"] == "error":
            assert s is None, c["name"]
        else:
            assert close(s, c["forward"], 1e-5), c["name"]
            if "fgrad1" in c:
                gc = out.shortest_distance_grad()
                a, b = out.compose_grad(gc, g1.A, g2.A)
                assert close(a, c["fgrad1"], 1e-4, 1e-6) and close(b, c["fgrad2"], 1e-4, 1e-6), c["name"]
    assert n_exact >= 0.9 * len(golden["compose"])


def test_golden_ctc(golden):
    for c in golden["ctc"]:
        em = np.array(c["emissions"], np.float32).reshape(c["T"], c["C"])
        loss, grad = ctc_loss(em, c["target"])
        assert loss == pytest.approx(c["loss"], rel=1e-5), c["name"]
        np.testing.assert_allclose(grad.reshape(-1), c["grad"], rtol=1e-4, atol=1e-6)


# ---- reference test/criterion_test.cpp:182-345: the ASG criterion --------------------------------
def _asg_transitions(N, weights=None):
    """examples/asg.cpp:36-47 / criterion_test.cpp:262-272: arc i: <s> -> i; arc N + i*N + j: j -> i"""
    arcs = [(0, i + 1, i, i, 0.0) for i in range(N)]
    for i in range(N):
        for j in range(N):
            arcs.append((j + 1, i + 1, i, i, 0.0 if weights is None else float(weights[i * N + j])))
    return G([0], list(range(1, N + 1)), arcs)


def _force_align(target):
    arcs = []
    for l in range(1, len(target) + 1):
        arcs.append((l - 1, l, target[l - 1], target[l - 1], 0.0))
        arcs.append((l, l, target[l - 1], target[l - 1], 0.0))
    return G([0], [len(target)], arcs)


def asg_utterance_grads(T, N, emissions, target, trans_w=None):
    """-> (loss, d loss / d emissions, d loss / d transitions) of one utterance through the oracle"""
    trans = _asg_transitions(N) if trans_w is None else trans_w
    e = OGraph.linear(T, N, np.asarray(emissions, np.float32))
    fal = _force_align(target)
    fcc = e.compose(trans)
    ft = fal.compose(trans)
    falc = ft.compose(e)
    loss = fcc.shortest_distance() - falc.shortest_distance()
    ge1, gt1 = fcc.compose_grad(fcc.shortest_distance_grad(delta=1.0), e.A, trans.A)
    gft, ge2 = falc.compose_grad(falc.shortest_distance_grad(delta=-1.0), ft.A, e.A)
    _, gt2 = ft.compose_grad(gft, fal.A, trans.A)
    return loss, ge1 + ge2, gt1.astype(np.float64) + gt2


def test_asg_criterion_known_answers():
    """losses, emission gradients and the transition gradient summed over the three utterances
    (wav2letter's vectors, criterion_test.cpp:186-305) through compose / shortest distance /
    their gradients of the oracle"""
    from test_parity_gpu import ASG_EMISSIONS, ASG_EM_GRADS, ASG_TRANS_GRAD
    T, N = 5, 6
    targets = [[2, 1, 5, 1, 3], [4, 3, 5], [3, 2, 2, 1]]
    expected_loss = [7.7417464256287, 6.4200420379639, 8.2780694961548]
    trans = _asg_transitions(N)
    gtrans = np.zeros(trans.A, np.float64)
    for b in range(3):
        e = OGraph.linear(T, N, np.asarray(ASG_EMISSIONS[b], np.float32))
        fal = _force_align(targets[b])
        fcc = e.compose(trans)                      # compose(emissions, transitions)
        ft = fal.compose(trans)
        falc = ft.compose(e)                        # compose(compose(fal, transitions), emissions)
        loss = fcc.shortest_distance() - falc.shortest_distance()
        assert loss == pytest.approx(expected_loss[b], abs=1e-3)
        # d loss: +1 through the full-connect term, -1 through the force-align term
        ge1, gt1 = fcc.compose_grad(fcc.shortest_distance_grad(delta=1.0), e.A, trans.A)
        gft, ge2 = falc.compose_grad(falc.shortest_distance_grad(delta=-1.0), ft.A, e.A)
        _, gt2 = ft.compose_grad(gft, fal.A, trans.A)
        np.testing.assert_allclose(ge1 + ge2, ASG_EM_GRADS[b], atol=1e-4)
        gtrans += gt1.astype(np.float64) + gt2
    np.testing.assert_allclose(gtrans[N:], ASG_TRANS_GRAD, atol=1e-4)


def test_asg_viterbi_path_known_answer():
    """criterion_test.cpp:308-345"""
    T, N = 4, 3
    em = np.array([0, 0, 7, 5, 4, 3, 5, 8, 5, 5, 4, 3], np.float32)
    tw = [0, 2, 0, 0, 0, 2, 2, 0, 0]
    comp = OGraph.linear(T, N, em).compose(_asg_transitions(N, tw))
    arcs, has = comp.shortest_path()
    assert has
    d = comp.to_dict()
    assert [d["ol"][a] for a in arcs] == [2, 1, 1, 0]


def test_golden_asg(golden):
    """tests/golden/golden.json "asg": the unmodified reference on random ASG instances
    (losses, both gradients, Viterbi decode)"""
    for c in golden["asg"]:
        T, N = c["T"], c["N"]
        tw = np.asarray(c["transitions"], np.float32)
        trans = _asg_transitions(N, tw[N:])
        d = trans.to_dict()
        d["w"][:N] = [float(x) for x in tw[:N]]
        d["sort"] = None
        trans = OGraph.from_dict(d)
        e = OGraph.linear(T, N, np.asarray(c["emissions"], np.float32))
        fal = _force_align(c["target"]) if c["target"] else G([0], [0], [])
        fcc = e.compose(trans)
        ft = fal.compose(trans)
        falc = ft.compose(e)
        assert fcc.shortest_distance() - falc.shortest_distance() == pytest.approx(c["loss"], rel=1e-5), c["name"]
        ge1, gt1 = fcc.compose_grad(fcc.shortest_distance_grad(delta=1.0), e.A, trans.A)
        gft, ge2 = falc.compose_grad(falc.shortest_distance_grad(delta=-1.0), ft.A, e.A)
        _, gt2 = ft.compose_grad(gft, fal.A, trans.A)
        np.testing.assert_allclose(ge1 + ge2, c["grad_emissions"], rtol=1e-4, atol=1e-5, err_msg=c["name"])
        np.testing.assert_allclose(gt1 + gt2, c["grad_transitions"], rtol=1e-4, atol=1e-5, err_msg=c["name"])
        arcs, has = fcc.shortest_path()
        dd = fcc.to_dict()
        assert has and [dd["ol"][a] for a in arcs] == c["viterbi_labels"], c["name"]
