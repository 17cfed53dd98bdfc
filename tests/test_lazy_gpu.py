"""Lazy (never materialised) chain products -- lazy.hip -- against the materialised
path (itself pinned to the reference/oracle by test_parity_gpu.py) and the oracle:
scores within 1e-4 relative, best-path labels exact, gradients at the tolerance
test_parity_gpu.py documents."""
import os
import sys

import numpy as np
import pytest

import graphgen as gg
from oracle_lib import OGraph, ctc_loss
from test_parity_gpu import asg_transitions

pytestmark = pytest.mark.gpu
RTOL = 1e-4


class lazy_mode:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.get("GTNX_LAZY_COMPOSE")
        os.environ["GTNX_LAZY_COMPOSE"] = self.v

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("GTNX_LAZY_COMPOSE", None)
        else:
            os.environ["GTNX_LAZY_COMPOSE"] = self.old


def asg_batch(gtn, B, T, N, seed):
    rng = np.random.default_rng(seed)
    em = rng.normal(0, 1, (B, T, N)).astype(np.float32)
    tw = rng.normal(0, 1, N * N + N).astype(np.float32)
    trans = asg_transitions(gtn, N, tw[N:])
    w = trans.weights_to_numpy()
    w[:N] = tw[:N]
    trans.set_weights(w)
    ems = []
    for b in range(B):
        e = gtn.linear_graph(T, N)
        e.set_weights(em[b])
        ems.append(e)
    return em, ems, trans


@pytest.mark.parametrize("B,T,N", [(1, 7, 4), (5, 30, 12), (19, 40, 35)])
def test_lazy_asg_forward_viterbi_grads_match_materialised(gtn, B, T, N):
    res = {}
    for mode in ("0", "1"):
        with lazy_mode(mode):
            em, ems, trans = asg_batch(gtn, B, T, N, 11)
            comp = gtn.compose(ems, [trans])
            fs = gtn.forward_score(comp)
            vs = gtn.viterbi_score(comp)
            paths = gtn.viterbi_path(comp)
            gtn.backward(fs)
            res[mode] = dict(
                fs=gtn.items(fs), vs=gtn.items(vs), labels=[p.labels_to_list() for p in paths],
                pw=[p.weights_to_numpy() for p in paths],
                ge=[e.grad().weights_to_numpy() for e in ems], gt=trans.grad().weights_to_numpy())
    a, b = res["0"], res["1"]
    np.testing.assert_allclose(b["fs"], a["fs"], rtol=RTOL)
    np.testing.assert_allclose(b["vs"], a["vs"], rtol=1e-6)
    assert b["labels"] == a["labels"]
    for x, y in zip(b["pw"], a["pw"]):
        np.testing.assert_allclose(x, y, rtol=1e-6)
    for x, y in zip(b["ge"], a["ge"]):
        np.testing.assert_allclose(x, y, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(b["gt"], a["gt"], rtol=1e-3, atol=1e-4)


def test_lazy_viterbi_grads_match_materialised(gtn):
    B, T, N = 4, 12, 6
    res = {}
    for mode in ("0", "1"):
        with lazy_mode(mode):
            em, ems, trans = asg_batch(gtn, B, T, N, 5)
            comp = gtn.compose(ems, [trans])
            vs = gtn.viterbi_score(comp)
            gtn.backward(vs)
            g1 = [e.grad().weights_to_numpy() for e in ems]
            gt1 = trans.grad().weights_to_numpy()
            em, ems2, trans2 = asg_batch(gtn, B, T, N, 5)
            paths = gtn.viterbi_path(gtn.compose(ems2, [trans2]))
            gtn.backward(paths)
            res[mode] = (g1, gt1, [e.grad().weights_to_numpy() for e in ems2], trans2.grad().weights_to_numpy())
    for x, y in zip(res["1"][0], res["0"][0]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(res["1"][1], res["0"][1])
    for x, y in zip(res["1"][2], res["0"][2]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(res["1"][3], res["0"][3])


@pytest.mark.parametrize("B,T,C,U", [(3, 60, 10, 7)])
def test_lazy_ctc_vs_oracle(gtn, B, T, C, U):
    """per-utterance target graphs (groups of one), chain as the SECOND argument"""
    import torch
    em, tg = gg.ctc_inputs(21, B, T, C, U)
    with lazy_mode("1"):
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
        comp = gtn.intersect(ctcs, ems)
        loss = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(comp))
        gtn.backward(loss)
        got = gtn.items(loss)
        # looking inside the product builds it for real
        oc = OGraph.from_dict(gg.ctc_target_graph(tg[0].tolist())).compose(OGraph.linear(T, C, em[0]), "intersect")
        assert (comp[0].num_nodes(), comp[0].num_arcs()) == (oc.N, oc.A)
    for b in range(B):
        want, wgrad = ctc_loss(em[b], tg[b])
        assert got[b] == pytest.approx(want, rel=RTOL)
        z = abs(float(OGraph.linear(T, C, em[b]).shortest_distance()))
        np.testing.assert_allclose(ems[b].grad().weights_to_numpy().reshape(T, C), wgrad,
                                   rtol=max(RTOL, 8 * 1.2e-7 * z), atol=1e-4)
        tgt = gg.ctc_target_graph(tg[b].tolist())
        o = OGraph.from_dict(tgt).compose(OGraph.linear(T, C, em[b]), "intersect")
        g1, _ = o.compose_grad(o.shortest_distance_grad(), len(tgt["src"]), T * C)
        np.testing.assert_allclose(ctcs[b].grad().weights_to_numpy(), -np.asarray(g1), rtol=1e-3, atol=1e-4)


def test_lazy_no_accepting_path(gtn):
    """a target longer than the emissions: the trimmed product is empty"""
    T, C = 3, 5
    with lazy_mode("1"):
        e = gtn.linear_graph(T, C)
        e.set_weights(np.zeros(T * C, np.float32))
        ctc = gg.to_api(gtn, gg.ctc_target_graph([1, 2, 3, 4, 1, 2]))
        comp = gtn.intersect(ctc, e)
        assert gtn.forward_score(comp).item() == -np.inf
        assert gtn.viterbi_score(comp).item() == -np.inf
        p = gtn.viterbi_path(comp)
        assert (p.num_nodes(), p.num_arcs()) == (0, 0)
        assert comp.num_arcs() == 0


@pytest.mark.parametrize("batched", [False, True])
def test_reference_asg_known_answers_under_lazy_mode(gtn, batched):
    """test/criterion_test.cpp:182-345 (ASG losses, emission and shared-transition
    gradients, Viterbi labels) with every eligible composition kept symbolic"""
    import test_parity_gpu as tp
    with lazy_mode("1"):
        tp.test_asg_criterion(gtn, batched)
        tp.test_asg_viterbi_path(gtn)


def test_reference_ctc_known_answers_under_lazy_mode(gtn):
    import test_parity_gpu as tp
    with lazy_mode("1"):
        tp.test_ctc_criterion_known_answers(gtn)


def test_full_size_c4_invariants(gtn):
    """BASELINE config C4 at FULL size (B=512, T=1000, C=512, dense transitions; 262 M
    product arcs per utterance, never built): posterior mass is 1 per time step, so the
    emission-gradient rows sum to 1 and the transition gradients sum to B*T; the
    Viterbi score is the weight of the Viterbi path and never exceeds the forward score."""
    import torch
    B, T, C = 512, 1000, 512
    torch.manual_seed(0)
    em = (torch.rand(B, T, C, device="cuda") * 10 - 5).contiguous()
    tw = np.random.default_rng(0).random(C * C + C).astype(np.float32)
    trans = gtn.Graph()
    n = np.arange(C)
    trans.add_nodes(np.array([1] + [0] * C, np.uint8), np.array([0] + [1] * C, np.uint8))
    trans.add_arcs(np.concatenate([np.zeros(C, np.int32), np.tile(n + 1, C)]).astype(np.int32),
                   np.concatenate([n + 1, np.repeat(n + 1, C)]).astype(np.int32),
                   np.concatenate([n, np.repeat(n, C)]).astype(np.int32), None, tw)
    ems = gtn.linear_graph_n(B, T, C, em)
    comp = gtn.compose(ems, [trans])
    fs = gtn.forward_score(comp)
    vs = gtn.viterbi_score(comp)
    paths = gtn.viterbi_path(comp[:4])
    gtn.backward(fs)
    f, v = gtn.items(fs), gtn.items(vs)
    assert np.isfinite(f).all() and (v <= f + 1e-3).all()
    for b in range(4):
        assert paths[b].num_arcs() == T
        assert float(paths[b].weights_to_numpy().sum()) == pytest.approx(v[b], rel=1e-5)
    out = torch.empty(B, T, C, device="cuda")
    gtn.grads_to_device(ems, out, [b * T * C for b in range(B)])
    gtn.synchronize()
    rows = out.sum(dim=2)
    # every time step is normalised by its own alpha-beta sum, so the float32 drift between
    # the two sweeps (scores of magnitude ~9.5e3 resolve 1e-3) does not reach the posteriors
    assert float((rows - 1).abs().max()) < 1e-3
    assert float(trans.grad().weights_to_numpy().sum()) == pytest.approx(B * T, rel=1e-3)


def test_dense_regime_with_impossible_emissions_and_transitions(gtn):
    """log(0) = -inf emission frames / transition weights through the probability-domain
    (dense) kernels, the record-walking lazy kernels and the materialised path"""
    B, T, N = 6, 25, 16
    res = {}
    for name, env in (("mat", {"GTNX_LAZY_COMPOSE": "0"}), ("walk", {"GTNX_LAZY_COMPOSE": "1", "GTNX_NO_DENSE": "1"}),
                      ("dense", {"GTNX_LAZY_COMPOSE": "1"})):
        for k, v in env.items():
            os.environ[k] = v
        try:
            em, ems, trans = asg_batch(gtn, B, T, N, 4)
            em = em.copy()
            em[0, 3, :5] = -np.inf
            em[2, :, 7] = -np.inf
            for b in (0, 2):
                ems[b].set_weights(em[b])
            w = trans.weights_to_numpy()
            w[N + 5 * N + 2] = -np.inf      # p(5 | 2) = 0
            w[3] = -np.inf                   # p(3 | <s>) = 0
            trans.set_weights(w)
            comp = gtn.compose(ems, [trans])
            fs = gtn.forward_score(comp)
            gtn.backward(fs)
            res[name] = (gtn.items(fs), [e.grad().weights_to_numpy() for e in ems], trans.grad().weights_to_numpy())
        finally:
            for k in env:
                os.environ.pop(k, None)
    # scores agree with the materialised (reference-order) path.  Its GRADIENTS are NaN
    # wherever a -inf weight meets a -inf score (the reference's exp(-inf - -inf),
    # autograd_test.cpp:339-386); the symbolic paths return the finite limit instead, so
    # they are compared with each other and, where the reference is finite, with it
    for other in ("walk", "dense"):
        np.testing.assert_allclose(res[other][0], res["mat"][0], rtol=1e-5)
        for x, y in zip(res[other][1], res["mat"][1]):
            assert np.isfinite(x).all()
            ok = np.isfinite(y)
            np.testing.assert_allclose(x[ok], y[ok], rtol=1e-3, atol=1e-5)
    for x, y in zip(res["dense"][1], res["walk"][1]):
        np.testing.assert_allclose(x, y, rtol=1e-3, atol=1e-5)
    assert np.isfinite(res["dense"][2]).all()
    np.testing.assert_allclose(res["dense"][2], res["walk"][2], rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------------------
# per-utterance sweep kernels (lazy_pair.hip): compose_mode(2), the path the criteria take
# ---------------------------------------------------------------------------
class pair_mode:
    """gtn.compose_mode(2) for the block; `used()` tells whether the per-utterance sweep kernels ran.
    band=False keeps banded partners (CTC targets) away from band.hip, i.e. on lazy_pair.hip."""

    def __init__(self, gtn, band=True):
        self.gtn = gtn
        self.band = band

    def __enter__(self):
        import os
        self.env = os.environ.pop("GTNX_NO_BAND", None)
        if not self.band:
            os.environ["GTNX_NO_BAND"] = "1"
        self.prev = self.gtn.compose_mode(2)
        self.gtn.prof_reset()
        self.gtn.prof_enable(True)
        return self

    def __exit__(self, *a):
        import os
        self.gtn.prof_enable(False)
        self.names = self.gtn.prof_names()
        self.gtn.compose_mode(self.prev)
        os.environ.pop("GTNX_NO_BAND", None)
        if self.env is not None:
            os.environ["GTNX_NO_BAND"] = self.env

    def used(self):
        return "lazy_pair_forward_score" in self.names or "band_forward_score" in self.names

    def used_band(self):
        return "band_forward_score" in self.names


def _ctc_pair_check(gtn, ems_np, targets, chain_first=False, band=True):
    """CTC losses and both gradients through the sweep kernels (band.hip, or lazy_pair.hip with
    band=False) against the oracle"""
    B = len(targets)
    with pair_mode(gtn, band) as pm:
        ems, ctcs = [], []
        for b in range(B):
            T, C = ems_np[b].shape
            e = gtn.linear_graph(T, C)
            e.set_weights(ems_np[b])
            ems.append(e)
            ctcs.append(gg.to_api(gtn, gg.ctc_target_graph(list(targets[b]))))
        comp = gtn.compose(ems, ctcs) if chain_first else gtn.intersect(ctcs, ems)
        loss = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(comp))
        gtn.backward(loss)
        got = gtn.items(loss)
    assert pm.used(), "the per-utterance sweep kernels did not run"
    # band.hip takes alphabets up to 1024 labels; wider ones stay on lazy_pair.hip
    assert pm.used_band() == (band and min(e.shape[1] for e in ems_np) <= 1024)
    for b in range(B):
        T, C = ems_np[b].shape
        want, wgrad = ctc_loss(ems_np[b], np.asarray(targets[b], np.int32))
        if np.isinf(want):
            assert np.isinf(got[b])
            continue
        assert got[b] == pytest.approx(want, rel=RTOL)
        z = abs(float(OGraph.linear(T, C, ems_np[b]).shortest_distance()))
        np.testing.assert_allclose(ems[b].grad().weights_to_numpy().reshape(T, C), wgrad,
                                   rtol=max(RTOL, 8 * 1.2e-7 * z), atol=1e-4)
        tgt = gg.ctc_target_graph(list(targets[b]))
        o = OGraph.from_dict(tgt).compose(OGraph.linear(T, C, ems_np[b]), "intersect")
        g1, _ = o.compose_grad(o.shortest_distance_grad(), len(tgt["src"]), T * C)
        np.testing.assert_allclose(ctcs[b].grad().weights_to_numpy(), -np.asarray(g1), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("band", [True, False])
@pytest.mark.parametrize("B,T,C,U,chain_first", [
    (4, 50, 9, 5, False),     # odd label count: scalar staging tail
    (3, 300, 64, 30, False),
    (3, 300, 64, 30, True),   # compose(emissions, target): matches the target's ilabels
    (2, 90, 300, 10, False),  # 4 time steps per chunk
    (2, 40, 1100, 6, False),  # 1 time step per chunk
    (2, 400, 20, 150, False), # 301-node targets: 512-lane workgroups
    (1, 1, 4, 1, False),      # a single frame
])
def test_pair_kernels_ctc_vs_oracle(gtn, B, T, C, U, chain_first, band):
    rng = np.random.default_rng(B * 1000 + T + C)
    ems = [rng.normal(0, 1, (T, C)).astype(np.float32) for _ in range(B)]
    tg = [rng.integers(1, C, U).tolist() for _ in range(B)]
    _ctc_pair_check(gtn, ems, tg, chain_first, band)


@pytest.mark.parametrize("B,T,U", [
    (2, 3, 1),      # fewer rows than one block of four
    (2, 7, 3),      # a full block and a partial one: no steady tick at all
    (2, 19, 9),     # the sweepers' first steady tick is the fifth
    (2, 21, 10),
    (2, 45, 20),    # a handful of steady ticks, T not a multiple of four
    (1, 110, 102),  # 205 nodes: LDS row stride 208 (a multiple of 16), the last group of 16 nodes mostly padding
    (1, 115, 104),  # 209 nodes: four rows per block no longer fit beside a second workgroup (two rows per block)
])
def test_band_backward_four_row_blocks_edge_shapes(gtn, B, T, U):
    """C = 256 (four rows per block, one node per sweeper lane: the timed configuration's kernel instantiation, with a
    block's four row sums by one staging wave and steady-state ticks in every role) at the sizes where its loops have
    no steady part, where the rounded LDS row stride differs from the alpha rows' stride in HBM, and where the block
    size falls back to two rows"""
    rng = np.random.default_rng(T * 1000 + U)
    ems = [rng.normal(0, 1, (T, 256)).astype(np.float32) for _ in range(B)]
    tg = [rng.integers(1, 256, U).tolist() for _ in range(B)]
    _ctc_pair_check(gtn, ems, tg)


def test_band_backward_four_row_blocks_dead_and_live_utterances_in_one_launch(gtn):
    """one launch at C = 256 with a target that cannot be aligned (more labels than frames), an empty target, a repeated
    label (needs the blank between: feasible only just) and ordinary ones: the utterance without a path takes the
    drain path of its own (normaliser's term only), the others the steady ticks"""
    rng = np.random.default_rng(77)
    T = 40
    ems = [rng.normal(0, 1, (T, 256)).astype(np.float32) for _ in range(5)]
    tg = [rng.integers(1, 256, 12).tolist(), rng.integers(1, 256, 41).tolist(), [], [7] * 20, rng.integers(1, 256, 19).tolist()]
    _ctc_pair_check(gtn, ems, tg)


@pytest.mark.parametrize("band", [True, False])
def test_pair_kernels_mixed_shapes_in_one_batch(gtn, band):
    """utterances of different length, alphabet and target size in ONE call (launch groups by label
    count), an empty target and a target that cannot be aligned"""
    rng = np.random.default_rng(5)
    shapes = [(30, 7), (55, 7), (12, 19), (30, 7), (3, 6), (20, 5)]
    ems = [rng.normal(0, 1, s).astype(np.float32) for s in shapes]
    tg = [[1, 2, 3], [4, 4, 4, 1], [18, 2], [], [1, 2, 3, 4, 5], [2, 4]]
    _ctc_pair_check(gtn, ems, tg, band=band)


def test_pair_kernels_shared_target_and_general_partner(gtn):
    """one target graph shared by the whole batch (its gradient accumulates) and a partner whose
    in-arcs carry DIFFERENT labels per node (per-arc gradient terms), against the built lattice"""
    B, T, C = 5, 25, 6
    rng = np.random.default_rng(8)
    em = rng.normal(0, 1, (B, T, C)).astype(np.float32)

    def run(mode):
        prev = gtn.compose_mode(mode)
        try:
            g = gtn.Graph()
            for i in range(7):
                g.add_node(i == 0, i >= 5)
            wts = rng_w.normal(0, 1, 16).astype(np.float32)
            arcs = [(0, 1, 1), (0, 2, 2), (1, 1, 1), (1, 2, 3), (2, 3, 4), (3, 3, 5), (1, 3, 2), (3, 4, 0), (2, 4, 1),
                    (4, 5, 2), (4, 4, 3), (4, 6, 4), (5, 6, 5), (5, 5, 0), (6, 6, 1), (3, 5, 9)]  # label 9 never matches
            for k, (s_, d_, l_) in enumerate(arcs):
                g.add_arc(s_, d_, l_, l_, float(wts[k]))
            ems = []
            for b in range(B):
                e = gtn.linear_graph(T, C)
                e.set_weights(em[b])
                ems.append(e)
            fs = gtn.forward_score(gtn.intersect([g], ems))
            gtn.backward(fs)
            return gtn.items(fs), [e.grad().weights_to_numpy() for e in ems], g.grad().weights_to_numpy()
        finally:
            gtn.compose_mode(prev)

    rng_w = np.random.default_rng(3)
    built = run(0)
    rng_w = np.random.default_rng(3)
    with pair_mode(gtn) as pm:
        swept = run(2)
    assert pm.used()
    np.testing.assert_allclose(swept[0], built[0], rtol=RTOL)
    for a, b in zip(swept[1], built[1]):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(swept[2], built[2], rtol=1e-3, atol=1e-5)


def test_pair_mode_other_uses_of_the_composition(gtn):
    """a symbolic composition is still a graph: sizes, Viterbi path and score agree with the built one"""
    import torch
    B, T, C, U = 3, 40, 8, 6
    em, tg = gg.ctc_inputs(31, B, T, C, U)
    res = {}
    for mode in (0, 2):
        prev = gtn.compose_mode(mode)
        try:
            ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
            ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
            comp = gtn.intersect(ctcs, ems)
            res[mode] = (gtn.items(gtn.viterbi_score(comp)), [p.labels_to_list() for p in gtn.viterbi_path(comp)],
                         gtn.items(gtn.forward_score(comp)), [(c.num_nodes(), c.num_arcs()) for c in comp])
        finally:
            gtn.compose_mode(prev)
    np.testing.assert_allclose(res[2][0], res[0][0], rtol=1e-6)
    assert res[2][1] == res[0][1]
    np.testing.assert_allclose(res[2][2], res[0][2], rtol=RTOL)
    assert res[2][3] == res[0][3]


def test_symbolic_composition_inspected_between_forward_and_backward(gtn):
    """Looking inside a symbolic composition after forward_score (which builds it) must not disturb the
    reverse sweep: the record of the built composition is filed where the symbolic one was."""
    import torch
    B, T, C, U = 3, 30, 8, 5
    em, tg = gg.ctc_inputs(77, B, T, C, U)
    grads = {}
    for inspect in (False, True):
        prev = gtn.compose_mode(2)
        try:
            ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
            ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
            comp = gtn.intersect(ctcs, ems)
            fs = gtn.forward_score(comp)
            if inspect:
                assert all(c.num_arcs() > 0 for c in comp)
                assert gtn.items(gtn.viterbi_score(comp)).shape == (B,)
            gtn.backward(fs)
            grads[inspect] = ([e.grad().weights_to_numpy() for e in ems], [c.grad().weights_to_numpy() for c in ctcs])
        finally:
            gtn.compose_mode(prev)
    for a, b in zip(grads[True][0], grads[False][0]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(grads[True][1], grads[False][1]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_gradient_read_on_host_then_accumulated_again(gtn):
    """weights() of a gradient hands out a mutable host pointer; a later backward into the same leaf
    accumulates on the device and must neither re-upload the stale host copy nor fail."""
    T, C = 6, 4
    rng = np.random.default_rng(4)
    e = gtn.linear_graph(T, C)
    e.set_weights(rng.normal(0, 1, (T, C)).astype(np.float32))
    first = None
    for k in range(3):
        gtn.backward(gtn.forward_score(e))
        g = e.grad().weights_to_numpy().copy()
        if first is None:
            first = g
        np.testing.assert_allclose(g, (k + 1) * first, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------
# the criteria's path at the sizes the benchmark times (BASELINE configs C3 / C5), against
# float64 arithmetic (tests/ctc_fp64.py) and against the float32 reference restatement
# ---------------------------------------------------------------------------
def _criterion_path(gtn, em, tg):
    """losses and gradients of benchmarks/ctc.cpp:150-165 through compose_mode(2), as ctcLossBatch runs it"""
    import torch
    B, T, C = em.shape
    with pair_mode(gtn) as pm:
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        ctcs = [gg.to_api(gtn, gg.ctc_target_graph(list(t))) for t in tg]
        comp = gtn.intersect(ctcs, ems)
        loss = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(comp))
        gtn.backward(loss)
        got = gtn.items(loss)
        ge = [e.grad().weights_to_numpy().reshape(T, C) for e in ems]
        gt = [c.grad().weights_to_numpy() for c in ctcs]
    return got, ge, gt, pm


def test_c3_full_size_timed_path_vs_fp64_and_reference(gtn):
    """BASELINE config C3 (T=1000, C=256, U=100) through the kernels bench.py times.  Every loss within 1e-4
    relative and every emission-gradient element within 1e-4 of float64 arithmetic; the float32 reference
    restatement (the C oracle: compose + shortestDistance, unnormalised scores of magnitude ~8.5 T) is measured
    against the same yardstick, so the comparison with it uses ITS error, demonstrated, not declared."""
    from ctc_fp64 import ctc_loss_fp64
    B, T, C, U = 6, 1000, 256, 100
    em, tg = gg.ctc_inputs(1234, B, T, C, U)
    got, ge, gt, pm = _criterion_path(gtn, em, tg)
    assert pm.used_band()
    worst_gpu = worst_ref = 0.0
    for b in range(B):
        l64, g64, _ = ctc_loss_fp64(em[b], tg[b])
        assert got[b] == pytest.approx(l64, rel=1e-4)
        err_gpu = np.abs(ge[b] - g64).max()
        assert err_gpu <= 1e-4, (b, err_gpu)
        worst_gpu = max(worst_gpu, err_gpu)
        if b < 4:  # the reference's float32 arithmetic on the same inputs
            l32, g32 = ctc_loss(em[b], tg[b])
            assert got[b] == pytest.approx(l32, rel=1e-4)
            err_ref = np.abs(g32 - g64).max()
            worst_ref = max(worst_ref, err_ref)
            assert err_gpu <= err_ref + 1e-6, "the sweep kernels are no less accurate than the float32 reference"
            # gradient of the target graph's arcs (benchmarks/ctc.cpp builds it with calcGrad = true)
            tgt = gg.ctc_target_graph(list(tg[b]))
            o = OGraph.from_dict(tgt).compose(OGraph.linear(T, C, em[b]), "intersect")
            g1, _ = o.compose_grad(o.shortest_distance_grad(), len(tgt["src"]), T * C)
            np.testing.assert_allclose(gt[b], -np.asarray(g1), rtol=2e-2, atol=1e-3)
    print(f"C3 full size: max |grad - fp64| band.hip {worst_gpu:.2e}, float32 reference {worst_ref:.2e}")


def test_c5_shape_vs_fp64(gtn):
    """BASELINE config C5's shape (T=2000, C=1024, U=200; 401-node targets: two nodes per lane, 2-row blocks)"""
    from ctc_fp64 import ctc_loss_fp64
    B, T, C, U = 3, 2000, 1024, 200
    em, tg = gg.ctc_inputs(77, B, T, C, U)
    got, ge, gt, pm = _criterion_path(gtn, em, tg)
    assert pm.used_band()
    for b in range(B):
        l64, g64, _ = ctc_loss_fp64(em[b], tg[b])
        assert got[b] == pytest.approx(l64, rel=1e-4)
        assert np.abs(ge[b] - g64).max() <= 1e-4
        assert np.isfinite(gt[b]).all() and gt[b].min() <= 0.0  # d loss / d target arcs = - posteriors


@pytest.mark.parametrize("T,C,U,kernel", [
    (120, 1100, 130, "<512, 4>"),  # 261-node targets, 4 time steps per chunk
    (60, 2100, 130, "<512, 1>"),   # one time step per chunk
])
def test_wide_alphabets_take_the_pair_kernels(gtn, T, C, U, kernel):
    """alphabets beyond band.hip's 1024 labels: lazy_pair.hip's 512-lane instantiations, vs float64"""
    from ctc_fp64 import ctc_loss_fp64
    B = 2
    em, tg = gg.ctc_inputs(5, B, T, C, U)
    got, ge, gt, pm = _criterion_path(gtn, em, tg)
    assert pm.used() and not pm.used_band()
    for b in range(B):
        l64, g64, _ = ctc_loss_fp64(em[b], tg[b])
        assert got[b] == pytest.approx(l64, rel=1e-4)
        assert np.abs(ge[b] - g64).max() <= 2e-4


def test_band_kernels_extreme_dynamic_range(gtn):
    """emissions spanning hundreds of nats per frame (peaky log-softmax outputs, masked classes at -1e4, a -inf):
    log-domain sweeps with per-wave shifts neither underflow nor lose the path; vs float64"""
    from ctc_fp64 import ctc_loss_fp64
    rng = np.random.default_rng(9)
    B, T, C, U = 3, 200, 32, 20
    em = rng.normal(0, 1, (B, T, C)).astype(np.float32)
    em[0] *= 60.0                       # +-200 nats between classes
    em[1, :, 5:9] = -1.0e4              # masked classes
    em[2, 3, 7] = -np.inf
    em[2] -= np.log(np.exp(em[2].astype(np.float64)).sum(1, keepdims=True)).astype(np.float32)
    tg = rng.integers(1, C, (B, U))
    got, ge, gt, pm = _criterion_path(gtn, em, tg)
    assert pm.used_band()
    for b in range(B):
        l64, g64, _ = ctc_loss_fp64(em[b], tg[b])
        assert got[b] == pytest.approx(l64, rel=1e-4)
        assert np.isfinite(ge[b]).all()
        # float32 carries |emission| * 6e-8 per score: with +-350 (log2 units) per frame no float32 implementation
        # resolves posteriors to 1e-4 -- the bar is the float32 reference's own error on the same input
        _, g32 = ctc_loss(em[b], tg[b])
        ok32 = np.isfinite(g32)
        err_ref = np.abs(g32[ok32] - g64[ok32]).max()
        assert np.abs(ge[b] - g64).max() <= max(2e-4, err_ref)


def _oracle_viterbi(em, target, chain_first=False):
    """viterbiScore, the best path's labels and the tropical gradients from the oracle, on the lattice the
    reference would build (shortest.cpp:86-272 over compose.cpp:377-522)"""
    T, C = em.shape
    tgt = gg.ctc_target_graph(list(target))
    a, b = OGraph.from_dict(tgt), OGraph.linear(T, C, em)
    o = b.compose(a, "compose") if chain_first else a.compose(b, "intersect")
    score = o.shortest_distance(tropical=True)
    arcs, has = o.shortest_path()
    d = o.to_dict()
    labels = [d["il"][x] for x in arcs]
    A1, A2 = (T * C, len(tgt["src"])) if chain_first else (len(tgt["src"]), T * C)
    g1, g2 = o.compose_grad(o.shortest_distance_grad(tropical=True), A1, A2)
    g_t, g_e = (g2, g1) if chain_first else (g1, g2)
    return score, labels, has, np.asarray(g_e).reshape(T, C), np.asarray(g_t)


@pytest.mark.parametrize("B,T,C,U,chain_first", [
    (4, 60, 9, 7, False),
    (3, 200, 64, 40, True),
    (2, 300, 20, 140, False),   # 281-node targets: two nodes per lane on the sweeps, 512 lanes here
    (2, 33, 300, 5, False),
    (1, 1, 4, 1, False),
])
def test_band_viterbi_vs_oracle(gtn, B, T, C, U, chain_first):
    """viterbiScore / viterbiPath of a SYMBOLIC CTC composition (band_viterbi_kernel: tropical sweep with
    back-pointers, never built): scores bit-equal to the oracle's on the built lattice, path labels equal,
    gradients of viterbiScore one-hot along the same path"""
    import torch
    rng = np.random.default_rng(T * 7 + C)
    em = rng.normal(0, 2, (B, T, C)).astype(np.float32)
    tg = [rng.integers(1, C, int(rng.integers(max(1, U // 2), U + 1))).tolist() for _ in range(B)]
    prev = gtn.compose_mode(2)
    try:
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t)) for t in tg]
        comp = gtn.compose(ems, ctcs) if chain_first else gtn.intersect(ctcs, ems)
        vs = gtn.viterbi_score(comp)
        paths = gtn.viterbi_path(comp)
        gtn.backward(vs)
        got = gtn.items(vs)
    finally:
        gtn.compose_mode(prev)
    for b in range(B):
        score, labels, has, g_e, g_t = _oracle_viterbi(em[b], tg[b], chain_first)
        if score is None or np.isinf(score):
            assert np.isinf(got[b])
            continue
        assert got[b] == np.float32(score) or abs(got[b] - score) <= 1e-6 * abs(score)
        assert paths[b].labels_to_list() == labels
        assert abs(float(np.sum(paths[b].weights_to_numpy())) - score) <= 1e-3 * max(1.0, abs(score))
        np.testing.assert_array_equal(ems[b].grad().weights_to_numpy().reshape(T, C), g_e)
        np.testing.assert_array_equal(ctcs[b].grad().weights_to_numpy(), g_t)


def test_band_viterbi_exact_ties_follow_the_reference(gtn):
    """integer-valued emissions make exact ties: which arc wins then depends on the built lattice's node
    numbering (in-list order for viterbiScore's gradient, queue order for viterbiPath) -- the kernel flags
    the tie and the utterance runs through the built lattice, so labels and gradients are the oracle's"""
    import torch
    B, T, C = 5, 12, 5
    rng = np.random.default_rng(3)
    em = rng.integers(-1, 2, (B, T, C)).astype(np.float32)
    em[0] = 0.0
    tg = [[1, 2], [3, 3, 1], [4], [1, 2, 3, 4], [2, 2]]
    prev = gtn.compose_mode(2)
    try:
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t)) for t in tg]
        comp = gtn.intersect(ctcs, ems)
        paths = gtn.viterbi_path(comp)
        ems2 = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        vs = gtn.viterbi_score(gtn.intersect(ctcs, ems2))
        gtn.backward(vs)
        got = gtn.items(vs)
    finally:
        gtn.compose_mode(prev)
    for b in range(B):
        score, labels, has, g_e, g_t = _oracle_viterbi(em[b], tg[b])
        assert got[b] == np.float32(score)
        assert paths[b].labels_to_list() == labels
        np.testing.assert_array_equal(ems2[b].grad().weights_to_numpy().reshape(T, C), g_e)


def _moore_graph(gtn, N, C, rng, density, integer, dup):
    """nearly complete graph whose nodes' in-arcs share one label (the dense regime's shape), with
    missing arcs, parallel arcs of different weight, several start and accept nodes and a node
    nobody enters"""
    g = gtn.Graph()
    for n in range(N):
        g.add_node(n < 2, n >= N - 3 or n == 1)
    src, dst, lab, w = [], [], [], []
    for d in range(1, N):  # node 0 has no in-arc
        for s in range(N):
            if rng.random() > density:
                continue
            for _ in range(2 if rng.random() < dup else 1):
                src.append(s)
                dst.append(d)
                lab.append(d % C)
                w.append(float(rng.integers(-2, 3)) if integer else float(rng.normal()))
    order = rng.permutation(len(src))  # arc ids (and with them in-row order) are not sorted by source
    a = lambda v, t: np.asarray(v, t)[order]
    g.add_arcs(a(src, np.int32), a(dst, np.int32), a(lab, np.int32), None, a(w, np.float32))
    return g


@pytest.mark.parametrize("B,T,N,C,integer", [(70, 33, 21, 7, True), (3, 1, 16, 16, False), (130, 20, 40, 40, False),
                                             (9, 50, 64, 11, True)])
def test_maxplus_viterbi_matches_the_record_walking_kernels(gtn, B, T, N, C, integer):
    """maxplus.hip (dense G, tropical semiring: max-plus sweeps without back-pointer planes + back-trace
    from alpha) against the record-walking kernels of lazy.hip (GTNX_NO_DENSE=1), which the oracle pins:
    scores, path labels and weights, and the one-hot gradients must be IDENTICAL, exact ties (integer
    weights), parallel arcs, missing arcs and -inf weights included"""
    res = {}
    # (both keep the first maximum in in-row order here: the re-run of tied paths on the built lattice has its own
    #  test below)
    os.environ["GTNX_NO_TIE_RERUN"] = "1"
    for name in ("walk", "maxplus"):
        if name == "walk":
            os.environ["GTNX_NO_DENSE"] = "1"
        try:
            with lazy_mode("1"):
                rng = np.random.default_rng(1234 + N)
                g = _moore_graph(gtn, N, C, rng, 0.85, integer, 0.15)
                w = g.weights_to_numpy()
                w[rng.integers(0, len(w), 5)] = -np.inf
                g.set_weights(w)
                em = (rng.integers(-3, 4, (B, T, C)) if integer else rng.normal(0, 2, (B, T, C))).astype(np.float32)
                em[0, 0, : C // 2] = -np.inf
                ems = []
                for b in range(B):
                    e = gtn.linear_graph(T, C)
                    e.set_weights(em[b])
                    ems.append(e)
                comp = gtn.compose(ems, [g])
                gtn.prof_reset()
                gtn.prof_enable(True)
                vs = gtn.viterbi_score(comp)
                paths = gtn.viterbi_path(comp)
                gtn.prof_enable(False)
                assert ("maxplus_viterbi" in gtn.prof_names()) == (name == "maxplus")
                gtn.backward(vs)
                res[name] = (gtn.items(vs), [p.labels_to_list() for p in paths], [p.weights_to_numpy() for p in paths],
                             [e.grad().weights_to_numpy() for e in ems], g.grad().weights_to_numpy())
        finally:
            os.environ.pop("GTNX_NO_DENSE", None)
    os.environ.pop("GTNX_NO_TIE_RERUN", None)
    a, b = res["walk"], res["maxplus"]
    np.testing.assert_array_equal(b[0], a[0])
    assert b[1] == a[1]
    for x, y in zip(b[2], a[2]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(b[3], a[3]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(b[4], a[4])
    assert np.isfinite(a[0]).sum() > 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_maxplus_viterbi_path_exact_ties_follow_the_reference(gtn, seed):
    """viterbiPath of a SYMBOLIC product with a dense partner (max-plus walk) under exact ties: the walk reports a
    tie on the best path (gtnx_debug_viterbi_ties) and a product small enough to build is re-run on the built
    lattice with the reference's queue order (shortest.cpp:215-218), so the labels are the oracle's; with the
    re-run switched off the ties are counted as unresolved (what C4-sized products, never buildable, get)"""
    B, T, N, C = 6, 9, 12, 6
    rng = np.random.default_rng(100 + seed)
    with lazy_mode("1"):
        g = _moore_graph(gtn, N, C, rng, 0.9, True, 0.1)
        gd = gg.from_api(g)
        em = rng.integers(-1, 2, (B, T, C)).astype(np.float32)
        em[0] = 0.0

        def run():
            ems = []
            for b in range(B):
                e = gtn.linear_graph(T, C)
                e.set_weights(em[b])
                ems.append(e)
            comp = gtn.compose(ems, [g])
            assert gtn.debug_symbolic_route(comp[0], True) == "maxplus"
            return gtn.viterbi_path(comp)

        seen0, un0 = gtn.debug_viterbi_ties()
        paths = run()
        seen1, un1 = gtn.debug_viterbi_ties()
        assert seen1 > seen0, "integer weights: some best path must run through an exact tie"
        assert un1 == un0, "products of 9 x ~130 arcs are small enough to build"
        for b in range(B):
            o = OGraph.linear(T, C, em[b]).compose(OGraph.from_dict(gd), "compose")
            arcs, has = o.shortest_path()
            d = o.to_dict()
            assert paths[b].labels_to_list() == [d["il"][x] for x in arcs], b
        os.environ["GTNX_NO_TIE_RERUN"] = "1"
        try:
            run()
        finally:
            os.environ.pop("GTNX_NO_TIE_RERUN", None)
        seen2, un2 = gtn.debug_viterbi_ties()
        assert un2 - un1 == seen2 - seen1 > 0


def test_inf_weights_through_a_symbolic_product_are_pinned(gtn):
    """-inf emissions (log 0) under a CTC target, forwardScore + backward: the BUILT lattice gives the reference's
    gradients, NaN included (autograd_test.cpp:339-386: a lattice node that no finite path enters poisons its
    in-arcs with exp(-inf - -inf)), checked against the oracle; the SYMBOLIC sweep (the default for host-built
    targets) gives the same loss, the same gradients wherever the reference's are numbers and, where the reference
    has NaN, the posteriors the NaN swallowed (finite, in [-1, 1]) --
    INTEGRATION.md "Where results can differ" 2: both behaviours are pinned here"""
    import torch
    T, C = 14, 6
    rng = np.random.default_rng(5)
    em = rng.normal(0, 1, (T, C)).astype(np.float32)
    em[3, :] = -np.inf
    em[3, 0] = 0.5       # only the blank survives frame 3
    em[7, 2] = -np.inf   # one label dead in frame 7
    tgt = [2, 4, 2]
    want_loss, want_grad = None, None
    o = OGraph.from_dict(gg.ctc_target_graph(tgt)).compose(OGraph.linear(T, C, em), "intersect")
    want_score = o.shortest_distance(tropical=False)
    g1, g2 = o.compose_grad(o.shortest_distance_grad(tropical=False), len(gg.ctc_target_graph(tgt)["src"]), T * C)
    want_grad = np.asarray(g2, np.float32).reshape(T, C)
    res = {}
    for mode in (0, 2):
        prev = gtn.compose_mode(mode)
        try:
            e = gtn.linear_graph_n(1, T, C, torch.from_numpy(em[None]).cuda())
            c = gg.to_api(gtn, gg.ctc_target_graph(tgt))
            s = gtn.forward_score(gtn.intersect([c], e))
            gtn.backward(s)
            res[mode] = (gtn.items(s)[0], e[0].grad().weights_to_numpy().reshape(T, C))
        finally:
            gtn.compose_mode(prev)
    # built: the reference's values, NaN pattern included
    assert abs(res[0][0] - want_score) <= 1e-4 * abs(want_score)
    np.testing.assert_array_equal(np.isnan(res[0][1]), np.isnan(want_grad))
    ok = ~np.isnan(want_grad)
    np.testing.assert_allclose(res[0][1][ok], want_grad[ok], rtol=1e-4, atol=1e-5)
    # symbolic: same loss; finite everywhere; equal to the reference wherever the reference is finite
    assert abs(res[2][0] - want_score) <= 1e-4 * abs(want_score)
    assert np.isfinite(res[2][1]).all()
    np.testing.assert_allclose(res[2][1][ok], want_grad[ok], rtol=1e-4, atol=1e-5)
    # ... and where it is NaN, the posteriors the NaN swallowed: numbers in [-1, 1]
    assert np.isnan(want_grad).any()
    assert np.all(np.abs(res[2][1][~ok]) <= 1.0 + 1e-6)


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
@pytest.mark.parametrize("chain_first", [False, True])
def test_built_lattice_viterbi_path_exact_ties_follow_the_reference(gtn, seed, chain_first):
    """the same ties on the BUILT lattice (compose mode 0): compose's own schedule breaks them by arc id,
    the reference's shortestPath by the order sources leave its queue -- path_tie_kernel flags an exact tie
    on the path and the graph is rerun on the queue-replaying schedule, so the labels are the oracle's"""
    import torch
    B, T, C = 6, 14, 5
    rng = np.random.default_rng(seed)
    em = rng.integers(-1, 2, (B, T, C)).astype(np.float32)
    em[0] = 0.0
    tg = [[1, 2], [3, 3, 1], [4], [1, 2, 3, 4], [2, 2], [1, 1, 1]]
    prev = gtn.compose_mode(0)
    try:
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t)) for t in tg]
        comp = gtn.compose(ems, ctcs) if chain_first else gtn.intersect(ctcs, ems)
        gtn.prof_reset()
        gtn.prof_enable(True)
        paths = gtn.viterbi_path(comp)
        gtn.prof_enable(False)
        assert "viterbi_path" in gtn.prof_names(), "the built-lattice path kernels did not run"
    finally:
        gtn.compose_mode(prev)
    for b in range(B):
        score, labels, has, g_e, g_t = _oracle_viterbi(em[b], tg[b], chain_first)
        assert paths[b].labels_to_list() == labels
        assert float(paths[b].weights_to_numpy().sum()) == np.float32(score)


from ctc_fp64 import asg_fp64 as _asg_fp64  # noqa: E402  (float64 full-connect ASG term: tests/ctc_fp64.py)


@pytest.mark.parametrize("T", [1, 17])
def test_dense_regime_at_the_real_alphabet_vs_fp64(gtn, T):
    """the matrix-core and max-plus kernels at C4's real shape (C = 512: 513 nodes = 16 column tiles after the
    rotation past the start node, 576 padded sources, 8 x 8 gradient blocks) against a float64 DP: scores,
    emission gradients, the shared transitions' gradient, Viterbi score and path"""
    import torch
    B, C = 3, 512
    rng = np.random.default_rng(7)
    em = rng.normal(0, 2, (B, T, C)).astype(np.float32)
    tw = rng.normal(0, 1, C * C + C).astype(np.float32)
    n = np.arange(C)
    trans = gtn.Graph()
    trans.add_nodes(np.array([1] + [0] * C, np.uint8), np.array([0] + [1] * C, np.uint8))
    trans.add_arcs(np.concatenate([np.zeros(C, np.int32), np.tile(n + 1, C)]).astype(np.int32),
                   np.concatenate([n + 1, np.repeat(n + 1, C)]).astype(np.int32),
                   np.concatenate([n, np.repeat(n, C)]).astype(np.int32), None, tw)
    with lazy_mode("1"):
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        comp = gtn.compose(ems, [trans])
        gtn.prof_reset()
        gtn.prof_enable(True)
        fs = gtn.forward_score(comp)
        vs = gtn.viterbi_score(comp)
        paths = gtn.viterbi_path(comp)
        gtn.prof_enable(False)
        assert "maxplus_viterbi" in gtn.prof_names() and "lazy_forward_score" in gtn.prof_names()
        gtn.backward(fs)
        f, v = gtn.items(fs), gtn.items(vs)
        ge = [e.grad().weights_to_numpy().reshape(T, C) for e in ems]
        gt = trans.grad().weights_to_numpy()
    want_gt = np.zeros(C * C + C)
    for b in range(B):
        Z, g_em, g_tw, vbest, lab = _asg_fp64(em[b], tw)
        want_gt += g_tw
        assert f[b] == pytest.approx(Z, rel=2e-6, abs=1e-4)
        assert np.abs(ge[b] - g_em).max() <= 2e-5
        assert v[b] == pytest.approx(vbest, rel=2e-6, abs=1e-4)
        # (labels are pinned bit-exact to the unmodified reference at this alphabet in
        #  test_c4_alphabet_pinned_to_the_reference; against float64 only the path's own weight is checked: the
        #  float32 optimum may be another path within rounding of the float64 one)
        got_lab = paths[b].labels_to_list()
        assert len(got_lab) == T
        w64 = tw[got_lab[0]] + em[b][0][got_lab[0]] + sum(
            float(tw[C + got_lab[t] * C + got_lab[t - 1]]) + float(em[b][t][got_lab[t]]) for t in range(1, T))
        assert w64 == pytest.approx(vbest, abs=1e-3)
        assert float(paths[b].weights_to_numpy().sum()) == pytest.approx(v[b], rel=1e-5)
    assert np.abs(gt - want_gt).max() <= 1e-4


@pytest.mark.parametrize("T", [17, 100, 1000])
def test_c4_alphabet_pinned_to_the_reference(gtn, T):
    """BASELINE config C4 (ASG, dense transitions, C = 512) against the UNMODIFIED reference compiled here:
    tests/golden/asg_c512.npz (tests/golden/make_golden_c4.py over oracle/_ref; 5 M / 26 M / 262 M product arcs per
    utterance -- T = 1000 is BASELINE's own size: 11 GB and two minutes of the reference per utterance).
    forwardScore / viterbiScore within the north-star's 1e-4 relative, viterbiPath's labels EQUAL, emission gradients
    and the shared transitions' gradient (summed over the utterances) within the reference's own float32 rounding,
    max(1e-4, 8 eps |score|) -- every product kept symbolic (matrix-core and max-plus kernels).  At T = 1000 that
    rounding is 8e-3, so there the gradients are ALSO held to the north star's 1e-4 against float64 arithmetic
    (tests/ctc_fp64.py: asg_fp64), and to being no further from it than the reference's own are (measured: the
    reference's emission gradients are 1.0e-3 from float64 at this size, its transitions' gradient 5.4e-4).
    Mirrors examples/asg.cpp:59-68 and test/criterion_test.cpp:308-345."""
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_c4 as mk
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "asg_c512.npz"))
    key = f"T{T}"
    B = gold[key + "_forward"].shape[0]
    C = mk.C
    em, tw = mk.inputs(T, B, int(gold[key + "_seed"]))
    assert em.astype(np.float64).sum() + 3.0 * tw.astype(np.float64).sum() == pytest.approx(
        float(gold[key + "_input_checksum"]), rel=1e-12), "the seeded inputs are not the ones the fixture was made from"
    trans = mk.transitions(gtn, tw)
    with lazy_mode("1"):
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        comp = gtn.compose(ems, [trans])
        fs = gtn.forward_score(comp)
        vs = gtn.viterbi_score(comp)
        paths = gtn.viterbi_path(comp)
        gtn.backward(fs)
        f, v = gtn.items(fs), gtn.items(vs)
        ge = np.stack([e.grad().weights_to_numpy().reshape(T, C) for e in ems])
        gt = trans.grad().weights_to_numpy()
    np.testing.assert_allclose(f, gold[key + "_forward"], rtol=1e-4)
    np.testing.assert_allclose(v, gold[key + "_viterbi"], rtol=1e-4)
    for b in range(B):
        assert paths[b].labels_to_list() == gold[key + "_labels"][b].tolist(), f"utterance {b}: Viterbi labels differ from the reference's"
        assert float(paths[b].weights_to_numpy().sum()) == pytest.approx(float(gold[key + "_viterbi"][b]), rel=1e-5)
    tol = max(1e-4, 8 * np.finfo(np.float32).eps * float(np.abs(gold[key + "_forward"]).max()))
    np.testing.assert_allclose(ge, gold[key + "_grad_emissions"], rtol=tol, atol=tol)
    # (an arc of the transitions collects up to B * T posteriors)
    np.testing.assert_allclose(gt, gold[key + "_grad_transitions"], rtol=tol, atol=tol)
    if T >= 1000:
        want_gt = np.zeros(C * C + C)
        for b in range(B):
            Z, g_em, g_tw, vbest, lab = _asg_fp64(em[b], tw)
            want_gt += g_tw
            assert f[b] == pytest.approx(Z, rel=1e-4) and v[b] == pytest.approx(vbest, rel=1e-4)
            err_gpu = np.abs(ge[b] - g_em).max()
            err_ref = np.abs(gold[key + "_grad_emissions"][b] - g_em).max()
            print(f"C4 full size, utterance {b}: max |emission grad - fp64| engine {err_gpu:.2e}, float32 reference {err_ref:.2e}")
            assert err_gpu <= 1e-4, (b, err_gpu)
            assert err_gpu <= err_ref + 1e-6
        err_gpu = np.abs(gt - want_gt).max()
        err_ref = np.abs(gold[key + "_grad_transitions"] - want_gt).max()
        print(f"C4 full size: max |transitions grad - fp64| engine {err_gpu:.2e}, float32 reference {err_ref:.2e}")
        assert err_gpu <= 1e-4 and err_gpu <= err_ref + 1e-6


def _route_case(kind, rng):
    """(partner graph dict, T, C) steering a symbolic product to one row of the route table"""
    if kind == "ctc":            # banded: CTC target
        C, T = 12, 30
        return gg.ctc_target_graph(rng.integers(1, C, 7).tolist()), T, C
    if kind == "small_skip":     # not banded (an arc jumps three nodes), small: the per-pair kernels
        C, T, N = 10, 9, 9
        d = {"start": [1] + [0] * (N - 1), "accept": [0] * (N - 1) + [1], "src": [], "dst": [], "il": [], "ol": [], "w": [], "sort": None}
        for n in range(N - 1):
            for dst, l in ((n, n % C), (n + 1, (n + 1) % C), (min(n + 3, N - 1), (n + 3) % C)):
                d["src"].append(n); d["dst"].append(dst); d["il"].append(l); d["ol"].append(l)
        d["w"] = gg._f32(rng.normal(0, 1, len(d["src"])))
        return d, T, C
    if kind == "dense":          # one label per node's in-arcs, complete: dense regime / max-plus
        C, T = 24, 11
        d = {"start": [1] * C, "accept": [1] * C, "src": [i for i in range(C) for j in range(C)], "dst": [j for i in range(C) for j in range(C)],
             "il": [j for i in range(C) for j in range(C)], "ol": [j for i in range(C) for j in range(C)], "w": [], "sort": None}
        d["w"] = gg._f32(rng.normal(0, 1, C * C))
        return d, T, C
    # "wide_sparse": nodes with eight out-arcs (more than the per-pair kernels cache), a tenth of the pairs
    # connected (not dense), arcs jumping anywhere (not banded): the record-walking kernels
    C, T, N = 16, 8, 80
    d = {"start": [1] + [0] * (N - 1), "accept": [int(x) for x in rng.random(N) < 0.3], "src": [], "dst": [], "il": [], "ol": [], "w": [], "sort": None}
    d["accept"][-1] = 1
    for n in range(N):
        for dst in rng.choice(N, 8, replace=False):
            d["src"].append(n); d["dst"].append(int(dst)); d["il"].append(int(dst) % C); d["ol"].append(int(dst) % C)
    d["w"] = gg._f32(rng.normal(0, 1, len(d["src"])))
    return d, T, C


@pytest.mark.parametrize("kind,env,log_route,trop_route", [
    ("ctc", {}, "band", "band"),
    ("ctc", {"GTNX_NO_BAND": "1"}, "pair", "walk"),
    ("ctc", {"GTNX_NO_LAZY_PAIRS": "1"}, "walk", "band"),
    ("small_skip", {}, "pair", "walk"),
    ("dense", {}, "dense_mfma", "maxplus"),
    ("dense", {"GTNX_DENSE_VALU": "1"}, "dense", "maxplus"),
    ("dense", {"GTNX_NO_DENSE": "1"}, "walk", "walk"),
    ("wide_sparse", {}, "walk", "walk"),
])
def test_route_table_of_symbolic_products(gtn, kind, env, log_route, trop_route):
    """gtn_amd/csrc/ops_symbolic.cpp is the one place that picks the kernels for a symbolic chain product: every
    row of its table is reached here (gtnx_debug_symbolic_route says which), with the chain on either side, and
    the scores / best path / gradients of that route are the built lattice's"""
    rng = np.random.default_rng(sum(map(ord, kind)))
    d, T, C = _route_case(kind, rng)
    em = rng.normal(0, 1, (T, C)).astype(np.float32)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for chain_first in (False, True):
            res = {}
            for mode in ("0", "1"):
                with lazy_mode(mode):
                    e = gtn.linear_graph(T, C)
                    e.set_weights(em)
                    p = gg.to_api(gtn, d)
                    comp = gtn.intersect(e, p) if chain_first else gtn.intersect(p, e)
                    if mode == "1":
                        assert gtn.debug_symbolic_route(comp, False) == log_route
                        assert gtn.debug_symbolic_route(comp, True) == trop_route
                    else:
                        assert gtn.debug_symbolic_route(comp, False) is None
                    fs, vs = gtn.forward_score(comp), gtn.viterbi_score(comp)
                    path = gtn.viterbi_path(comp)
                    gtn.backward(fs)
                    res[mode] = (fs.item(), vs.item(), path.labels_to_list(), e.grad().weights_to_numpy(), p.grad().weights_to_numpy())
            a, b = res["0"], res["1"]
            assert b[0] == pytest.approx(a[0], rel=RTOL) and b[1] == pytest.approx(a[1], rel=1e-6)
            assert b[2] == a[2]
            np.testing.assert_allclose(b[3], a[3], rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(b[4], a[4], rtol=1e-3, atol=1e-4)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_default_policy_keeps_host_built_targets_symbolic(gtn):
    """gtnx_compose_mode -1 (what a caller who never set it gets): intersect(target built on the host, emissions)
    stays symbolic and is swept by band.hip; a partner that only exists on the device (the product of an earlier
    compose) or that is too wide is built.  Same loss, gradients and best path as the built lattice; looking at
    the arcs builds it."""
    rng = np.random.default_rng(3)
    T, C = 60, 14
    em = rng.normal(0, 1, (T, C)).astype(np.float32)
    tgt = gg.ctc_target_graph(rng.integers(1, C, 9).tolist())
    old = gtn.compose_mode(-1)
    try:
        res = {}
        for mode in (-1, 0):
            gtn.compose_mode(mode)
            e = gtn.linear_graph(T, C)
            e.set_weights(em)
            ctc = gg.to_api(gtn, tgt)
            comp = gtn.intersect(ctc, e)
            assert gtn.debug_symbolic_route(comp) == ("band" if mode < 0 else None)
            loss = gtn.subtract(gtn.forward_score(e), gtn.forward_score(comp))
            gtn.backward(loss)
            res[mode] = (loss.item(), e.grad().weights_to_numpy(), ctc.grad().weights_to_numpy(),
                         gtn.viterbi_path(comp).labels_to_list())
            if mode < 0:
                n_arcs = comp.num_arcs()                      # looking inside builds it
                assert gtn.debug_symbolic_route(comp) is None and n_arcs > T
        a, b = res[-1], res[0]
        assert a[0] == pytest.approx(b[0], rel=RTOL) and a[3] == b[3]
        np.testing.assert_allclose(a[1], b[1], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-3, atol=1e-4)
        # a device-built partner (ctc o bigram) and a wide one (the bigram graph itself) are built
        gtn.compose_mode(-1)
        d = {"start": [1] * C, "accept": [1] * C, "src": [i for i in range(C) for j in range(C)], "dst": [j for i in range(C) for j in range(C)],
             "il": [j for i in range(C) for j in range(C)], "ol": [j for i in range(C) for j in range(C)],
             "w": gg._f32(rng.normal(0, 1, C * C)), "sort": "i"}
        bigram = gg.to_api(gtn, d)
        e = gtn.linear_graph(T, C)
        e.set_weights(em)
        assert gtn.debug_symbolic_route(gtn.intersect(e, bigram)) is None
        assert gtn.debug_symbolic_route(gtn.intersect(gtn.intersect(gg.to_api(gtn, tgt), bigram), e)) is None
    finally:
        gtn.compose_mode(old)


def test_dense_regime_beta_sweep_beside_the_alpha_sweep_life_cycle(gtn):
    """The backward sweep of a dense product starts with its forward sweep on a second stream (ops_lazy.cpp:
    LazyGroupState::EagerBeta).  What must hold whatever the caller does next: a forwardScore that is never
    differentiated can be dropped while the sweep may still be running; backward with retain_graph twice accumulates
    like the reference (autograd.cpp:40-67: the root holds 1, then 2; the product's gradient 1 P, then 3 P; its
    inputs FOUR times the gradient of one backward -- ops.cpp through_delta keeps that sum for a product that has no
    gradient of its own); and two products in flight at once do not share sweep buffers."""
    import torch
    B, C, T = 2, 64, 9
    rng = np.random.default_rng(11)
    em = rng.normal(0, 2, (B, T, C)).astype(np.float32)
    tw = rng.normal(0, 1, C * C + C).astype(np.float32)
    n = np.arange(C)

    def transitions():
        g = gtn.Graph()
        g.add_nodes(np.array([1] + [0] * C, np.uint8), np.array([0] + [1] * C, np.uint8))
        g.add_arcs(np.concatenate([np.zeros(C, np.int32), np.tile(n + 1, C)]).astype(np.int32),
                   np.concatenate([n + 1, np.repeat(n + 1, C)]).astype(np.int32),
                   np.concatenate([n, np.repeat(n, C)]).astype(np.int32), None, tw)
        return g

    with lazy_mode("1"):
        def product():
            tr = transitions()
            ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
            fs = gtn.forward_score(gtn.compose(ems, [tr]))
            return tr, ems, fs
        # one backward (from here on the process has been seen to differentiate dense products: the early sweep is on)
        tr1, ems1, fs1 = product()
        first = gtn.items(fs1)
        gtn.backward(fs1)
        g1 = tr1.grad().weights_to_numpy().copy()
        e1 = ems1[0].grad().weights_to_numpy().copy()
        # dropped without a backward, with the early sweep possibly still running (that turns it off again) ...
        tr, ems, fs = product()
        assert gtn.items(fs) == pytest.approx(first, rel=1e-6)
        del tr, ems, fs
        # ... so this backward runs the sweep itself and turns it on for the products below
        tr, ems, fs = product()
        gtn.backward(fs)
        assert np.abs(tr.grad().weights_to_numpy() - g1).max() <= 1e-5
        del tr, ems, fs
        # two products in flight, the second differentiated twice with the graph retained
        tr2, ems2, fs2 = product()
        tr3, ems3, fs3 = product()
        gtn.backward(fs3, retain_graph=True)
        gtn.backward(fs3, retain_graph=True)
        gtn.backward(fs2)
        assert np.abs(tr2.grad().weights_to_numpy() - g1).max() <= 1e-5
        assert np.abs(ems2[0].grad().weights_to_numpy() - e1).max() <= 1e-6
        assert np.abs(tr3.grad().weights_to_numpy() - 4 * g1).max() <= 4e-5
        assert np.abs(ems3[0].grad().weights_to_numpy() - 4 * e1).max() <= 4e-6


@pytest.mark.parametrize("mode", ["0", "2"])
def test_backward_twice_with_retain_graph_through_a_ctc_product(gtn, mode):
    """backward(loss, retain_graph=True) twice on forwardScore(intersect(ctc, emissions)): the reference accumulates
    the product's gradient over the passes (autograd.cpp:40-67), so the emissions and the target get 4 x the
    single-pass gradient -- on the built lattice (mode 0: the compose gradient fused into forwardScore's backward)
    and on the symbolic product (mode 2: band sweeps) alike."""
    import torch
    import graphgen as gg
    T, C, U = 40, 12, 6
    em, tg = gg.ctc_inputs(5, 1, T, C, U)

    def loss_and_inputs():
        ctc = gg.to_api(gtn, gg.ctc_target_graph(tg[0].tolist()))
        ctc.arc_sort()
        e = gtn.linear_graph_n(1, T, C, torch.from_numpy(em).cuda())[0]
        return ctc, e, gtn.forward_score(gtn.intersect(ctc, e))

    with lazy_mode(mode):
        c1, e1, l1 = loss_and_inputs()
        gtn.backward(l1)
        ge, gc = e1.grad().weights_to_numpy().copy(), c1.grad().weights_to_numpy().copy()
        c2, e2, l2 = loss_and_inputs()
        gtn.backward(l2, retain_graph=True)
        gtn.backward(l2, retain_graph=True)
        assert np.abs(e2.grad().weights_to_numpy() - 4 * ge).max() <= 1e-4
        assert np.abs(c2.grad().weights_to_numpy() - 4 * gc).max() <= 1e-4


@pytest.mark.parametrize("ctc_first,B,T,C,Umax", [(True, 24, 40, 8, 8), (False, 24, 40, 8, 8), (True, 6, 120, 8, 50),
                                                 (False, 5, 260, 12, 110), (True, 4, 340, 8, 150)])
def test_band_viterbi_ties_of_sorted_ctc_targets_are_decided_by_node_ranks(gtn, ctc_first, B, T, C, Umax):
    """integer-valued emissions: nearly every utterance has exact ties on its best path.  For arcSort'ed CTC targets
    they are decided by a second launch with the reference's queue / creation order as node ranks (ops_band.cpp:
    tie_ranks; band_viterbi_wave_kernel<NPL, RANKED>) -- no lattice is built -- and the answers are the UNMODIFIED
    reference's: viterbiPath's labels and weights, viterbiScore's scores and its one-hot emission gradients."""
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbackend"))
    try:
        import gtn_ref as ref
    except Exception as e:
        pytest.skip("needs oracle/_ref: %s" % e)
    # (Umax 8 / 50 / 110 / 150: 2 U + 1 nodes = one / two / four / eight nodes per lane of the wave kernel)
    rng = np.random.default_rng(17 + Umax)
    em = rng.integers(-1, 2, (B, T, C)).astype(np.float32)
    em[0] = 0.0
    tgs = []
    for b in range(B):
        t = []
        for _ in range(int(rng.integers(max(1, Umax - 7), Umax + 1))):
            t.append(t[-1] if t and rng.random() < 0.3 else int(rng.integers(1, C)))
        tgs.append(t)

    def run(api, em_graphs):
        ctcs = [gg.to_api(api, gg.ctc_target_graph(t)) for t in tgs]
        for g in ctcs:
            g.arc_sort()
        comp = api.intersect(ctcs, em_graphs) if ctc_first else api.intersect(em_graphs, ctcs)
        return ctcs, comp

    prev = gtn.compose_mode(2)
    try:
        ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        gtn.prof_reset()
        gtn.prof_enable(True)
        _, comp = run(gtn, ems)
        paths = gtn.viterbi_path(comp)
        ems2 = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
        _, comp2 = run(gtn, ems2)
        vs = gtn.viterbi_score(comp2)
        gtn.backward(vs)
        gtn.prof_enable(False)
        names = gtn.prof_names()
        got_scores = gtn.items(vs)
    finally:
        gtn.compose_mode(prev)
    if not (os.environ.get("GTNX_NO_RANKED_TIES") or os.environ.get("GTNX_VITERBI_WG")):  # (those take the built lattice)
        if os.environ.get("GTNX_NO_RANKED_FIRST"):
            # rounds 4-5: a second launch over the tied utterances, decided by node ranks
            assert "band_viterbi_path_ranked" in names and "band_viterbi_score_ranked" in names
        else:
            # round 6: the ranks are known before the first launch (closed form, ops_band.cpp: tie_ranks) and that launch
            # decides the ties itself -- no second one
            assert "band_viterbi_path" in names and "band_viterbi_score" in names
            assert "band_viterbi_path_ranked" not in names and "band_viterbi_score_ranked" not in names
        assert "intersect" not in names and "viterbi_path" not in names  # nothing was built
    rems = []
    for b in range(B):
        e = ref.linear_graph(T, C)
        e.set_weights(em[b].reshape(-1))
        rems.append(e)
    _, rcomp = run(ref, rems)
    rpaths = ref.viterbi_path(rcomp)
    rems2 = []
    for b in range(B):
        e = ref.linear_graph(T, C)
        e.set_weights(em[b].reshape(-1))
        rems2.append(e)
    _, rcomp2 = run(ref, rems2)
    rvs = ref.viterbi_score(rcomp2)
    ref.backward(rvs)
    want_scores = ref.items(rvs)
    for b in range(B):
        assert paths[b].labels_to_list() == rpaths[b].labels_to_list(), b
        np.testing.assert_array_equal(paths[b].weights_to_numpy(), rpaths[b].weights_to_numpy())
        assert got_scores[b] == want_scores[b]
        np.testing.assert_array_equal(ems2[b].grad().weights_to_numpy(), rems2[b].grad().weights_to_numpy())


@pytest.mark.parametrize("B,T,C,sort", [(4, 9, 9, True), (3, 14, 17, True), (3, 14, 17, False), (2, 12, 40, True)])
def test_maxplus_viterbi_path_ties_on_asg_transitions_follow_the_reference(gtn, B, T, C, sort):
    """viterbiPath of emissions o (ASG transitions), never built, under exact ties (integer weights), with the re-run
    on the built lattice switched OFF: the max-plus walk itself breaks a tie by the smallest source node (and takes
    the first of equal accept nodes) -- for a graph whose every node reaches every label node, lists in node order,
    that IS the order the reference's queue visits the sources in every layer (ops_lazy.cpp:
    dense_ties_by_node_order), so the labels are the UNMODIFIED reference's and no tie is left unresolved.  C = 17, 40:
    alphabets whose in-lists std::sort leaves in no particular order (the old in-row rule failed there)."""
    import torch
    if os.environ.get("GTNX_NO_NODE_ORDER_TIES"):
        pytest.skip("the switch brings the in-row rule back, which is what this test shows to be wrong here")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbackend"))
    try:
        import gtn_ref as ref
    except Exception as e:
        pytest.skip("needs oracle/_ref: %s" % e)

    def transitions(api, tw):
        g = api.Graph()
        g.add_node(True, False)
        for c in range(C):
            g.add_node(False, True)
        for c in range(C):
            g.add_arc(0, c + 1, c, c, float(tw[c]))
        for s in range(C):
            for d in range(C):
                g.add_arc(s + 1, d + 1, d, d, float(tw[C + s * C + d]))
        if sort:
            g.arc_sort()
        return g

    os.environ["GTNX_NO_TIE_RERUN"] = "1"
    try:
        seen0, un0 = gtn.debug_viterbi_ties()
        for seed in range(4):
            rng = np.random.default_rng(seed * 7 + C)
            em = rng.integers(-1, 2, (B, T, C)).astype(np.float32)
            tw = rng.integers(-1, 2, C * C + C).astype(np.float32)
            with lazy_mode("1"):
                ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
                comp = gtn.compose(ems, [transitions(gtn, tw)])
                assert gtn.debug_symbolic_route(comp[0], True) == "maxplus"
                paths = gtn.viterbi_path(comp)
            rtr = transitions(ref, tw)
            for b in range(B):
                e = ref.linear_graph(T, C)
                e.set_weights(em[b].reshape(-1))
                want = ref.viterbi_path(ref.compose(e, rtr))
                assert paths[b].labels_to_list() == want.labels_to_list(), (seed, b)
                np.testing.assert_array_equal(paths[b].weights_to_numpy(), want.weights_to_numpy())
        seen1, un1 = gtn.debug_viterbi_ties()
        assert seen1 > seen0 and un1 == un0
    finally:
        os.environ.pop("GTNX_NO_TIE_RERUN", None)
