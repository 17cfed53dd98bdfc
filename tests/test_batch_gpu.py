"""Batch records (gtnx_batch_*, gtn_amd.Batch): B graphs as one object.

Parity = the per-element results of the reference's per-graph functions: the C oracle
(oracle/gtn_oracle.c, pinned on the reference in tests/test_oracle.py) for the CTC loss and
its gradients, and this engine's own per-graph path -- itself oracle-checked in
tests/test_parity_gpu.py -- for everything the batch functions hand over to it.
Reference: benchmarks/ctc.cpp:40-58,150-165; functions.cpp:18-64,225-251,320-330.
"""
import numpy as np
import pytest

import graphgen as gg
from oracle_lib import ctc_loss

pytestmark = pytest.mark.gpu


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")


def _ragged_inputs(seed, B, T, C, Umax, Umin=0):
    rng = np.random.default_rng(seed)
    em = (rng.random((B, T, C), dtype=np.float32) * 10 - 5).astype(np.float32)
    tg = [rng.integers(1, C, size=int(rng.integers(Umin, Umax + 1))).astype(np.int32) for _ in range(B)]
    return em, tg


def _batch_ctc(gtn, em_dev, tg, T, C, target_grad=True, bind=None):
    B = len(tg)
    ctcs = gtn.Batch.ctc_targets(tg, 0, target_grad)
    ems = gtn.Batch.linear(B, T, C, em_dev, True, True)
    if bind is not None:
        ems.bind_grads(bind, np.arange(B, dtype=np.int64) * T * C)
    score = gtn.forward_score(gtn.intersect(ctcs, ems))
    loss = gtn.subtract(gtn.forward_score(ems), score)
    return ctcs, ems, loss


def test_batch_subtract_into_writes_the_callers_memory(gtn):
    """gtnx_batch_subtract_into: the losses land in the caller's tensor with no copy (what criteria::ctcLossBatch does);
    items_to_device to the same tensor is a no-op, to another one a copy; an element taken as a graph owns its value;
    the gradients are those of subtract"""
    import torch
    B, T, C = 6, 50, 20
    em, tg = _ragged_inputs(3, B, T, C, 8)
    em_dev = _dev(em)
    ctcs = gtn.Batch.ctc_targets(tg, 0, True)
    ems = gtn.Batch.linear(B, T, C, em_dev, True, True)
    out = torch.full((B,), float("nan"), device="cuda:0")
    score = gtn.forward_score(gtn.intersect(ctcs, ems))  # (the order of _batch_ctc: the sweep leaves the normaliser behind)
    loss = gtn.subtract_into(gtn.forward_score(ems), score, out)
    gtn.backward(loss)
    gtn.synchronize()
    got = out.cpu().numpy().copy()
    want = np.array([ctc_loss(em[b], tg[b])[0] for b in range(B)], np.float32)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    loss.items_to_device(out)  # (same address: nothing to do)
    other = torch.zeros(B, device="cuda:0")
    loss.items_to_device(other)
    gtn.synchronize()
    np.testing.assert_array_equal(other.cpu().numpy(), got)
    np.testing.assert_allclose(np.asarray(loss.items()), got)
    g0 = loss[2]  # an element as a graph: the batch takes a block of its own
    out.fill_(0.0)
    gtn.synchronize()
    assert g0.item() == pytest.approx(float(got[2]))
    ctcs2, ems2, loss2 = _batch_ctc(gtn, em_dev, tg, T, C)
    gtn.backward(loss2)
    for b in range(B):
        np.testing.assert_array_equal(ems[b].grad().weights_to_numpy(), ems2[b].grad().weights_to_numpy())


@pytest.mark.parametrize("B,T,C,Umax", [(5, 40, 12, 9), (3, 150, 29, 40), (4, 64, 260, 30), (2, 33, 7, 140)])
def test_batch_ctc_loss_vs_oracle(gtn, B, T, C, Umax):
    """losses and emission gradients of the batch path against the oracle, ragged targets (empty
    ones included), odd and wide alphabets, targets of more than 256 nodes"""
    import torch
    em, tg = _ragged_inputs(11 + B, B, T, C, Umax)
    tg[0] = tg[0][:0] if B > 2 else tg[0]  # an empty target
    em_dev = _dev(em)
    grad = torch.full((B, T, C), float("nan"), device="cuda:0")
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, tg, T, C, bind=grad)
    gtn.backward(loss)
    got = loss.items()
    ems.grads_to_device(grad, np.arange(B, dtype=np.int64) * T * C)
    g = grad.cpu().numpy()
    from ctc_fp64 import ctc_loss_fp64
    for b in range(B):
        want, wgrad = ctc_loss(em[b], tg[b])
        if np.isinf(want):
            assert np.isinf(got[b])
            continue
        assert abs(got[b] - want) <= 1e-4 * max(1.0, abs(want)), (b, got[b], want)
        # gradients against float64: the float32 oracle restates the reference's unnormalised running
        # scores and carries ~8 eps |score| of relative noise itself -- the batch
        # path has to be at least as close to exact arithmetic as the oracle is
        _, exact, _ = ctc_loss_fp64(em[b], tg[b])
        err = np.abs(g[b] - exact).max()
        assert err <= 2e-5, err
        assert err <= np.abs(wgrad - exact).max() + 2e-6


def test_batch_elements_are_the_reference_graphs(gtn):
    """elements taken out of a target batch are the graphs of benchmarks/ctc.cpp:40-58: same nodes,
    arc ids and labels; the device-built records give them the gradients the per-graph path gives"""
    B, T, C = 6, 50, 20
    em, tg = _ragged_inputs(5, B, T, C, 12, Umin=1)
    tg[1] = np.array([3, 3, 3, 4, 4], np.int32)  # repeated labels: no skip arcs there
    em_dev = _dev(em)
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, tg, T, C)
    gtn.backward(loss)
    # per-graph path on the same inputs
    ref_t = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
    ref_e = gtn.linear_graph_n(B, T, C, em_dev)
    prev = gtn.compose_mode(2)
    try:
        ref_l = gtn.subtract(gtn.forward_score(ref_e), gtn.forward_score(gtn.intersect(ref_t, ref_e)))
    finally:
        gtn.compose_mode(prev)
    gtn.backward(ref_l)
    for b in range(B):
        el = ctcs[b]
        assert gtn.equal(el, ref_t[b])
        np.testing.assert_allclose(el.grad().weights_to_numpy(), ref_t[b].grad().weights_to_numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(ems[b].grad().weights_to_numpy(), ref_e[b].grad().weights_to_numpy(), rtol=2e-4,
                                   atol=2e-5)
        assert abs(loss[b].item() - ref_l[b].item()) <= 1e-4 * max(1.0, abs(ref_l[b].item()))


def test_batch_fallbacks_match_per_graph_path(gtn):
    """what the batch functions do not cover natively runs through the vector functions on the
    elements: viterbi of a product batch, labels outside the alphabet, target-first and
    emissions-first products, batches made from ordinary graphs"""
    B, T, C = 4, 30, 9
    em, tg = _ragged_inputs(21, B, T, C, 6, Umin=1)
    em_dev = _dev(em)
    ctcs = gtn.Batch.ctc_targets(tg, 0, False)
    ems = gtn.Batch.linear(B, T, C, em_dev, True, False)
    ref_t = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist()), calc_grad=False) for t in tg]
    ref_e = gtn.linear_graph_n(B, T, C, em_dev)
    # viterbi score and path of the product
    vs = gtn.viterbi_score(gtn.intersect(ctcs, ems)).items()
    want = gtn.items(gtn.viterbi_score(gtn.intersect(ref_t, ref_e)))
    np.testing.assert_allclose(vs, want, rtol=1e-5)
    vp = gtn.viterbi_path(gtn.compose(ems, ctcs))
    rp = gtn.viterbi_path(gtn.compose(ref_e, ref_t))
    for b in range(B):
        assert vp[b].labels_to_list() == rp[b].labels_to_list()
    # emissions-first product, forward score, backward through the batch tape
    fs = gtn.forward_score(gtn.compose(ems, ctcs))
    gtn.backward(gtn.negate(fs))
    rf = gtn.forward_score(gtn.compose(ref_e, ref_t))
    gtn.backward(gtn.negate(rf))
    np.testing.assert_allclose(fs.items(), gtn.items(rf), rtol=1e-5)
    for b in range(B):
        np.testing.assert_allclose(ems[b].grad().weights_to_numpy(), ref_e[b].grad().weights_to_numpy(), rtol=2e-4,
                                   atol=2e-5)
    # a label outside the alphabet: the product is built by the ordinary compose (it matches nothing)
    bad = gtn.Batch.ctc_targets([np.array([C + 3], np.int32)], 0, False)
    one = gtn.Batch.linear(1, T, C, em_dev, False, False)
    assert np.isinf(gtn.forward_score(gtn.intersect(bad, one)).items()[0])
    # a batch made from ordinary graphs
    wrapped = gtn.Batch(ref_t)
    fs2 = gtn.forward_score(gtn.intersect(wrapped, gtn.Batch.linear(B, T, C, em_dev, False, False)))
    np.testing.assert_allclose(fs2.items(), gtn.items(rf), rtol=1e-5)


def test_batch_backward_twice_accumulates(gtn):
    """a second backward over a retained tape adds to every gradient on the way, the seed included
    (autograd.cpp:44-62, graph.cpp:108-129): same numbers as the per-graph path; without retain a
    second backward throws as the reference's does (autograd_test.cpp:57-60)"""
    import torch
    B, T, C = 3, 25, 8
    em, tg = _ragged_inputs(3, B, T, C, 5, Umin=1)
    em_dev = _dev(em)
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, tg, T, C)
    gtn.backward(loss, True)
    gtn.backward(loss)
    off = np.arange(B, dtype=np.int64) * T * C
    g2 = torch.empty(B, T, C, device="cuda:0")
    ems.grads_to_device(g2, off)
    ref_t = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
    ref_e = gtn.linear_graph_n(B, T, C, em_dev)
    prev = gtn.compose_mode(2)
    try:
        ref_l = gtn.subtract(gtn.forward_score(ref_e), gtn.forward_score(gtn.intersect(ref_t, ref_e)))
    finally:
        gtn.compose_mode(prev)
    gtn.backward(ref_l, True)
    gtn.backward(ref_l)
    want = np.stack([ref_e[b].grad().weights_to_numpy().reshape(T, C) for b in range(B)])
    np.testing.assert_allclose(g2.cpu().numpy(), want, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(ctcs[0].grad().weights_to_numpy(), ref_t[0].grad().weights_to_numpy(), rtol=2e-4, atol=2e-5)
    with pytest.raises(ValueError):
        gtn.backward(loss)


def test_batch_ctc_full_size_c3_invariants(gtn):
    """BASELINE config C3 through the batch path (what bench.py times): four utterances against the
    float64 forward-backward, every gradient row sums to zero, losses finite"""
    import torch
    from ctc_fp64 import ctc_loss_fp64
    B, T, C, U = 64, 1000, 256, 100
    em, tg = gg.ctc_inputs(77, B, T, C, U)
    em_dev = _dev(em)
    grad = torch.empty(B, T, C, device="cuda:0")
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, list(tg), T, C, bind=grad)
    gtn.backward(loss)
    got = loss.items()
    assert np.all(np.isfinite(got))
    rows = grad.sum(dim=2).abs().max().item()
    assert rows < 5e-4, rows
    g = grad[:4].cpu().numpy()
    for b in range(4):
        want, wgrad, _ = ctc_loss_fp64(em[b], tg[b])
        assert abs(got[b] - want) <= 1e-5 * abs(want), (got[b], want)
        assert np.abs(g[b] - wgrad).max() <= 1e-4


def _asg_transitions(gtn, N, w):
    """examples/asg.cpp:36-47 / gtn_amd/criteria/asg_criterion.h: arc i = start -> label i, arc N + i N + j = j -> i"""
    g = gtn.Graph()
    g.add_nodes(np.array([1] + [0] * N, np.uint8), np.array([0] + [1] * N, np.uint8))
    n = np.arange(N)
    src = np.concatenate([np.zeros(N, np.int32), np.tile(n + 1, N).astype(np.int32)])
    dst = np.concatenate([n + 1, np.repeat(n + 1, N)]).astype(np.int32)
    lab = np.concatenate([n, np.repeat(n, N)]).astype(np.int32)
    g.add_arcs(src, dst, lab, lab, w.astype(np.float32))
    g.arc_sort()
    return g


def test_batch_asg_criterion_vs_per_graph_path(gtn):
    """the ASG criterion of examples/asg.cpp:59-68 on batch records -- force-alignment acceptors composed with the
    transitions built on the device, the full-connect term through the per-graph functions on the elements, the two
    mixed in one batch expression -- against the same criterion written with the per-graph functions: losses,
    emission gradients and the (batch-summed) transitions gradient"""
    import torch
    B, T, N = 5, 40, 7
    rng = np.random.default_rng(9)
    em = (rng.random((B, T, N), dtype=np.float32) * 6 - 3).astype(np.float32)
    tw = rng.normal(0, 1, N + N * N).astype(np.float32)
    tg = [rng.integers(0, N, size=int(rng.integers(1, 9))).astype(np.int32) for _ in range(B)]
    tg[2] = np.array([3, 3, 3], np.int32)
    em_dev = _dev(em)
    res = {}
    for batch in (True, False):
        trans = _asg_transitions(gtn, N, tw)
        prev = gtn.compose_mode(2)
        try:
            if batch:
                ems = gtn.Batch.linear(B, T, N, em_dev, True, True)
                fcc = gtn.forward_score(gtn.compose(ems, gtn.Batch([trans])))
                fals = gtn.Batch.asg_force_align(tg, trans, N)
                loss = gtn.subtract(fcc, gtn.forward_score(gtn.compose(ems, fals)))
                gtn.backward(loss)
                lv = loss.items()
                ge = np.stack([ems[b].grad().weights_to_numpy().reshape(T, N) for b in range(B)])
            else:
                ems = gtn.linear_graph_n(B, T, N, em_dev)
                fals = []
                for t in tg:
                    f = gtn.Graph(False)
                    f.add_node(True, len(t) == 0)
                    for l in range(1, len(t) + 1):
                        f.add_node(False, l == len(t))
                        f.add_arc(l - 1, l, int(t[l - 1]))
                        f.add_arc(l, l, int(t[l - 1]))
                    fals.append(f)
                fcc = gtn.forward_score(gtn.compose(ems, [trans]))
                loss = gtn.subtract(fcc, gtn.forward_score(gtn.compose(ems, gtn.compose(fals, [trans]))))
                gtn.backward(loss)
                lv = gtn.items(loss)
                ge = np.stack([ems[b].grad().weights_to_numpy().reshape(T, N) for b in range(B)])
        finally:
            gtn.compose_mode(prev)
        res[batch] = (lv, ge, trans.grad().weights_to_numpy().copy())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(res[True][2], res[False][2], rtol=2e-4, atol=5e-5)
    assert np.all(res[True][0] >= -1e-4)  # a loss is a -log probability ratio


def test_batch_ctc_c5_shape_vs_fp64(gtn):
    """BASELINE config 5's shape through the batch path (T = 2000, C = 1024, U = 200: two nodes per lane, one
    workgroup per CU, 32 staging registers per stream): losses and emission gradients against float64"""
    import torch
    from ctc_fp64 import ctc_loss_fp64
    B, T, C, U = 4, 2000, 1024, 200
    em, tg = gg.ctc_inputs(5, B, T, C, U)
    em_dev = _dev(em)
    grad = torch.empty(B, T, C, device="cuda:0")
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, list(tg), T, C, bind=grad)
    gtn.backward(loss)
    got = loss.items()
    g = grad.cpu().numpy()
    for b in range(2):
        want, wgrad, _ = ctc_loss_fp64(em[b], tg[b])
        assert abs(got[b] - want) <= 1e-5 * abs(want), (got[b], want)
        assert np.abs(g[b] - wgrad).max() <= 2e-4
    assert grad.sum(dim=2).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,T,C", [(256, 150, 32), (5, 37, 7), (9, 1, 1030)])
def test_batch_linear_forward_score_is_one_record(gtn, B, T, C):
    """BASELINE config C2: forwardScore of B linear chains over one tensor through gtn::Batch runs as ONE record (no
    element graphs: batch.cpp batch_shortest_distance, LINEAR) -- scores against float64 row log-sum-exps, the gradient
    (softmax of every row, creations.cpp:20-33 + shortest.cpp:33-62 on a chain) against float64 and against the
    per-graph vector path; odd alphabets (scalar rows) and C > 1024 (the split kernel) included"""
    import torch
    rng = np.random.default_rng(B * 7 + C)
    em = (rng.random((B, T, C), dtype=np.float32) * 10 - 5).astype(np.float32)
    em_dev = _dev(em)
    ems = gtn.Batch.linear(B, T, C, em_dev, True, True)
    gtn.prof_reset()
    gtn.prof_enable(True)
    fs = gtn.forward_score(ems)
    gtn.prof_enable(False)
    assert "linear_forward" in gtn.prof_names()
    grad = torch.full((B, T, C), float("nan"), device="cuda:0")
    ems.bind_grads(grad, np.arange(B, dtype=np.int64) * T * C)
    gtn.backward(fs)
    got = fs.items()
    ems.grads_to_device(grad, np.arange(B, dtype=np.int64) * T * C)
    g = grad.cpu().numpy()
    e64 = em.astype(np.float64)
    lse = np.logaddexp.reduce(e64, axis=2)
    np.testing.assert_allclose(got, lse.sum(axis=1), rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(g, np.exp(e64 - lse[:, :, None]), rtol=1e-5, atol=2e-6)
    # the per-graph vector path on the same tensor
    gs = gtn.linear_graph_n(B, T, C, em_dev)
    fv = gtn.forward_score(gs)
    gtn.backward(fv)
    np.testing.assert_allclose(got, gtn.items(fv), rtol=1e-6, atol=1e-5)
    k = B // 2
    np.testing.assert_allclose(g[k], gs[k].grad().weights_to_numpy().reshape(T, C), rtol=1e-6, atol=1e-7)


def _ctc_arc_counts(tg):
    """arcs of benchmarks/ctc.cpp:40-58's graph per target: a self loop per node, a step per node but the first, a
    skip into a label node whose previous label differs"""
    return [(2 * len(t) + 1) + 2 * len(t) + sum(1 for i in range(1, len(t)) if t[i] != t[i - 1]) for t in tg]


TARGET_GRAD_TOL = 1e-3


def test_timed_route_at_full_size_against_the_reference(gtn):
    """The route bench.py TIMES -- criteria::ctcLossBatch itself, through libgtn_criteria.so: gtn::Batch::ctcTargets
    (ctc_targets_kernel) -> Batch::linear -> batched intersect / forwardScore x2 / subtract / backward (the band sweeps)
    -- at BASELINE config C3's utterance shape (T = 1000, C = 256, U <= 100), 32 utterances with RAGGED targets
    (an empty one, repeated labels, a full-length one), against the UNMODIFIED reference running
    benchmarks/ctc.cpp:40-58,150-160 on the same inputs (oracle/_ref: ref_ctc_ragged): losses, emission gradients
    AND the target graphs' arc gradients (criterion_test.cpp:56-180 checks emissions only; the reference's
    benchmark builds its targets with calcGrad = true, so the timed step computes them)."""
    import ctypes as C
    import os
    import torch
    from ctc_fp64 import ctc_loss_fp64
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_path = os.path.join(root, "oracle", "_ref", "libgtn_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/libgtn_ref.so not built (needs /root/reference at build time)")
    B, T, Cn, Umax = 32, 1000, 256, 100
    em, tg = _ragged_inputs(77, B, T, Cn, Umax, Umin=1)
    tg[0] = tg[0][:0]                                            # empty target
    tg[1] = np.full(Umax, 7, np.int32)                           # one label repeated: no skip arcs at all
    tg[2] = np.random.default_rng(5).integers(1, Cn, size=Umax).astype(np.int32)  # full length
    tg[3] = np.repeat(np.arange(1, 26, dtype=np.int32), 4)       # runs of repeats
    lens = np.array([len(t) for t in tg], np.int32)
    flat = np.concatenate(tg).astype(np.int32)
    narcs = _ctc_arc_counts(tg)
    toff = np.concatenate([[0], np.cumsum(narcs)]).astype(np.int64)

    # ---- the product: libgtn_criteria.so
    lib = C.CDLL(os.path.join(root, "gtn_amd", "lib", "libgtn_criteria.so"))
    lib.gtn_ctc_loss_target_grads_n.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 4
    lib.gtn_ctc_loss_target_grads_n.restype = C.c_int
    lib.gtn_criteria_last_error.restype = C.c_char_p
    em_dev = _dev(em)
    loss_dev = torch.full((B,), float("nan"), device="cuda:0")
    grad_dev = torch.full((B, T, Cn), float("nan"), device="cuda:0")
    tgrad_dev = torch.full((int(toff[-1]),), float("nan"), device="cuda:0")
    torch.cuda.synchronize()
    rc = lib.gtn_ctc_loss_target_grads_n(em_dev.data_ptr(), flat.ctypes.data, lens.ctypes.data, B, T, Cn, 0,
                                         loss_dev.data_ptr(), grad_dev.data_ptr(), tgrad_dev.data_ptr(), toff.ctypes.data)
    assert rc == 0, lib.gtn_criteria_last_error().decode()
    gtn.synchronize()
    got_l, got_g, got_t = loss_dev.cpu().numpy(), grad_dev.cpu().numpy(), tgrad_dev.cpu().numpy()

    # ---- the checker: the reference itself
    ref = C.CDLL(ref_path)
    ref.ref_ctc_ragged.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 5
    ref.ref_ctc_ragged.restype = C.c_int
    want_l = np.zeros(B, np.float32)
    want_g = np.zeros((B, T, Cn), np.float32)
    want_t = np.zeros(int(toff[-1]), np.float32)
    arcs = np.zeros(B, np.int32)
    ref.ref_ctc_ragged(em.ctypes.data, flat.ctypes.data, lens.ctypes.data, B, T, Cn, 0, want_l.ctypes.data,
                       want_g.ctypes.data, want_t.ctypes.data, toff.ctypes.data, arcs.ctypes.data)
    assert arcs.tolist() == narcs

    # losses: 1e-4 relative (north_star), measured ~1e-6
    assert np.all(np.isfinite(got_l))
    np.testing.assert_allclose(got_l, want_l, rtol=1e-4)
    # gradients.  The float32 reference keeps unnormalised scores ~8.5 T, so ITS gradients are ~1e-3 from exact
    # arithmetic at T = 1000 (measured below, per utterance); tolerance: <= 1e-4 against float64 (posteriors in
    # [-1, 1]; target-arc gradients, sums of T posteriors, relative to max(1, |g|)) AND <= 1e-2 against the reference
    worst = {"em64": 0.0, "emref": 0.0, "t64": 0.0, "tref": 0.0, "ref_em64": 0.0, "ref_t64": 0.0}
    for b in range(B):
        l64, g64, t64 = ctc_loss_fp64(em[b], tg[b])
        assert abs(got_l[b] - l64) <= 1e-4 * abs(l64)
        sl = slice(int(toff[b]), int(toff[b + 1]))
        scale = np.maximum(1.0, np.abs(t64))
        worst["em64"] = max(worst["em64"], np.abs(got_g[b] - g64).max())
        worst["emref"] = max(worst["emref"], np.abs(got_g[b] - want_g[b]).max())
        worst["ref_em64"] = max(worst["ref_em64"], np.abs(want_g[b] - g64).max())
        worst["t64"] = max(worst["t64"], (np.abs(got_t[sl] - t64) / scale).max())
        worst["tref"] = max(worst["tref"], (np.abs(got_t[sl] - want_t[sl]) / scale).max())
        worst["ref_t64"] = max(worst["ref_t64"], (np.abs(want_t[sl] - t64) / scale).max())
    print("timed route vs float64 / reference:", {k: float(v) for k, v in worst.items()})
    assert worst["em64"] <= 1e-4, worst
    # (target-arc gradients are sums of T arc posteriors accumulated in float32 registers by the sweep and are NOT
    #  rescaled per row as the emission gradients are: measured 3.9e-4 on the first run of this test, the reference's
    #  own 2.0e-3)
    assert worst["t64"] <= TARGET_GRAD_TOL, worst
    assert worst["emref"] <= 1e-2 and worst["tref"] <= 1e-2, worst
    # (and the engine is the closer of the two to exact arithmetic)
    assert worst["em64"] <= worst["ref_em64"] and worst["t64"] <= worst["ref_t64"], worst


@pytest.mark.parametrize("passes", [2, 3])
def test_repeated_backward_over_a_retained_tape_matches_the_reference(gtn, passes):
    """backward(loss, retain) called two and three times: in the reference every node passes its ACCUMULATED gradient on
    in every pass (autograd.cpp:40-52) and compose's gradient function re-scatters the product's accumulated gradient
    (compose.cpp:496-518), so after k passes the emissions hold (1 + 3 + 6 ...) S - (1 + 4 + 10 ...) P: 4 S - 5 P after
    two, 10 S - 15 P after three.  The symbolic product has no gradient of its own; the batch records pushed the
    CURRENT gradient past it (4 S - 4 P) until round 6 (tools/double_backward_fit.py found it).  Against the UNMODIFIED
    reference (oracle/_ref behind tests/refbackend) through all three routes: batch records, the vector overloads with
    the products symbolic, and with the lattices built."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbackend"))
    try:
        import gtn_ref as ref
    except Exception as e:
        pytest.skip("needs oracle/_ref: %s" % e)
    B, T, C = 3, 25, 8
    em, tg = _ragged_inputs(3, B, T, C, 5, Umin=1)
    want = []
    for b in range(B):
        e = ref.linear_graph(T, C)
        e.set_weights(em[b].reshape(-1))
        c = gg.to_api(ref, gg.ctc_target_graph(tg[b].tolist()))
        c.arc_sort()
        l = ref.subtract(ref.forward_score(e), ref.forward_score(ref.intersect(c, e)))
        for p in range(passes):
            ref.backward(l, p < passes - 1)
        want.append(e.grad().weights_to_numpy().reshape(T, C))
    want = np.stack(want)
    em_dev = _dev(em)
    # batch records
    ctcs, ems, loss = _batch_ctc(gtn, em_dev, tg, T, C)
    for p in range(passes):
        gtn.backward(loss, p < passes - 1)
    g = torch.empty(B, T, C, device="cuda:0")
    ems.grads_to_device(g, np.arange(B, dtype=np.int64) * T * C)
    np.testing.assert_allclose(g.cpu().numpy(), want, rtol=2e-4, atol=2e-4)
    # vector overloads: products symbolic (2) and built (0)
    for mode in (2, 0):
        prev = gtn.compose_mode(mode)
        try:
            es = gtn.linear_graph_n(B, T, C, em_dev)
            cs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
            for c in cs:
                c.arc_sort()
            l = gtn.subtract(gtn.forward_score(es), gtn.forward_score(gtn.intersect(cs, es)))
            for p in range(passes):
                gtn.backward(l, p < passes - 1)
            got = np.stack([es[b].grad().weights_to_numpy().reshape(T, C) for b in range(B)])
        finally:
            gtn.compose_mode(prev)
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4, err_msg="compose mode %d" % mode)


def test_batch_route_random_differential_against_the_reference(gtn):
    """120 random small batches through the batch records (the route bench.py times) against the UNMODIFIED reference
    run per utterance (benchmarks/ctc.cpp:40-58,150-160): T from 1 to 11, alphabets of 4 to 12 labels, targets of 0 to
    7 labels -- so empty targets, infeasible ones (more labels than frames: the loss is +inf) and single-frame
    utterances all occur --, -inf emissions in a third of the batches, calcGrad switched off on either side.  Losses,
    emission gradients and target-arc gradients must agree wherever the reference's number is finite; where the
    reference has NaN (a lattice node no finite path enters poisons everything upstream of it, autograd_test.cpp:339-386)
    the symbolic route has the posteriors the NaN stands for -- the one stated difference."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbackend"))
    try:
        import gtn_ref as ref
    except Exception as e:
        pytest.skip("needs oracle/_ref: %s" % e)
    rng = np.random.default_rng(5)
    for trial in range(120):
        B, T, C = int(rng.integers(1, 5)), int(rng.integers(1, 12)), int(rng.choice([4, 5, 8, 12]))
        em = (rng.random((B, T, C), dtype=np.float32) * 8 - 4).astype(np.float32)
        if rng.random() < 0.3:
            em[rng.random(em.shape) < 0.15] = -np.inf
        tg = [rng.integers(1, C, size=int(rng.integers(0, 8))).astype(np.int32) for _ in range(B)]
        cg_t, cg_e = bool(rng.random() < 0.7), bool(rng.random() < 0.9)
        wl, wg, wt = [], [], []
        for b in range(B):
            e = ref.linear_graph(T, C, cg_e)
            e.set_weights(em[b].reshape(-1))
            c = gg.to_api(ref, gg.ctc_target_graph(tg[b].tolist()), cg_t)
            c.arc_sort()
            l = ref.subtract(ref.forward_score(e), ref.forward_score(ref.intersect(c, e)))
            if cg_e or cg_t:
                ref.backward(l)
            wl.append(np.float32(l.item()))
            wg.append(e.grad().weights_to_numpy().reshape(T, C) if cg_e else None)
            wt.append(c.grad().weights_to_numpy() if cg_t else None)
        em_dev = _dev(em)
        ctcs = gtn.Batch.ctc_targets(tg, 0, cg_t)
        ems = gtn.Batch.linear(B, T, C, em_dev, cg_e, True)
        loss = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(gtn.intersect(ctcs, ems)))
        if cg_e or cg_t:
            gtn.backward(loss)
        gl = np.array(loss.items(), np.float32)
        for b in range(B):
            a, w = gl[b], wl[b]
            assert (np.isinf(a) and np.isinf(w) and np.sign(a) == np.sign(w)) or (np.isnan(a) and np.isnan(w)) or \
                abs(a - w) <= 1e-4 * max(1.0, abs(w)), (trial, b, a, w)
        if cg_e:
            g = torch.empty(B, T, C, device="cuda:0")
            ems.grads_to_device(g, np.arange(B, dtype=np.int64) * T * C)
            g = g.cpu().numpy()
            for b in range(B):
                fin = ~np.isnan(wg[b])
                assert np.all(np.abs(g[b][fin] - wg[b][fin]) <= 2e-4), (trial, b, T, len(tg[b]))
        if cg_t:
            for b in range(B):
                a, w = np.asarray(ctcs[b].grad().weights_to_numpy()), wt[b]
                assert a.shape == w.shape
                fin = ~np.isnan(w)
                assert np.all(np.abs(a[fin] - w[fin]) <= 2e-4 * np.maximum(1.0, np.abs(w[fin]))), (trial, b, T, len(tg[b]))
