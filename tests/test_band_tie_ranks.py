"""How exact ties of CTC products are decided without the lattice (gtn_amd/csrc/ops_band.cpp: tie_ranks): the engine's
two node orders of a CTC target -- the reference's QUEUE order (viterbiPath keeps the arc relaxed first,
shortest.cpp:208-224) and compose's CREATION order (in-lists and accept list: viterbiScore's gradient,
shortest.cpp:118-127, :148-160) -- against the UNMODIFIED reference: intersect(ctc, emissions) is built by the
reference (oracle/_ref behind the same ABI, tests/refbackend), its queue is replayed on the lattice here, and in every
layer of the lattice both orders must be the engine's ranks restricted to the nodes alive in that layer.  Host only."""
import collections
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import graphgen as gg


@pytest.fixture(scope="module")
def apis():
    try:
        import gtn_amd as gtn
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refbackend"))
        import gtn_ref as ref
    except Exception as e:  # oracle/_ref is built by __graft_entry__.build() where /root/reference exists
        pytest.skip("needs libgtn_amd.so and oracle/_ref: %s" % e)
    return gtn, ref


def _layers_of_the_reference_lattice(ref, tgt, blank, T, C, ctc_first):
    ctc = gg.to_api(ref, gg.ctc_target_graph(tgt, blank))
    ctc.arc_sort()
    em = ref.linear_graph(T, C)
    comp = ref.intersect(ctc, em) if ctc_first else ref.intersect(em, ctc)
    N, A = comp.num_nodes(), comp.num_arcs()
    if N == 0:
        return None
    src, dst, lab, _, _ = comp.arcs()
    src, dst, lab = src.tolist(), dst.tolist(), lab.tolist()
    cs, cd, ci, _, _ = ctc.arcs()
    step = {(int(cs[a]), int(ci[a])): int(cd[a]) for a in range(ctc.num_arcs())}
    out = {n: comp.out(n) for n in range(N)}
    deg = [0] * N
    for a in range(A):
        deg[dst[a]] += 1
    where = {comp.start()[0]: (0, 0)}  # lattice node -> (time, target node)
    queue, popped = collections.deque(comp.start()), collections.defaultdict(list)
    while queue:  # shortest.cpp:208-224
        n = queue.popleft()
        t, l = where[n]
        popped[t].append(l)
        for a in out[n]:
            d = dst[a]
            where.setdefault(d, (t + 1, step[(l, lab[a])]))
            deg[d] -= 1
            if deg[d] == 0:
                queue.append(d)
    assert len(where) == N
    created = collections.defaultdict(list)
    for n in range(N):  # node ids = creation order
        created[where[n][0]].append(where[n][1])
    return popped, created


@pytest.mark.parametrize("T,C,U,repeat", [(30, 6, 5, 0.3), (60, 12, 10, 0.2), (25, 4, 8, 0.5), (21, 7, 10, 0.0),
                                          (40, 3, 12, 0.6)])
def test_node_ranks_are_the_references_orders_in_every_layer(apis, T, C, U, repeat):
    gtn, ref = apis
    checked = 0
    for seed in range(6):
        rng = np.random.default_rng(1000 * U + seed)
        blank = int(rng.integers(0, C))
        tgt = []
        for _ in range(U):
            if tgt and rng.random() < repeat:
                tgt.append(tgt[-1])
            else:
                tgt.append(int(rng.choice([x for x in range(C) if x != blank])))
        g = gg.to_api(gtn, gg.ctc_target_graph(tgt, blank))
        g.arc_sort()
        ranks = gtn.debug_tie_ranks(g)
        assert ranks is not None
        for ctc_first in (True, False):
            lay = _layers_of_the_reference_lattice(ref, tgt, blank, T, C, ctc_first)
            if lay is None:
                continue
            for rank, layers in zip(ranks, lay):
                for nodes in layers.values():
                    assert nodes == sorted(nodes, key=lambda l: rank[l])
            checked += 1
    assert checked > 0


def test_ranks_do_not_apply_to_other_graphs(apis):
    gtn, _ = apis
    g = gg.to_api(gtn, gg.ctc_target_graph([1, 2, 3]))
    g.arc_sort()
    assert gtn.debug_tie_ranks(g) is not None
    g.add_arc(0, 2, 5)  # no longer ctcGraph(labels)
    assert gtn.debug_tie_ranks(g) is None


_CLOSED_FORM_CHECK = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import gtn_amd as gtn
import graphgen as gg


def layer_order(t, last):
    # the recursion of gtn_amd/csrc/ops_band.cpp: layer_order, restated on ctcGraph(t) with blank 0 (benchmarks/ctc.cpp:40-58
    # after arcSort: out-lists by label)
    L = 2 * len(t) + 1
    out = [[] for _ in range(L)]
    a = 0
    for l in range(L):
        lab = t[(l - 1) // 2] if l %% 2 else 0
        out[l].append((lab, a, l)); a += 1
        if l > 0:
            out[l - 1].append((lab, a, l)); a += 1
        if l %% 2 and l > 1 and lab != t[(l - 1) // 2 - 1]:
            out[l - 2].append((lab, a, l)); a += 1
    for l in range(L):
        out[l].sort()
    layer = [0]
    for _ in range(4 * L + 8):
        key = {}
        for i, l in enumerate(layer):
            for k, (_, _, d) in enumerate(out[l]):
                kk = i * 8 + k
                key[d] = kk if d not in key else (max(key[d], kk) if last else min(key[d], kk))
        nxt = sorted(key, key=lambda d: key[d])
        if nxt == layer:
            rank = [0] * L
            for i, n in enumerate(layer):
                rank[n] = i
            return rank
        layer = nxt
    raise AssertionError("no fixed point")


rng = np.random.default_rng(11)
n = 0
for trial in range(%(trials)d):
    U = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 100]))
    C = int(rng.choice([2, 3, 5, 28, 256]))
    rep = float(rng.choice([0.0, 0.3, 0.7]))
    t = []
    for _ in range(U):
        t.append(t[-1] if t and rng.random() < rep else int(rng.integers(1, C)))
    g = gg.to_api(gtn, gg.ctc_target_graph(t, 0))
    g.arc_sort()
    ranks = gtn.debug_tie_ranks(g)   # (GTNX_CHECK_CLOSED_RANKS=1: the engine compares its closed form with its own recursion)
    assert ranks is not None, t
    assert ranks[0] == layer_order(t, True), t
    assert ranks[1] == layer_order(t, False) == list(range(2 * U + 1)), t
    n += 1
print("CLOSED_FORM_OK", n)
"""


def test_closed_form_ranks_equal_the_recursions_fixed_point():
    """Round 6: for targets whose blank is below every label the two tie orders come from a closed form
    (ops_band.cpp: tie_ranks) -- here against the recursion it replaces, twice over: inside the engine
    (GTNX_CHECK_CLOSED_RANKS=1 computes both and throws on a difference) and against a Python restatement of the
    recursion, on random targets up to U = 100 with alphabets from 2 to 256 labels and up to 70 %% repeats."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "gtn_amd", "lib", "libgtn_amd.so")):
        pytest.skip("libgtn_amd.so not built")
    env = dict(os.environ, GTNX_CHECK_CLOSED_RANKS="1")
    r = subprocess.run([sys.executable, "-c", _CLOSED_FORM_CHECK % {"root": root, "trials": 400}], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "CLOSED_FORM_OK 400" in r.stdout, (r.stdout + r.stderr)[-3000:]
    # ... and with the closed form switched off the same answers come from the recursion
    env = dict(os.environ, GTNX_NO_CLOSED_RANKS="1")
    r = subprocess.run([sys.executable, "-c", _CLOSED_FORM_CHECK % {"root": root, "trials": 60}], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "CLOSED_FORM_OK 60" in r.stdout, (r.stdout + r.stderr)[-3000:]
