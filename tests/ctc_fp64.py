"""CTC loss and gradients in float64 (TEST INFRASTRUCTURE ONLY) -- the yardstick for float32 error.

What it computes is what benchmarks/ctc.cpp:150-160 computes: forwardScore(emissions) -
forwardScore(intersect(ctcGraph(target), emissions)) and d loss / d emissions, i.e.
shortest.cpp:86-170 / :33-62 over the product compose.cpp:377-522 builds -- here as the alpha-beta
recursion over (time, node of the target graph) in float64 with numpy, so that the float32 results
of the reference, of the C oracle and of the HIP kernels can each be measured against (practically)
exact arithmetic.  The untrimmed recursion visits the dead states the reference trims; they carry
zero posterior mass, so every score and gradient is the same.
"""
import numpy as np


def _lse(a, axis=None):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        return (np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m).squeeze(axis)


def ctc_loss_fp64(emissions, target, blank=0):
    """-> (loss, grad[T, C], target_arc_grad or None): float64"""
    em = np.asarray(emissions, dtype=np.float64)
    T, C = em.shape
    tg = np.asarray(target, dtype=np.int64)
    U = tg.size
    S = 2 * U + 1
    lab = np.full(S, blank, dtype=np.int64)
    lab[1::2] = tg
    skip = np.zeros(S, dtype=bool)  # arc from s-2 (label nodes whose previous label differs)
    for s in range(3, S, 2):
        skip[s] = lab[s] != lab[s - 2]
    NEG = -np.inf
    alpha = np.full((T + 1, S), NEG)
    alpha[0, 0] = 0.0
    e = em[:, lab]  # [T, S]
    for t in range(T):
        a = alpha[t]
        x1 = np.concatenate([[NEG], a])[:S]
        x2 = np.where(skip, np.concatenate([[NEG, NEG], a])[:S], NEG)
        alpha[t + 1] = _lse(np.stack([a, x1, x2]), axis=0) + e[t]
    acc = [S - 1] if S == 1 else [S - 1, S - 2]
    z = _lse(alpha[T, acc])
    norm_rows = _lse(em, axis=1)
    loss = float(np.sum(norm_rows) - z)
    grad = np.exp(em - norm_rows[:, None])  # d forwardScore(emissions)
    if not np.isfinite(z):
        return loss, grad, None
    beta = np.full((T + 1, S), NEG)
    beta[T, acc] = 0.0
    skip_out = np.concatenate([skip, [False, False]])[2:S + 2]  # arc s -> s+2
    for t in range(T - 1, -1, -1):
        q = e[t] + beta[t + 1]
        y1 = np.concatenate([q, [NEG]])[1:S + 1]
        y2 = np.where(skip_out, np.concatenate([q, [NEG, NEG]])[2:S + 2], NEG)
        beta[t] = _lse(np.stack([q, y1, y2]), axis=0)
    occ = np.exp(alpha[1:] + beta[1:] - z)  # node posteriors after consuming frame t
    for s in range(S):
        grad[:, lab[s]] -= occ[:, s]
    # d loss / d (weights of the target graph's arcs), arc ids in benchmarks/ctc.cpp:40-58's addArc order: per node
    # its self loop, the arc from s-1 (s > 0), the arc from s-2 (skip[s]); an arc's posterior summed over the frames
    q = e + beta[1:]  # [T, S]: consume frame t into node s, then finish
    tgrad = []
    for s in range(S):
        for k in (0, 1, 2):
            if (k == 1 and s == 0) or (k == 2 and not skip[s]):
                continue
            tgrad.append(-np.sum(np.exp(alpha[:T, s - k] + q[:, s] - z)))
    return loss, grad, np.asarray(tgrad)


def asg_fp64(em, tw):
    """float64 restatement of the ASG full-connect term on the dense transitions graph of
    tests/golden/make_golden_c4.py: transitions (start arcs tw[:C], arc j -> i at tw[C + i * C + j]): score, d/d emissions,
    d/d transitions (same layout as tw), and the max-plus optimum with its labels"""
    T, C = em.shape
    em = em.astype(np.float64)
    st, W = tw[:C].astype(np.float64), tw[C:].astype(np.float64).reshape(C, C)  # W[i][j]: j -> i
    lse = lambda x, ax: (lambda m: m + np.log(np.exp(x - np.expand_dims(m, ax)).sum(ax)))(x.max(ax))
    alpha = np.zeros((T + 1, C))
    alpha[1] = st + em[0]
    for t in range(1, T):
        alpha[t + 1] = em[t] + lse(alpha[t][None, :] + W, 1)
    Z = lse(alpha[T], 0)
    beta = np.zeros((T + 1, C))
    for t in range(T - 1, 0, -1):
        beta[t] = lse(W + (em[t] + beta[t + 1])[:, None], 0)
    g_em = np.exp(alpha[1:] + beta[1:] - Z)
    g_st = np.exp(st + em[0] + beta[1] - Z)
    g_W = np.zeros((C, C))
    for t in range(1, T):
        g_W += np.exp(alpha[t][None, :] + W + (em[t] + beta[t + 1])[:, None] - Z)
    v = st + em[0]
    back = []
    for t in range(1, T):
        cand = v[None, :] + W
        back.append(cand.argmax(1))
        v = em[t] + cand.max(1)
    lab = [int(v.argmax())]
    for bp in reversed(back):
        lab.append(int(bp[lab[-1]]))
    return Z, g_em, np.concatenate([g_st, g_W.reshape(-1)]), float(v.max()), lab[::-1]
