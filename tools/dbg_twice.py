import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import gtn_amd as gtn
import graphgen as gg
from test_batch_gpu import _ragged_inputs, _dev
B, T, C = 3, 25, 8
em, tg = _ragged_inputs(3, B, T, C, 5, Umin=1)
em_dev = _dev(em)
sm = torch.softmax(torch.from_numpy(em), -1).numpy()
def coef(g):  # coefficient of the softmax term at a column whose label is not in the target
    b = 0
    cols = [c for c in range(1, C) if c not in tg[0].tolist()]
    c = cols[0]
    return float(g[0 * T * C:(1) * T * C].reshape(T, C)[3, c] / sm[0, 3, c])
for order in ("norm_first", "int_first"):
  for second in (False, True):
    ref_t = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
    ref_e = gtn.linear_graph_n(B, T, C, em_dev)
    prev = gtn.compose_mode(2)
    if order == "norm_first":
        n = gtn.forward_score(ref_e); s = gtn.forward_score(gtn.intersect(ref_t, ref_e))
    else:
        s = gtn.forward_score(gtn.intersect(ref_t, ref_e)); n = gtn.forward_score(ref_e)
    ref_l = gtn.subtract(n, s)
    gtn.compose_mode(prev)
    gtn.backward(ref_l, True)
    if second: gtn.backward(ref_l)
    print(order, "second" if second else "first", "softmax coefficient", coef(ref_e[0].grad().weights_to_numpy()))
