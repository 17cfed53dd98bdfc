import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gtn_amd as gtn
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
prev = gtn.compose_mode(int(os.environ.get("DBG_MODE", "1")))
def product():
    tr = transitions()
    ems = gtn.linear_graph_n(B, T, C, torch.from_numpy(em).cuda())
    fs = gtn.forward_score(gtn.compose(ems, [tr]))
    return tr, ems, fs
tr1, ems1, fs1 = product()
gtn.backward(fs1)
g1 = tr1.grad().weights_to_numpy().copy(); e1 = ems1[0].grad().weights_to_numpy().copy()
tr3, ems3, fs3 = product()
gtn.backward(fs3, retain_graph=True)
ga = tr3.grad().weights_to_numpy().copy()
gtn.backward(fs3, retain_graph=True)
gb = tr3.grad().weights_to_numpy().copy()
print("names", gtn.prof_names() if hasattr(gtn, "prof_names") else None)
print("after 1:", np.abs(ga - g1).max(), "after 2:", np.abs(gb - 2 * g1).max(), "max g1", np.abs(g1).max(), "sum", g1.sum(), ga.sum(), gb.sum())
print("em after 2:", np.abs(ems3[0].grad().weights_to_numpy() - 2 * e1).max())
