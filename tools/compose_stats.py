"""GTNX_COMPOSE_STATS of intersect(ctc_target, emissions) at C3's shape (diagnostic): phase times of graph 0"""
import os, sys, time
os.environ["GTNX_COMPOSE_STATS"] = "1"; os.environ["GTNX_SYNC_COMPOSE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gtn_amd as gtn
import graphgen as gg
T, Cn, U = 1000, 256, 100
for B in (1, 64, 512):
    em, tg = gg.ctc_inputs(1234, B, T, Cn, U)
    em_dev = torch.from_numpy(em).cuda()
    ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
    for g in ctcs: g.arc_sort()
    ems = gtn.linear_graph_n(B, T, Cn, em_dev)
    prev = gtn.compose_mode(0)
    for it in range(3):
        gtn.synchronize(); t0 = time.perf_counter()
        comp = gtn.intersect(ctcs, ems)
        gtn.synchronize()
        print("B", B, "wall ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
        del comp
    gtn.compose_mode(prev)
