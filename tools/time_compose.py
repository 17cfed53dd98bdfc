import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import gtn_amd as gtn, graphgen as gg
from bench import build_ctc_graphs
B, T, C, U = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 1000, 256, 100
em, tg = gg.ctc_inputs(1, B, T, C, U)
dev = torch.from_numpy(em).cuda()
ctcs = build_ctc_graphs(gtn, tg)
gtn.prof_enable(True)
for it in range(3):
    ems = gtn.linear_graph_n(B, T, C, dev)
    try:
        comp = gtn.intersect(ctcs, ems)
    except Exception as e:
        print("err", e)
    gtn.synchronize()
print({n: round(gtn.prof_get(n)["total_ms"] / 3, 2) for n in gtn.prof_names()})
