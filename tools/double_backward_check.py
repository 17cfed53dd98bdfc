import os, sys
import numpy as np, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tests","refbackend"))
import gtn_amd as gtn, gtn_ref as ref, graphgen as gg
B,T,C=3,25,8
rng=np.random.default_rng(3)
em=(rng.random((B,T,C),dtype=np.float32)*10-5).astype(np.float32)
tg=[rng.integers(1,C,size=int(rng.integers(1,6))).astype(np.int32) for _ in range(B)]
# reference
want=[]
for b in range(B):
    e=ref.linear_graph(T,C); e.set_weights(em[b].reshape(-1))
    c=gg.to_api(ref, gg.ctc_target_graph(tg[b].tolist())); c.arc_sort()
    l=ref.subtract(ref.forward_score(e), ref.forward_score(ref.intersect(c,e)))
    ref.backward(l, True); ref.backward(l)
    want.append(e.grad().weights_to_numpy().reshape(T,C))
want=np.stack(want)
em_dev=torch.from_numpy(em).cuda()
def pergraph(mode):
    prev=gtn.compose_mode(mode)
    try:
        es=gtn.linear_graph_n(B,T,C,em_dev)
        cs=[gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
        for c in cs: c.arc_sort()
        l=gtn.subtract(gtn.forward_score(es), gtn.forward_score(gtn.intersect(cs,es)))
        gtn.backward(l, True); gtn.backward(l)
        return np.stack([es[b].grad().weights_to_numpy().reshape(T,C) for b in range(B)])
    finally: gtn.compose_mode(prev)
def batch():
    ctcs=gtn.Batch.ctc_targets(tg,0,True); ems=gtn.Batch.linear(B,T,C,em_dev,True,True)
    loss=gtn.subtract(gtn.forward_score(ems), gtn.forward_score(gtn.intersect(ctcs,ems)))
    gtn.backward(loss,True); gtn.backward(loss)
    g2=torch.empty(B,T,C,device="cuda:0"); ems.grads_to_device(g2, np.arange(B,dtype=np.int64)*T*C)
    return g2.cpu().numpy()
for name,fn in (("per-graph built (mode 0)",lambda:pergraph(0)),("per-graph symbolic (mode 2)",lambda:pergraph(2)),("batch records",batch)):
    got=fn(); print(name, "max |engine - reference| =", float(np.abs(got-want).max()))
