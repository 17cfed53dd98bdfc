import os, sys
import numpy as np, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tests","refbackend"))
import gtn_amd as gtn, gtn_ref as ref, graphgen as gg
B,T,C=3,6,5
rng=np.random.default_rng(3)
em=(rng.random((B,T,C),dtype=np.float32)*4-2).astype(np.float32)
tg=[np.array([1,2],np.int32),np.array([3],np.int32),np.array([2,2,1],np.int32)]
em_dev=torch.from_numpy(em).cuda()
UTT=1
def run(api, mode, passes, which):
    prev = gtn.compose_mode(mode) if api is gtn else None
    try:
        if api is gtn: es=gtn.linear_graph_n(B,T,C,em_dev)
        else:
            es=[]
            for b in range(B):
                e=ref.linear_graph(T,C); e.set_weights(em[b].reshape(-1)); es.append(e)
        cs=[gg.to_api(api, gg.ctc_target_graph(t.tolist())) for t in tg]
        for c in cs: c.arc_sort()
        if which=="loss": l=api.subtract(api.forward_score(es), api.forward_score(api.intersect(cs,es)))
        elif which=="norm": l=api.forward_score(es)
        elif which=="score": l=api.forward_score(api.intersect(cs,es))
        elif which=="negscore": l=api.negate(api.forward_score(api.intersect(cs,es)))
        for p in range(passes): api.backward(l, p<passes-1)
        return es[UTT].grad().weights_to_numpy().reshape(T,C)
    finally:
        if api is gtn: gtn.compose_mode(prev)
S=run(ref,0,1,"norm"); P=run(ref,0,1,"score")
def fit(g):
    A=np.stack([S.ravel(),P.ravel()],1); x,res,_,_=np.linalg.lstsq(A,g.ravel(),rcond=None); return np.round(x,3)
for which in ("loss","score","negscore","norm"):
    for passes in (1,2,3):
        print(which, passes, "ref", fit(run(ref,0,passes,which)), "built", fit(run(gtn,0,passes,which)), "symbolic", fit(run(gtn,2,passes,which)))

def batch(passes):
    ctcs=gtn.Batch.ctc_targets(tg,0,True); ems=gtn.Batch.linear(B,T,C,em_dev,True,True)
    loss=gtn.subtract(gtn.forward_score(ems), gtn.forward_score(gtn.intersect(ctcs,ems)))
    for p in range(passes): gtn.backward(loss, p<passes-1)
    g2=torch.empty(B,T,C,device="cuda:0"); ems.grads_to_device(g2, np.arange(B,dtype=np.int64)*T*C)
    return g2.cpu().numpy()[UTT]
for passes in (1,2,3): print("batch records loss", passes, fit(batch(passes)))
