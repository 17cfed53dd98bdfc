import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, graphgen as gg
import gtn_amd as gtn
from ctc_fp64 import ctc_loss_fp64
B,T,C,U=2,1000,256,100
em,tg=gg.ctc_inputs(1234,B,T,C,U)
prev=gtn.compose_mode(2)
ems=gtn.linear_graph_n(B,T,C,torch.from_numpy(em).cuda())
ctcs=[gg.to_api(gtn, gg.ctc_target_graph(list(t))) for t in tg]
comp=gtn.intersect(ctcs,ems)
fs=gtn.forward_score(comp)
gtn.backward(fs)
z=gtn.items(fs)
for b in range(B):
    g=ems[b].grad().weights_to_numpy().reshape(T,C).astype(np.float64)   # = posterior occupancy per (t,c)
    l64,g64,_=ctc_loss_fp64(em[b],tg[b])
    sm=np.exp(em[b].astype(np.float64)); sm/=sm.sum(1,keepdims=True)
    occ64=sm-g64
    rs=g.sum(1)
    print("utt",b,"score",z[b],"row sums of posteriors: min %.6f max %.6f mean %.6f"%(rs.min(),rs.max(),rs.mean()), "max|err|",np.abs(g-occ64).max())
    gn=g/rs[:,None]
    print("   after per-row normalisation: max|err|",np.abs(gn-occ64).max())
    print("   row-sum error by time (every 100):", np.round((rs[::100]-1)*1e5,1))
gtn.compose_mode(prev)
