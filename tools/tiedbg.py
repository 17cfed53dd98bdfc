import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, gtn_amd as gtn, graphgen as gg
B,T,Cn,U=512,1000,256,100
em,tg=gg.ctc_inputs(1234,B,T,Cn,U); em_dev=torch.from_numpy(em).cuda()
ctcs=[gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
ems=gtn.linear_graph_n(B,T,Cn,em_dev)
gtn.compose_mode(2)
p=gtn.viterbi_path(gtn.intersect(ctcs,ems))
print(len(p))
