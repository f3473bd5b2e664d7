import sys; sys.path.insert(0,'.')
import torch, numpy as np
from dskd_amd import native
from oracle import msda_ref
S=[(12,17),(6,9),(3,5),(2,3)]
g=torch.Generator().manual_seed(0)
Nv=sum(h*w for h,w in S)
value=torch.randn(1,Nv,8,32,generator=g)
loc=torch.rand(1,1,8,4,4,2,generator=g)*0.5+0.25
attn=torch.softmax(torch.randn(1,1,8,16,generator=g),-1).view(1,1,8,4,4)
ref=msda_ref.msda_grid_sample(value,S,loc,attn)
out=native.msda_forward_raw(value.cuda(),S,loc.cuda(),attn.cuda()).cpu()
err=(out-ref).abs().view(8,32)
print("per-head max err",err.max(1).values)
print("per-lane-part err (head0)",err[0])
# single point tests: only one (h,l,p) has weight
for l in range(4):
  a2=torch.zeros_like(attn); a2[0,0,:,l,0]=1
  ref=msda_ref.msda_grid_sample(value,S,loc,a2); out=native.msda_forward_raw(value.cuda(),S,loc.cuda(),a2.cuda()).cpu()
  print("level",l,"err per head",(out-ref).abs().view(8,32).max(1).values)
