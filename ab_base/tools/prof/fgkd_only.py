import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
S=[(100,167),(50,84),(25,42),(13,21)]
Nv=sum(h*w for h,w in S)
def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
B=4
g=torch.Generator().manual_seed(0)
fs=[torch.randn(B,256,h,w,generator=g).cuda() for h,w in S]; ft=[f+0.3*torch.randn_like(f) for f in fs]
boxes=[]
for b in range(B):
    xy=torch.rand(10,2,generator=g)*torch.tensor([800.,480.]); sz=torch.rand(10,2,generator=g)*torch.tensor([460.,280.])+8
    boxes.append(torch.cat([xy,xy+sz],1).cuda())
N=B*300
hs_s=torch.randn(N,256,generator=g).cuda().requires_grad_(True); hs_t=(hs_s.detach()+0.1*torch.randn(N,256,generator=g).cuda())
labels=torch.full((N,),80)
for b in range(B): labels[b*300:b*300+10]=torch.randint(0,70,(10,),generator=g)
labels=labels.cuda(); keep=torch.cat([b*300+torch.randperm(300,generator=g)[:10] for b in range(B)]).cuda()
prev=torch.zeros(80,dtype=torch.bool); prev[:70]=True; prev=prev.cuda()
t=timeit(lambda: native.fgkd_loss(fs,ft,boxes,[(800,1333)]*B,hs_t,keep,hs_s,labels,prev,2.0,1.0))
byts=B*2*Nv*256*4
v=native.fgkd_loss(fs,ft,boxes,[(800,1333)]*B,hs_t,keep,hs_s,labels,prev,2.0,1.0)
print(f"fgkd B={B}: {t:.1f} us (all launches) algo {byts/1e6:.1f} MB -> {byts/t/1e6:.2f} TB/s  value {float(v):.6f}")
