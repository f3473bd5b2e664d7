import sys; sys.path.insert(0,'.')
import torch, time
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
    return e0.elapsed_time(e1)/n*1e3  # us
def ref_points(B):
    pts=[]
    for (H,W) in S:
        ys,xs=torch.meshgrid(torch.linspace(0.5,H-0.5,H)/H, torch.linspace(0.5,W-0.5,W)/W, indexing='ij')
        pts.append(torch.stack([xs.reshape(-1),ys.reshape(-1)],-1))
    return torch.cat(pts,0)[None].expand(B,-1,-1)
for B in (1,4):
  for dt in (torch.float32, torch.bfloat16):
    g=torch.Generator().manual_seed(0)
    value=torch.randn(B,Nv,8,32,generator=g).to(dt).cuda()
    ref=ref_points(B)  # [B,Nv,2]
    # realistic offsets: init-style grid, few pixels
    off=torch.randn(B,Nv,8,4,4,2,generator=g)*2.0
    norm=torch.tensor([[w,h] for h,w in S],dtype=torch.float32).view(1,1,1,4,1,2)
    loc=(ref[:,:,None,None,None,:]+off/norm).cuda().contiguous()
    attn=torch.softmax(torch.randn(B,Nv,8,16,generator=g),-1).view(B,Nv,8,4,4).cuda()
    go=torch.randn(B,Nv,256,generator=g).to(dt).cuda()
    t=timeit(lambda: native.msda_forward_raw(value,S,loc,attn))
    esz=4 if dt==torch.float32 else 2
    byts=B*(Nv*256*esz + Nv*1024 + Nv*512 + Nv*256*esz)
    print(f"enc fwd B={B} {dt}: {t:.1f} us  algo {byts/1e6:.1f} MB -> {byts/t/1e6:.2f} TB/s ({byts/t/1e6/8*100:.1f}% of 8TB/s)")
    t=timeit(lambda: native.msda_backward_raw(value,S,loc,attn,go), n=5, w=1)
    bb=B*(Nv*256*esz + Nv*1024 + Nv*512 + Nv*256*esz + Nv*256*4 + Nv*1024+Nv*512)
    print(f"enc bwd B={B} {dt}: {t:.1f} us  algo {bb/1e6:.1f} MB -> {bb/t/1e6:.2f} TB/s")
    # decoder
    locd=torch.rand(B,300,8,4,4,2,generator=g).cuda(); attd=torch.softmax(torch.randn(B,300,8,16,generator=g),-1).view(B,300,8,4,4).cuda()
    t=timeit(lambda: native.msda_forward_raw(value,S,locd,attd))
    print(f"dec fwd B={B} {dt}: {t:.1f} us")
# fgkd
B=4
g=torch.Generator().manual_seed(0)
fs=[torch.randn(B,256,h,w,generator=g).cuda() for h,w in S]; ft=[f+0.3*torch.randn_like(f) for f in fs]
boxes=[]
for b in range(B):
    xy=torch.rand(10,2,generator=g)*torch.tensor([800.,480.]); sz=torch.rand(10,2,generator=g)*torch.tensor([460.,280.])+8
    boxes.append(torch.cat([xy,xy+sz],1).cuda())
N=B*300
hs_s=torch.randn(N,256,generator=g).cuda().requires_grad_(True); hs_t=(hs_s.detach()+0.1*torch.randn(N,256,generator=g).cuda())
labels=torch.full((N,),80); 
for b in range(B): labels[b*300:b*300+10]=torch.randint(0,70,(10,),generator=g)
labels=labels.cuda(); keep=torch.cat([b*300+torch.randperm(300,generator=g)[:10] for b in range(B)]).cuda()
labt=torch.randint(0,70,(40,),generator=g).cuda()
prev=torch.zeros(80,dtype=torch.bool); prev[:70]=True; prev=prev.cuda()
t=timeit(lambda: native.fgkd_loss(fs,ft,boxes,[(800,1333)]*B,hs_t,keep,hs_s,labels,prev,2.0,1.0))
byts=B*2*Nv*256*4
print(f"fgkd B={B}: {t:.1f} us algo {byts/1e6:.1f} MB -> {byts/t/1e6:.2f} TB/s")
t=timeit(lambda: native.proto_corr_loss(hs_s,labels,prev,hs_t,keep,labt,70,1.0))
print(f"proto_corr B={B} L=70: {t:.1f} us")
import numpy as np
mats=[torch.rand(300,17) for _ in range(24)]
flat=torch.cat([m.reshape(-1) for m in mats]).cuda()
offs=list(np.cumsum([0]+[m.numel() for m in mats])[:-1])
t=timeit(lambda: native.lsap_batched(flat,[300]*24,[17]*24,offs))
print(f"lsap 24x(300x17): {t:.1f} us")
mats=[torch.rand(300,110) for _ in range(24)]
flat=torch.cat([m.reshape(-1) for m in mats]).cuda()
offs=list(np.cumsum([0]+[m.numel() for m in mats])[:-1])
t=timeit(lambda: native.lsap_batched(flat,[300]*24,[110]*24,offs))
print(f"lsap 24x(300x110): {t:.1f} us")
