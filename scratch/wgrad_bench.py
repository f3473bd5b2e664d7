import torch, time
def timeit(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
K=88892
for (M,N) in [(256,256),(256,1024),(1024,256),(256,384),(128,256)]:
    x=torch.randn(K,M,device='cuda',dtype=torch.bfloat16); g=torch.randn(K,N,device='cuda',dtype=torch.bfloat16)
    t0=timeit(lambda: x.t() @ g)
    ref=(x.t().float() @ g.float())
    res={}
    for chunk in (313, 626, 1252):
        nb=K//chunk
        f=lambda: torch.bmm(x.view(nb,chunk,M).transpose(1,2), g.view(nb,chunk,N)).sum(0)
        res[chunk]=timeit(f)
    f32=lambda: torch.bmm(x.view(71,1252,M).transpose(1,2), g.view(71,1252,N)).float().sum(0)
    out=f32(); err=((out-ref).abs().max()/ref.abs().max()).item()
    out0=(x.t()@g).float(); err0=((out0-ref).abs().max()/ref.abs().max()).item()
    print(f"M={M} N={N}: mm {t0:.0f} us (relerr {err0:.1e}); chunked bmm+sum:", {k:round(v) for k,v in res.items()}, f"relerr {err:.1e}", f"flops {2*M*N*K/1e9:.1f}G")
