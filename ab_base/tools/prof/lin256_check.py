import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dskd_amd import native
dev="cuda"; T=88892
x=torch.randn(T,256,device=dev).bfloat16()
def timeit(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for N in (256, 384):
    w=(torch.randn(N,256,device=dev)/16).bfloat16(); b=(torch.randn(N,device=dev)*0.1).bfloat16()
    pk=native.lin256_pack(w)
    y=native.lin256(x,pk,N,b); ref=torch.addmm(b,x,w.t())
    print(N, "max rel err vs addmm", float((y.float()-ref.float()).abs().max()/ref.float().abs().max()),
          " us lin256 %.1f (pack %.1f)  addmm %.1f" % (timeit(lambda: native.lin256(x,pk,N,b)), timeit(lambda: native.lin256_pack(w)), timeit(lambda: torch.addmm(b,x,w.t()))))
