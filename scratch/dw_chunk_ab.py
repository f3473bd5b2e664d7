import torch, time
dev="cuda"; T=88892
gh=torch.randn(T,1024,device=dev).bfloat16(); x=torch.randn(T,256,device=dev).bfloat16(); gy=torch.randn(T,256,device=dev).bfloat16()
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for chunk in (313, 626, 1252, 2504, 5008):
    nb=T//chunk
    f1=lambda: torch.bmm(gh.view(nb,chunk,-1).transpose(1,2), x.view(nb,chunk,-1)).sum(0,dtype=torch.float32)
    f2=lambda: torch.bmm(gy.view(nb,chunk,-1).transpose(1,2), gh.view(nb,chunk,-1)).sum(0,dtype=torch.float32)
    print(f"chunk {chunk:5d} nb {nb:4d}: dW1 {timeit(f1):7.1f} us   dW2 {timeit(f2):7.1f} us")
print("plain mm: dW1 %.1f us  dW2 %.1f us" % (timeit(lambda: gh.t() @ x), timeit(lambda: gy.t() @ gh)))
