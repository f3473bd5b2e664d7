import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dskd_amd import native
dev="cuda"
def timeit(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
gn=torch.nn.GroupNorm(32,256).to(dev)
for (H,W) in ((100,167),(50,84),(25,42),(13,21)):
    x=torch.randn(4,256,H,W,device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    up=torch.randn_like(x)
    def own():
        y=native.group_norm_cl(x,gn); return torch.autograd.grad(y,(x,gn.weight,gn.bias),up)
    def aten():
        with torch.autocast("cuda",dtype=torch.bfloat16):
            y=gn(x)
        return torch.autograd.grad(y,(x,gn.weight,gn.bias),up.float())
    mb=x.numel()*2/1e6
    print(f"GroupNorm(32,256) [4,256,{H},{W}] bf16 ({mb:.1f} MB): own fwd+bwd {timeit(own):7.1f} us   ATen under autocast fwd+bwd {timeit(aten):7.1f} us")
t=torch.randn(4,256,100,167,device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
print("nchw_f32 level 0: own %.1f us   ATen contiguous().float() %.1f us" % (timeit(lambda: native.nchw_f32(t)), timeit(lambda: t.contiguous().float())))
