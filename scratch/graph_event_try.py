import torch
x=torch.randn(4096,4096,device='cuda')
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y=x@x
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
try:
    e0=torch.cuda.Event(enable_timing=True, external=True); e1=torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external flag", ex); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e0.record()
    y=x@x
    e1.record()
    z=y+1
for _ in range(3):
    g.replay(); torch.cuda.synchronize()
    try: print("elapsed ms", e0.elapsed_time(e1))
    except Exception as ex: print("elapsed failed:", repr(ex)[:200])
