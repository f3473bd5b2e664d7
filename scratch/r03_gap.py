"""Cost of a dependent tiny kernel on one stream: eager launches vs one hipGraph of the same launches."""
import torch, time
x = torch.zeros(1024, device="cuda")
def run(n):
    for _ in range(n):
        x.add_(1.0)
for n in (2000,):
    run(200); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(n); e1.record(); torch.cuda.synchronize()
    print(f"eager: {e0.elapsed_time(e1) / n * 1e3:.2f} us per tiny kernel (GPU time between events)")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(10)
        with torch.cuda.graph(g, stream=s):
            run(n)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"graph: {e0.elapsed_time(e1) / n * 1e3:.2f} us per tiny kernel")
# a long kernel stream: does the gap hide behind execution?  8 MB adds (~10 us each)
y = torch.zeros(4 * 1024 * 1024, device="cuda")
def run2(n):
    for _ in range(n):
        y.add_(1.0)
run2(50); torch.cuda.synchronize()
e0.record(); run2(1000); e1.record(); torch.cuda.synchronize()
print(f"eager 16 MB add_: {e0.elapsed_time(e1) / 1000 * 1e3:.2f} us per kernel")
