"""Do the dense detection losses (fwd + bwd) contain memset nodes / multi-block reductions?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg, model = bench.build_models(dev, 111, None)
head = model.bbox_head
nl, B, Q, C = 6, 4, 300, 80
g = torch.Generator().manual_seed(0)
cls = torch.randn(nl, B * Q, C, generator=g).to(dev).requires_grad_(True)
cxcywh = torch.rand(nl, B * Q, 4, generator=g).to(dev).requires_grad_(True)
lrtb = torch.rand(nl, B * Q, 4 * 17, generator=g).to(dev).requires_grad_(True)
labels = torch.randint(0, 81, (nl, B * Q), generator=g).to(dev)
tgt = torch.rand(nl, B * Q, 4, generator=g).to(dev)
pos = (labels < 80)
factors = torch.tensor([[1333., 800., 1333., 800.]]).repeat(B * Q, 1).to(dev)
def run():
    out = head.loss_layers_dense(cls, cxcywh, lrtb, labels, tgt, pos, factors, 17.0)
    tot = sum(o.sum() for o in out)
    grads = torch.autograd.grad(tot, (cls, cxcywh, lrtb))
    return out, grads
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
ev = prof.key_averages()
print("memset events:", [(e.key, e.count) for e in ev if "emset" in e.key])
nk = sum(e.count for e in ev if e.device_type == torch.autograd.DeviceType.CUDA)
print("device kernels:", nk, " device time: %.2f ms" % (sum(e.self_device_time_total for e in ev) / 1e3))
# graph replay check with changing inputs
gph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(gph, stream=s):
        gout, ggr = run()
torch.cuda.current_stream().wait_stream(s)
worst = 0.0
for rep in range(6):
    with torch.no_grad():
        cls.copy_(torch.randn(cls.shape, generator=g)); cxcywh.copy_(torch.rand(cxcywh.shape, generator=g))
        lrtb.copy_(torch.rand(lrtb.shape, generator=g)); labels.copy_(torch.randint(0, 81, labels.shape, generator=g))
        tgt.copy_(torch.rand(tgt.shape, generator=g)); pos.copy_(labels < 80)
    gph.replay(); torch.cuda.synchronize()
    eout, egr = run(); torch.cuda.synchronize()
    d = max(float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(list(gout) + list(ggr), list(eout) + list(egr)))
    worst = max(worst, d)
    print(f"replay {rep}: max rel diff graph vs eager {d:.2e}", flush=True)
print("WORST", worst)
