"""Is the frozen teacher forward safe to replay as a hipGraph on this runtime (memset-node bug)?
Replays with CHANGING inputs against eager results."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
static = data["img"].contiguous(memory_format=torch.channels_last)
metas = data["img_metas"]
t = model.teacher_model
def fwd(x):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        feats = t.extract_feat(x)
        outs = t.bbox_head.forward(feats, metas)
    return feats, outs
for _ in range(3): fwd(static)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fwd(static); torch.cuda.synchronize()
ms = [e for e in prof.key_averages() if "emset" in e.key]
print("memset events in one teacher forward:", [(e.key, e.count) for e in ms])
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        gfeats, gouts = fwd(static)
torch.cuda.current_stream().wait_stream(s)
gen = torch.Generator().manual_seed(5)
worst = 0.0
for rep in range(6):
    new = torch.randn(static.shape, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    static.copy_(new)
    g.replay(); torch.cuda.synchronize()
    ef, eo = fwd(static.clone(memory_format=torch.channels_last)); torch.cuda.synchronize()
    d = [float((a.float() - b.float()).abs().max()) for a, b in zip(gfeats, ef)]
    d += [float((a.float() - b.float()).abs().max()) for a, b in zip(gouts[:2], eo[:2])]
    d.append(float((gouts[3].float() - eo[3].float()).abs().max()))
    fin = all(bool(torch.isfinite(a.float()).all()) for a in list(gfeats) + [gouts[0], gouts[1], gouts[3]])
    worst = max(worst, max(d))
    print(f"replay {rep}: finite={fin} max|graph - eager| feats {max(d[:4]):.4f} cls {d[4]:.4f} box {d[5]:.5f} hs {d[6]:.4f}", flush=True)
print("WORST", worst)
