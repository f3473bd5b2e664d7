"""Does an MIOpen bf16 conv backward survive hipGraph replay?  (full-step graphs get NaN weight grads)"""
import sys, os, torch
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
torch.backends.cudnn.benchmark = mode in ("bench", "bench_det")
torch.backends.cudnn.deterministic = mode in ("det", "bench_det")
print("mode", mode, "benchmark", torch.backends.cudnn.benchmark, "deterministic", torch.backends.cudnn.deterministic)
B = int(os.environ.get("B", "1"))
torch.manual_seed(0)
specs = [  # (cin, cout, k, stride, H, W)
    (128, 128, 3, 1, 100, 167), (128, 512, 1, 1, 100, 167), (512, 128, 1, 1, 100, 167), (128, 128, 3, 1, 100, 167),
    (512, 1024, 1, 2, 100, 167), (256, 1024, 1, 1, 50, 84), (1024, 256, 1, 1, 50, 84), (256, 256, 3, 1, 50, 84),
]
convs = [torch.nn.Conv2d(ci, co, k, s, k // 2, bias=False).to(dev).to(memory_format=torch.channels_last) for ci, co, k, s, H, W in specs]
xs = [torch.randn(B, ci, H, W, device=dev).to(memory_format=torch.channels_last) for ci, co, k, s, H, W in specs]

def run():
    outs = []
    for c, x in zip(convs, xs):
        c.weight.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        tot = 0
        for c, x in zip(convs, xs):
            y = c(x)
            tot = tot + y.float().square().mean()
    tot.backward()
    return tot

for _ in range(3):
    run()
torch.cuda.synchronize()
ref = [c.weight.grad.clone() for c in convs]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t = run()
for rep in range(4):
    g.replay(); torch.cuda.synchronize()
    msg = []
    for i, c in enumerate(convs):
        gr = c.weight.grad
        bad = int((~torch.isfinite(gr)).sum())
        d = float((gr - ref[i]).abs().max()) if bad == 0 else float("nan")
        msg.append(f"{i}:{'BAD%d' % bad if bad else 'ok'}({d:.2g}/{float(ref[i].abs().max()):.2g})")
    print("replay", rep, float(t), " ".join(msg), flush=True)
