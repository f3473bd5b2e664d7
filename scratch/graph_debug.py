"""hipGraph step vs eager step at full scale: same init, same data, dropout 0 -> losses must track."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from dskd_amd import native
from dskd_amd.runner import build_optimizer
from dskd_amd.graph_step import GraphedDistillStep

B = int(os.environ.get("B", "4")); STEPS = int(os.environ.get("STEPS", "6"))
dev = torch.device("cuda:0")
native.load()
torch.backends.cudnn.benchmark = True

def make(use_graphs):
    cfg, model = bench.build_models(dev, 111, 0.0)
    model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
    model.lazy_log = True
    opt = build_optimizer(model, cfg.optimizer[0], capturable=True)
    return cfg, model, GraphedDistillStep(model, opt, amp_dtype=torch.bfloat16, max_norm=0.1, use_graphs=use_graphs, warmup=2)

cfg, mA, sA = make(False)
_, mB, sB = make(True)
data, synth = bench.make_batch(B, cfg.num_prev, 111, dev)
data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
inject = {"pred_bboxes": synth["t_b"], "pred_labels": synth["t_l"], "pred_keepid": synth["keep"]}
_orig_exchange = sB._exchange
def _checked_exchange(flat_logs):
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in mB.named_parameters()
           if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print("   [before U] non-finite grads:", len(bad), bad[:12], flush=True)
    return _orig_exchange(flat_logs)
sB._exchange = _checked_exchange
for i in range(STEPS):
    la = float(sA.step(data, inject)); lb = float(sB.step(data, inject))
    print(f"step {i}: eager {la:.5f}  graph {lb:.5f}  graphs captured {len(sB._graphs)}", flush=True)
    if sB._graphs:
        g = list(sB._graphs.values())[0]
        def nf(ts): return sum(0 if bool(torch.isfinite(t.float()).all()) else 1 for t in ts if torch.is_tensor(t))
        fp = {id(p) for p in sB._feat_params()}
        hg = [p.grad for p in mB.parameters() if p.grad is not None and id(p) not in fp]
        fg = [p.grad for p in mB.parameters() if p.grad is not None and id(p) in fp]
        print(f"   nonfinite: T.outs {nf(g['outs'])} T.feats {nf(g['feats'])} F.xs {nf(g['xs_raw'])} S.loss {nf([g['loss']])} "
              f"S.xgrad {nf([x.grad for x in g['xs']])} head grads {nf(hg)}/{len(hg)} feat grads {nf(fg)}/{len(fg)} "
              f"params {nf(list(mB.parameters()))}", flush=True)
        if nf(hg):
            names = [n for n, p in mB.named_parameters() if p.grad is not None and id(p) not in fp and not bool(torch.isfinite(p.grad).all())]
            print("   bad head grads:", names[:40])
worst = 0.0; bad = 0
for (n, pa), (_, pb) in zip(mA.named_parameters(), mB.named_parameters()):
    if not bool(torch.isfinite(pb).all()): bad += 1
    worst = max(worst, float((pa - pb).abs().max()))
print("non-finite params in graph model:", bad, " max |p_eager - p_graph| =", worst)
import time
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): sB.step(data, inject)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"graph step {dt*1e3:.2f} ms -> {B/dt:.1f} img/s")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): sA.step(data, inject)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"eager step {dt*1e3:.2f} ms -> {B/dt:.1f} img/s")
