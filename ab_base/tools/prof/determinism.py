"""Which module's output differs first between two EAGER forwards of the same image on the same weights?
(name the nondeterministic kernel behind the whole-step graph-vs-eager spread.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_model as T
dev = torch.device("cuda:0")
if os.environ.get("DET"):
    torch.backends.cudnn.deterministic = True
cfg, m = T._build(seed=13)
m.to(dev).train()
g = torch.Generator().manual_seed(31)
data, inj = T._batch(dev)
rec = []


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in tensors(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        ts = [t.detach().clone() for t in tensors(out) if t.is_floating_point()]
        if ts:
            rec.append((name, ts))
    return f


for tag, root in (("student", m), ("teacher", m.teacher_model)):
    for n, mod in root.named_modules():
        if n and not n.startswith("teacher_model"):
            mod.register_forward_hook(hook(f"{tag}.{n}"))


def run(img):
    rec.clear()
    m.bbox_head.graph_head = False
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats, outs, *_ = m.out_teacher(img, data["img_metas"])
        ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=inj["pred_keepid"], pred_logits=None,
                  pred_scores=None, pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
        out = m.train_step(dict(data, img=img, teacher_info=ti))
    torch.cuda.synchronize()
    return list(rec), out["log_vars"]


run(torch.randn(2, 3, 192, 256, generator=g).to(dev))
for step in range(5):
    img = torch.randn(2, 3, 192, 256, generator=g).to(dev)
    a, la = run(img)
    b, lb = run(img)
    if [n for n, _ in a] != [n for n, _ in b]:
        print("module call lists differ:", len(a), len(b))
    diff = []
    for (n, ta), (_, tb) in zip(a, b):
        for x, y in zip(ta, tb):
            if not torch.equal(x, y):
                diff.append((n, float((x.float() - y.float()).abs().max()), float(x.float().abs().max())))
                break
    keys = [k for k in la if la[k] != lb[k]]
    print(f"step {step}: {len(diff)} of {len(a)} module outputs differ; differing loss keys {len(keys)}")
    for d in diff[:6]:
        print("   first differing:", d)
