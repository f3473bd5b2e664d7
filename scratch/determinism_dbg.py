"""Is the student's eager forward bit-reproducible (same weights, same image)?  And the graphed-path eager call?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_model as T
dev = torch.device("cuda:0")
cfg, m = T._build(seed=13)
m.to(dev).train()
g = torch.Generator().manual_seed(31)
data, inj = T._batch(dev)
def run(img, graphed):
    m.bbox_head.graph_head = graphed
    for p in m.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats, outs, *_ = m.out_teacher(img, data["img_metas"])
        ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=inj["pred_keepid"], pred_logits=None,
                  pred_scores=None, pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
        out = m.train_step(dict(data, img=img, teacher_info=ti))
    out["loss"].backward()
    return out["log_vars"]
for step in range(4):
    img = torch.randn(2, 3, 192, 256, generator=g).to(dev)
    a = run(img, False); b = run(img, False); c = run(img, True)
    d1 = {k: abs(a[k] - b[k]) for k in a if a[k] != b[k]}
    d2 = {k: (a[k], c[k]) for k in a if abs(a[k] - c[k]) > 1e-6 * max(1, abs(a[k]))}
    print(step, "eager vs eager differing keys:", len(d1), dict(list(d1.items())[:4]))
    print(step, "eager vs graphed-path:", len(d2), dict(list(d2.items())[:6]))
