import sys, os, time; sys.path.insert(0,'.')
import torch, bench
from dskd_amd import native
from dskd_amd.runner import build_optimizer
dev=torch.device('cuda:0')
torch.backends.cudnn.benchmark=True
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log=True
ocfg=dict(cfg.optimizer[0]); ocfg['capturable']=True
opt = build_optimizer(model, ocfg)
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"]=data["img"].contiguous(memory_format=torch.channels_last)
amp=torch.bfloat16
def teacher_part():
    with torch.no_grad(), torch.autocast('cuda',dtype=amp):
        feats = model.teacher_model.extract_feat(data["img"])
        outs = model.teacher_model.bbox_head.forward(feats, data["img_metas"])
    return feats, outs
def decode(outs):
    with torch.no_grad():
        return model.teacher_model.bbox_head.get_bboxes(*outs, img_metas=data["img_metas"], rescale=False, cfg=model.teacher_test_cfg, need_logits=True)
def student_part(feats, outs):
    with torch.autocast('cuda',dtype=amp):
        ti={"neck_feats": feats, "head_outs": outs, "pred_keepid": synth["keep"], "pred_logits": None, "pred_scores": None, "pred_labels": synth["t_l"], "pred_bboxes": synth["t_b"]}
        losses = model(img=data["img"], img_metas=data["img_metas"], gt_bboxes=data["gt_bboxes"], gt_labels=data["gt_labels"], teacher_info=ti)
        loss, lv = model._parse_losses(losses)
    loss.backward()
    params=[p for g in opt.param_groups for p in g["params"] if p.grad is not None]
    torch.nn.utils.clip_grad_norm_(params, max_norm=0.1, norm_type=2, foreach=True)
    opt.step()
    return loss
# warmup on side stream
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        f,o=teacher_part(); decode(o); l=student_part(f,o)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print("eager loss", float(l))
g1=torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    feats, outs = teacher_part()
torch.cuda.synchronize(); print("teacher graph captured")
g2=torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g2):
    loss = student_part(feats, outs)
torch.cuda.synchronize(); print("student graph captured")
def step():
    g1.replay(); decode(outs); g2.replay()
for _ in range(3): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print(f"graph step {dt*1e3:.2f} ms -> {4/dt:.1f} img/s, loss {float(loss):.4f}")
