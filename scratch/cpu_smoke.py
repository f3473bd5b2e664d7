import sys, time, copy; sys.path.insert(0,'.')
import torch
from dskd_amd.config import Config
from dskd_amd import builder, native
import dskd_amd.backbones, dskd_amd.necks, dskd_amd.transformer, dskd_amd.bbox, dskd_amd.losses
import dskd_amd.gfl_deformable_detr_head_il, dskd_amd.deformable_detr_il
from oracle.checker import OracleChecker
native.install_cpu_checker(OracleChecker())
cfg = Config.fromfile('/root/reference/configs/deformable_detr/chaosuan_gfl_deformable_detr_40_r50_8x4_1x_qoqo_il.py')
print(cfg.model.type, cfg.data.train.catsplit, type(cfg.optimizer), cfg.runner[0])
torch.manual_seed(0)
mcfg = cfg.model
mcfg.backbone.init_cfg = None
model = builder.build_detector(mcfg)
model.init_weights()
n = sum(p.numel() for p in model.parameters()); print("params", n/1e6)
teacher = copy.deepcopy(model)
with torch.no_grad():
    for p in teacher.parameters(): p.add_(torch.randn_like(p)*1e-3)
model.set_teacher(model=teacher)
model.LableInPCNTask = {'prev': list(range(40)), 'curr': list(range(40,80)), 'next': []}
model.train()
B,H,W=2,128,160
img=torch.randn(B,3,H,W)
metas=[dict(img_shape=(H,W,3), batch_input_shape=(H,W), scale_factor=1.0) for _ in range(B)]
gt_b=[torch.tensor([[10.,12.,60.,70.],[30.,20.,120.,100.]]), torch.tensor([[5.,5.,50.,40.]])]
gt_l=[torch.tensor([45,50]), torch.tensor([41])]
# inject teacher detections
N=B*300
g=torch.Generator().manual_seed(1)
with torch.no_grad():
    feats, outs, *_ = model.out_teacher(img, metas)
ti=dict(neck_feats=feats, head_outs=outs, pred_keepid=torch.tensor([3,17,300+5]), pred_logits=None, pred_scores=None,
        pred_labels=[torch.tensor([1,7]), torch.tensor([3])], pred_bboxes=[torch.tensor([[20.,20.,80.,90.],[0.,0.,30.,30.]]), torch.tensor([[40.,40.,100.,120.]])])
t=time.time()
out = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=ti))
print({k: round(v,5) for k,v in out['log_vars'].items()})
out['loss'].backward()
print("step time", time.time()-t)
gn = {n_: p.grad.norm().item() for n_,p in model.named_parameters() if p.grad is not None}
print(len(gn), "params with grad; none-grad:", [n_ for n_,p in model.named_parameters() if p.requires_grad and p.grad is None][:5])
