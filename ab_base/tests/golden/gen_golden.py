"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own functions
(read-only checkout at /root/reference) in this container.  The fixtures (.npz: inputs and
expected outputs only) are committed; this script documents how they were made and is only
runnable where /root/reference exists (never on the GPU box).

The reference package cannot be imported as a whole (``import mmdet`` needs the
un-vendored ``mmcv``).  Its LEAF files for this path are loaded by
module path with a minimal stand-in for the few mmcv symbols they touch -- identity
decorators (``mmcv.jit``, ``force_fp32``, ``auto_fp16``) and a ``Registry`` -- none of which
takes part in the arithmetic.  What runs is the reference's code:
``GFLDeformableDETRHead_il.loss`` (Hungarian via the local scipy, QFL/DFL/L1/GIoU,
``loss_corr``, ``decode_v1``), ``GFLHungarianAssigner.assign``, ``bbox_overlaps``,
``Integral_average`` and the loss modules.

    python tests/golden/gen_golden.py               # rewrites tests/golden/*.npz
    python tests/golden/gen_golden.py --variants    # loss_variants_b2_l70.npz
    python tests/golden/gen_golden.py --datasplit   # data_split_cases.json
    python tests/golden/gen_golden.py --decode      # decode_cases.npz
    python tests/golden/gen_golden.py --head-forward  # head_forward_cases.npz
    python tests/golden/gen_golden.py --transformer-forward  # transformer_forward_cases.npz
    python tests/golden/gen_golden.py --decoder-loop  # decoder_loop_cases.npz
    python tests/golden/gen_golden.py --bbox2result   # bbox2result_cases.npz
    python tests/golden/gen_golden.py --ragged        # loss_ragged_*.npz
    python tests/golden/gen_golden.py --two-rank      # loss_two_rank_r{0,1}.npz
    python tests/golden/gen_golden.py --gfl           # gfl_cases.npz (stock GFL head: anchors, ATSS, targets, losses)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- loader
def _identity_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


class _Registry:
    def __init__(self, name, **kw):
        self.name, self.module_dict = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco if module is None else module

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)

    def get(self, k):
        return self.module_dict.get(k)


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return registry.get(args.pop("type"))(**args)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    models_reg = _Registry("models")
    mmcv = _mod("mmcv", jit=_identity_decorator)
    _mod("mmcv.utils", Registry=_Registry, build_from_cfg=build_from_cfg)
    mmcv.utils = sys.modules["mmcv.utils"]
    _mod("mmcv.cnn", MODELS=models_reg, Linear=nn.Linear, bias_init_with_prob=lambda p: float(-np.log((1 - p) / p)),
         constant_init=lambda m, v, bias=0: None)
    _mod("mmcv.runner", force_fp32=_identity_decorator, auto_fp16=_identity_decorator, BaseModule=_Dummy)
    _mod("mmcv.ops", batched_nms=None)

    for p in ("mmdet", "mmdet.core", "mmdet.core.bbox", "mmdet.core.bbox.assigners", "mmdet.core.bbox.samplers",
              "mmdet.core.bbox.match_costs", "mmdet.core.bbox.iou_calculators", "mmdet.core.utils", "mmdet.models",
              "mmdet.models.losses", "mmdet.models.dense_heads", "mmdet.models.utils", "mmdet.utils"):
        _pkg(p, os.path.join(REF, *p.split(".")))
    # registries (the reference's builder files only need mmcv's Registry)
    _load("mmdet.models.builder", "mmdet/models/builder.py")
    _load("mmdet.core.bbox.builder", "mmdet/core/bbox/builder.py")
    _load("mmdet.core.bbox.match_costs.builder", "mmdet/core/bbox/match_costs/builder.py")
    iou = _load("mmdet.core.bbox.iou_calculators.iou2d_calculator", "mmdet/core/bbox/iou_calculators/iou2d_calculator.py")
    sys.modules["mmdet.core.bbox.iou_calculators"].bbox_overlaps = iou.bbox_overlaps
    sys.modules["mmdet.core.bbox.iou_calculators"].BboxOverlaps2D = iou.BboxOverlaps2D
    tr = _load("mmdet.core.bbox.transforms", "mmdet/core/bbox/transforms.py")
    mc = _load("mmdet.core.bbox.match_costs.match_cost", "mmdet/core/bbox/match_costs/match_cost.py")
    sys.modules["mmdet.core.bbox.match_costs"].build_match_cost = sys.modules["mmdet.core.bbox.match_costs.builder"].build_match_cost
    _load("mmdet.utils.util_mixins", "mmdet/utils/util_mixins.py")
    ar = _load("mmdet.core.bbox.assigners.assign_result", "mmdet/core/bbox/assigners/assign_result.py")
    ba = _load("mmdet.core.bbox.assigners.base_assigner", "mmdet/core/bbox/assigners/base_assigner.py")
    ga = _load("mmdet.core.bbox.assigners.gfl_hungarian_assigner", "mmdet/core/bbox/assigners/gfl_hungarian_assigner.py")
    _load("mmdet.core.bbox.samplers.sampling_result", "mmdet/core/bbox/samplers/sampling_result.py")
    _load("mmdet.core.bbox.samplers.base_sampler", "mmdet/core/bbox/samplers/base_sampler.py")
    ps = _load("mmdet.core.bbox.samplers.pseudo_sampler", "mmdet/core/bbox/samplers/pseudo_sampler.py")
    _pkg("mmdet.core.mask")            # misc.py only imports two mask classes for isinstance checks
    _mod("mmdet.core.mask.structures", BitmapMasks=type("BitmapMasks", (), {}), PolygonMasks=type("PolygonMasks", (), {}))
    misc = _load("mmdet.core.utils.misc", "mmdet/core/utils/misc.py")
    du = types.ModuleType("mmdet.core.utils.dist_utils")
    du.reduce_mean = lambda t: t           # single process: dist_utils.py:68-74 returns the tensor
    sys.modules["mmdet.core.utils.dist_utils"] = du
    cu = sys.modules["mmdet.core.utils"]
    cu.filter_scores_and_topk, cu.multi_apply, cu.reduce_mean = misc.filter_scores_and_topk, misc.multi_apply, du.reduce_mean
    core = sys.modules["mmdet.core"]
    bb = sys.modules["mmdet.core.bbox.builder"]
    for k, v in dict(bbox_cxcywh_to_xyxy=tr.bbox_cxcywh_to_xyxy, bbox_xyxy_to_cxcywh=tr.bbox_xyxy_to_cxcywh,
                     build_assigner=bb.build_assigner, build_sampler=bb.build_sampler, multi_apply=misc.multi_apply,
                     reduce_mean=du.reduce_mean, anchor_inside_flags=None, bbox_overlaps=iou.bbox_overlaps,
                     images_to_levels=None, unmap=None, build_bbox_coder=None).items():
        setattr(core, k, v)
    sys.modules["mmdet.core.bbox"].bbox_overlaps = iou.bbox_overlaps
    _load("mmdet.models.losses.utils", "mmdet/models/losses/utils.py")
    losses = {n: _load(f"mmdet.models.losses.{n}", f"mmdet/models/losses/{n}.py")
              for n in ("kd_loss", "mse_loss", "gfocal_loss", "iou_loss", "smooth_l1_loss")}
    _mod("mmdet.models.utils.transformer",
         inverse_sigmoid=lambda x, eps=1e-5: torch.log(x.clamp(0, 1).clamp(min=eps) / (1 - x.clamp(0, 1)).clamp(min=eps)))
    _mod("mmdet.models.dense_heads.detr_head", DETRHead=_Dummy)
    head = _load("mmdet.models.dense_heads.gfl_deformable_detr_head_il",
                 "mmdet/models/dense_heads/gfl_deformable_detr_head_il.py")
    return dict(iou=iou, tr=tr, mc=mc, ga=ga, ps=ps, misc=misc, losses=losses, head=head)


# ----------------------------------------------------------------------------- synthetic inputs
def make_loss_inputs(B, L, seed, shapes, img_hw, n_t=4, n_gt=3, Q=300, C=80, D=256, nl=3):
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(nl, B, Q, C, generator=g) * 2 - 3
    box = torch.rand(nl, B, Q, 70, generator=g) * 0.9 + 0.05
    hs = torch.randn(1, B, Q, D, generator=g)          # loss() reads only hs[-1]
    hs_t = hs + 0.2 * torch.randn(1, B, Q, D, generator=g)
    feats_s = [torch.randn(B, D, h, w, generator=g) for h, w in shapes]
    feats_t = [f + 0.3 * torch.randn(f.shape, generator=g) for f in feats_s]
    gt_b, gt_l, t_b, t_l, keep = [], [], [], [], []
    for b in range(B):
        H, W = img_hw[b]

        def boxes(n):
            xy = torch.rand(n, 2, generator=g) * torch.tensor([0.6 * W, 0.6 * H])
            sz = torch.rand(n, 2, generator=g) * torch.tensor([0.35 * W, 0.35 * H]) + 8
            bx = torch.cat([xy, xy + sz], 1)
            bx[:, 0::2].clamp_(0, W)
            bx[:, 1::2].clamp_(0, H)
            return bx
        gt_b.append(boxes(n_gt))
        gt_l.append(torch.randint(L, C, (n_gt,), generator=g))
        t_b.append(boxes(n_t))
        t_l.append(torch.randint(0, L, (n_t,), generator=g))
        keep.append(b * Q + torch.randperm(Q, generator=g)[:n_t])
    return dict(cls=cls, box=box, hs=hs, hs_t=hs_t, feats_s=feats_s, feats_t=feats_t, gt_b=gt_b, gt_l=gt_l, t_b=t_b,
                t_l=t_l, keep=torch.cat(keep))


def run_reference_loss(ref, inp, L, img_hw, shapes):
    """Calls the reference's GFLDeformableDETRHead_il.loss on a namespace `self`."""
    H = ref["head"]
    L_ = ref["losses"]
    cls_ = H.GFLDeformableDETRHead_il
    self = types.SimpleNamespace()
    self.has_teacher = True
    self.cates_distill, self.feats_distill = "hard + teacher-first", "corr + fg_info + decode_v1"
    self.locat_distill, self.memory_distill = "", ""
    self.num_classes = self.cls_out_channels = 80
    self.bg_cls_weight, self.sync_cls_avg_factor, self.reg_max = 0, True, 16
    self.integral_average = H.Integral_average(16)
    self.assigner = ref["ga"].GFLHungarianAssigner(
        cls_cost=dict(type="QualityFocalLossCost", weight=2.0), reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
        iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))
    self.sampler = ref["ps"].PseudoSampler()
    self.loss_cls = L_["gfocal_loss"].QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=2.0)
    self.loss_dfl = L_["gfocal_loss"].DistributionFocalLoss(loss_weight=0.5)
    self.loss_bbox = L_["smooth_l1_loss"].L1Loss(loss_weight=5.0)
    self.loss_iou = L_["iou_loss"].GIoULoss(loss_weight=2.0)
    self.loss_fg_feature = L_["kd_loss"].KnowledgeDistillationKLDivLoss(loss_weight=1, T=2, reduction="sum")
    self.loss_corr = L_["mse_loss"].MSELoss(loss_weight=1, reduction="mean")
    for name in ("loss_single_split", "get_targets", "_get_target_single", "correlation_mat"):
        setattr(self, name, types.MethodType(getattr(cls_, name), self))
    B = inp["cls"].shape[1]
    metas = [dict(img_shape=(img_hw[b][0], img_hw[b][1], 3)) for b in range(B)]
    cls = inp["cls"].clone().requires_grad_(True)
    box = inp["box"].clone().requires_grad_(True)
    hs = inp["hs"].clone().requires_grad_(True)
    fs = [f.clone().requires_grad_(True) for f in inp["feats_s"]]
    teacher_info = dict(neck_feats=inp["feats_t"], head_outs=(None, None, None, inp["hs_t"]), pred_keepid=inp["keep"],
                        pred_labels=[t.clone() for t in inp["t_l"]], pred_bboxes=[t.clone() for t in inp["t_b"]])
    spatial = torch.tensor(shapes)
    losses = cls_.loss(self, cls, box, (None, spatial), hs, [b.clone() for b in inp["gt_b"]], [l.clone() for l in inp["gt_l"]],
                       metas, gt_bboxes_ignore=None, student_feat=fs, teacher_info=teacher_info,
                       task_labels={"prev": list(range(L)), "curr": list(range(L, 80)), "next": []})
    total = sum(v for k, v in losses.items() if "loss" in k)
    total.backward()
    out = {f"loss/{k}": v.detach().numpy() for k, v in losses.items()}
    out["grad/cls"] = cls.grad.numpy()
    out["grad/box"] = box.grad.numpy()
    out["grad/hs"] = hs.grad.numpy()
    out["grad/feats_s_absmax"] = np.array([0.0 if f.grad is None else float(f.grad.abs().max()) for f in fs])
    # per-term gradients w.r.t. hs for the two DSKD losses
    for key in ("loss_corr", "loss_fg_feature"):
        hs2 = inp["hs"].clone().requires_grad_(True)
        l2 = cls_.loss(self, inp["cls"].clone(), inp["box"].clone(), (None, spatial), hs2, [b.clone() for b in inp["gt_b"]],
                       [l.clone() for l in inp["gt_l"]], metas, gt_bboxes_ignore=None,
                       student_feat=[f.clone() for f in inp["feats_s"]], teacher_info=teacher_info,
                       task_labels={"prev": list(range(L)), "curr": list(range(L, 80)), "next": []})
        l2[key].backward()
        out[f"grad_hs/{key}"] = hs2.grad[-1].numpy()
    return out


def run_reference_variant(ref, inp, L, img_hw, shapes, feats_distill, memory_distill, mem_s, mem_t,
                          cates_distill="hard + teacher-first", locat_distill="", cls_t=None, box_t=None):
    """The other feature-distillation branches of the reference's loss():
    ``decode_v2`` (:721-772), ``kldv`` (:646-651), ``memory`` (:652-661).  Same harness as
    run_reference_loss; returns the branch's loss and the gradients it sends anywhere."""
    H, L_ = ref["head"], ref["losses"]
    cls_ = H.GFLDeformableDETRHead_il
    self = types.SimpleNamespace()
    self.has_teacher = True
    self.cates_distill, self.feats_distill = cates_distill, feats_distill
    self.locat_distill, self.memory_distill = locat_distill, memory_distill
    self.num_classes = self.cls_out_channels = 80
    self.bg_cls_weight, self.sync_cls_avg_factor, self.reg_max = 0, True, 16
    self.integral_average = H.Integral_average(16)
    self.assigner = ref["ga"].GFLHungarianAssigner(
        cls_cost=dict(type="QualityFocalLossCost", weight=2.0), reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
        iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))
    self.sampler = ref["ps"].PseudoSampler()
    self.loss_cls = L_["gfocal_loss"].QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=2.0)
    self.loss_dfl = L_["gfocal_loss"].DistributionFocalLoss(loss_weight=0.5)
    self.loss_bbox = L_["smooth_l1_loss"].L1Loss(loss_weight=5.0)
    self.loss_iou = L_["iou_loss"].GIoULoss(loss_weight=2.0)
    kd = L_["kd_loss"].KnowledgeDistillationKLDivLoss
    self.loss_fg_feature = kd(loss_weight=1, T=2, reduction="sum")
    self.loss_fd = kd(loss_weight=1, T=2)
    self.loss_memory = kd(loss_weight=1, T=2)
    self.loss_kd = kd(loss_weight=10, T=2)                                        # ctor defaults (:98-106)
    self.loss_ld_bbox = L_["smooth_l1_loss"].SmoothL1Loss(loss_weight=10, reduction="mean")
    self.loss_ld_logit = kd(loss_weight=0.25, T=10)
    self.loss_corr = L_["mse_loss"].MSELoss(loss_weight=1, reduction="mean")
    for name in ("loss_single_split", "get_targets", "_get_target_single", "correlation_mat"):
        setattr(self, name, types.MethodType(getattr(cls_, name), self))
    B = inp["cls"].shape[1]
    metas = [dict(img_shape=(img_hw[b][0], img_hw[b][1], 3)) for b in range(B)]
    hs = inp["hs"].clone().requires_grad_(True)
    fs = [f.clone().requires_grad_(True) for f in inp["feats_s"]]
    ms = mem_s.clone().requires_grad_(True)
    cls_in = inp["cls"].clone().requires_grad_(True)
    box_in = inp["box"].clone().requires_grad_(True)
    spatial = torch.tensor(shapes)
    teacher_info = dict(neck_feats=inp["feats_t"], head_outs=(cls_t, box_t, (mem_t, spatial), inp["hs_t"]),
                        pred_keepid=inp["keep"], pred_labels=[t.clone() for t in inp["t_l"]],
                        pred_bboxes=[t.clone() for t in inp["t_b"]])
    losses = cls_.loss(self, cls_in, box_in, (ms, spatial), hs, [b.clone() for b in inp["gt_b"]],
                       [l.clone() for l in inp["gt_l"]], metas, gt_bboxes_ignore=None, student_feat=fs,
                       teacher_info=teacher_info, task_labels={"prev": list(range(L)), "curr": list(range(L, 80)), "next": []})
    out = {}
    for key in ("loss_kd", "loss_ld_bbox", "loss_ld_logit"):
        if key in losses:
            out[f"loss/{key}"] = losses[key].detach().numpy()
            g = torch.autograd.grad(losses[key], [cls_in, box_in], allow_unused=True, retain_graph=True)
            if g[0] is not None:
                out[f"grad_cls_last/{key}"] = g[0][-1].numpy()
            if g[1] is not None:
                out[f"grad_box_last/{key}"] = g[1][-1].numpy()
    for key in ("loss_fg_feature", "loss_fd", "loss_memory"):
        if key in losses:
            out[f"loss/{key}"] = losses[key].detach().numpy()
            if losses[key].requires_grad:
                g = torch.autograd.grad(losses[key], [hs, ms] + fs, allow_unused=True, retain_graph=True)
                out[f"grad_hs_absmax/{key}"] = np.array(0.0 if g[0] is None else float(g[0].abs().max()))
                if g[1] is not None:
                    out[f"grad_mem/{key}"] = g[1].numpy()
                for i, gi in enumerate(g[2:]):
                    if gi is not None:
                        out[f"grad_feat{i}/{key}"] = gi.numpy()
            else:
                out[f"nograd/{key}"] = np.array(1)
    out["keys"] = np.array(sorted(losses.keys()))
    return out


def main_variants():
    """tests/golden/loss_variants_b2_l70.npz: inputs of loss_b2_l70 + student / teacher memories."""
    ref = load_reference()
    B, L, shapes, img_hw = 2, 70, [(9, 14), (5, 7)], [(72, 112), (70, 100)]
    inp = make_loss_inputs(B, L, 11 + B, shapes, img_hw)
    g = torch.Generator().manual_seed(77)
    n = sum(h * w for h, w in shapes)
    mem_s = torch.randn(n, B, 256, generator=g)
    mem_t = mem_s + 0.3 * torch.randn(n, B, 256, generator=g)
    cls_t = inp["cls"][-1:] + 0.5 * torch.randn(inp["cls"][-1:].shape, generator=g)      # teacher head outputs (last layer)
    box_t = (inp["box"][-1:] + 0.05 * torch.randn(inp["box"][-1:].shape, generator=g)).clamp(0.01, 0.99)
    flat = {"mem_s": mem_s.numpy(), "mem_t": mem_t.numpy(), "cls_t": cls_t.numpy(), "box_t": box_t.numpy()}
    for tag, fd, md in (("decode_v2", "corr + fg_info + decode_v2", ""), ("kldv", "corr + kldv", ""),
                        ("memory", "corr", "memory"), ("sg_out", "corr + fg_info + sg_out", ""),
                        ("fg_only", "corr + fg_info + fg_only", "")):
        out = run_reference_variant(ref, inp, L, img_hw, shapes, fd, md, mem_s, mem_t)
        for k, v in out.items():
            flat[f"{tag}/{k}"] = v
        print(tag, {k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items() if k != "keys"})
    for tag, cd, ld in (("soft", "hard + soft + teacher-first", ""), ("ld", "hard + teacher-first", "bbox + logit")):
        out = run_reference_variant(ref, inp, L, img_hw, shapes, "corr", "", mem_s, mem_t, cates_distill=cd,
                                    locat_distill=ld, cls_t=cls_t, box_t=box_t)
        for k, v in out.items():
            flat[f"{tag}/{k}"] = v
        print(tag, {k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items() if k != "keys"})
    np.savez_compressed(os.path.join(OUT, "loss_variants_b2_l70.npz"), **flat)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref = load_reference()

    # 1. known answers of the reference's own tests
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    giou = ref["iou"].bbox_overlaps(b1, b2, "giou", is_aligned=True, eps=1e-7)
    kd = ref["losses"]["kd_loss"].KnowledgeDistillationKLDivLoss(loss_weight=1.0, T=1)
    kd_eq = kd(torch.Tensor([[100.0, 100.0]]), torch.Tensor([[1.0, 1.0]]))            # test_losses.py:98-102
    kd_w = kd(torch.Tensor([[100.0, -100.0], [100.0, 100.0]]), torch.Tensor([[1.0, 0.0], [1.0, 1.0]]),
              torch.Tensor([0.0, 1.0]))                                                # test_losses.py:104-110
    np.savez(os.path.join(OUT, "known_answers.npz"), b1=b1.numpy(), b2=b2.numpy(), giou=giou.numpy(),
             kd_equal=kd_eq.numpy(), kd_weighted=kd_w.numpy())

    # 2. cost matrices + assignments from the reference assigner (local scipy inside)
    g = torch.Generator().manual_seed(5)
    assigner = ref["ga"].GFLHungarianAssigner(
        cls_cost=dict(type="QualityFocalLossCost", weight=2.0), reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
        iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))
    cases = {}
    for k, (Q, G, w, h) in enumerate([(300, 17, 1333, 800), (300, 1, 640, 480), (300, 60, 1333, 800), (20, 25, 200, 100)]):
        bbox = torch.rand(Q, 4, generator=g) * torch.tensor([1, 1, 0.5, 0.5])
        cls = torch.randn(Q, 80, generator=g) * 3
        xy = torch.rand(G, 2, generator=g) * torch.tensor([0.6 * w, 0.6 * h])
        sz = torch.rand(G, 2, generator=g) * torch.tensor([0.35 * w, 0.35 * h]) + 8
        gt = torch.cat([xy, xy + sz], 1)
        lab = torch.randint(0, 80, (G,), generator=g)
        meta = dict(img_shape=(h, w, 3))
        # the cost as the reference builds it (lines 120-140 of its assign())
        factor = gt.new_tensor([w, h, w, h]).unsqueeze(0)
        ngt = gt / factor
        cost = assigner.cls_cost(cls, lab, ref["tr"].bbox_cxcywh_to_xyxy(bbox), ngt) + assigner.reg_cost(bbox, ngt) + \
            assigner.iou_cost(ref["tr"].bbox_cxcywh_to_xyxy(bbox) * factor, gt)
        res = assigner.assign(bbox, cls, gt, lab, torch.zeros(Q, 68), meta)
        cases.update({f"c{k}/bbox": bbox.numpy(), f"c{k}/cls": cls.numpy(), f"c{k}/gt": gt.numpy(), f"c{k}/lab": lab.numpy(),
                      f"c{k}/wh": np.array([w, h], dtype=np.float32), f"c{k}/cost": cost.numpy(),
                      f"c{k}/gt_inds": res.gt_inds.numpy(), f"c{k}/labels": res.labels.numpy()})
    np.savez_compressed(os.path.join(OUT, "assign_cases.npz"), **cases)

    # 3. Integral_average + elementwise loss modules
    ia = ref["head"].Integral_average(16)
    x = torch.rand(12, 68, generator=g)
    pred = torch.randn(16, 80, generator=g)
    lab = torch.randint(0, 81, (16,), generator=g)
    sc = torch.rand(16, generator=g)
    qfl = ref["losses"]["gfocal_loss"].QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=2.0)
    dfl = ref["losses"]["gfocal_loss"].DistributionFocalLoss(loss_weight=0.5)
    dp = torch.rand(24, 17, generator=g)
    dl = torch.rand(24, generator=g) * 0.4
    dw = (torch.rand(24, generator=g) > 0.5).float()
    np.savez(os.path.join(OUT, "elementwise.npz"), ia_in=x.numpy(), ia_out=ia(x).numpy(), qfl_pred=pred.numpy(),
             qfl_label=lab.numpy(), qfl_score=sc.numpy(), qfl_out=qfl(pred, (lab, sc), None, avg_factor=3.0).numpy(),
             dfl_pred=dp.numpy(), dfl_label=dl.numpy(), dfl_w=dw.numpy(), dfl_out=dfl(dp, dl, weight=dw, avg_factor=12.0).numpy())

    # 4. the reference loss() end to end (two cases)
    for name, B, L, shapes, img_hw in [("loss_b1_l40", 1, 40, [(13, 21), (7, 11), (4, 6), (2, 3)], [(100, 167)]),
                                       ("loss_b2_l70", 2, 70, [(9, 14), (5, 7)], [(72, 112), (70, 100)])]:
        inp = make_loss_inputs(B, L, 11 + B, shapes, img_hw)
        out = run_reference_loss(ref, inp, L, img_hw, shapes)
        flat = {"B": np.array(B), "L": np.array(L), "shapes": np.array(shapes), "img_hw": np.array(img_hw),
                "cls": inp["cls"].numpy(), "box": inp["box"].numpy(), "hs": inp["hs"].numpy(),
                "hs_t_last": inp["hs_t"][-1].numpy(), "keep": inp["keep"].numpy()}
        for i in range(len(shapes)):
            flat[f"feat_s{i}"] = inp["feats_s"][i].numpy()
            flat[f"feat_t{i}"] = inp["feats_t"][i].numpy()
        for b in range(B):
            flat[f"gt_b{b}"], flat[f"gt_l{b}"] = inp["gt_b"][b].numpy(), inp["gt_l"][b].numpy()
            flat[f"t_b{b}"], flat[f"t_l{b}"] = inp["t_b"][b].numpy(), inp["t_l"][b].numpy()
        flat.update(out)
        # keep fixtures small: float16 would change inputs, so only drop what is re-derivable
        flat.pop("grad/cls")
        flat["grad/cls_sum_abs"] = np.abs(out["grad/cls"]).sum(axis=-1)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **flat)
        print(name, {k: float(v) for k, v in out.items() if k.startswith("loss/")})


def main_ragged():
    """tests/golden/loss_ragged_*.npz: the reference's loss() on RAGGED batches -- the second image carries no
    teacher detection, no ground truth, or neither (then its matching problems are empty) -- same format
    as loss_b2_l70.npz, 60 queries to keep the fixtures small."""
    ref = load_reference()
    B, L, shapes, img_hw, n_t, Q = 2, 70, [(9, 14), (5, 7)], [(72, 112), (70, 100)], 4, 60
    for case in ("no_teacher_boxes", "no_gt", "empty"):
        inp = make_loss_inputs(B, L, 21, shapes, img_hw, n_t=n_t, Q=Q)
        if case in ("no_teacher_boxes", "empty"):
            inp["t_b"][1], inp["t_l"][1] = torch.zeros(0, 4), torch.zeros(0, dtype=torch.long)
            inp["keep"] = inp["keep"][:n_t]
        if case in ("no_gt", "empty"):
            inp["gt_b"][1], inp["gt_l"][1] = torch.zeros(0, 4), torch.zeros(0, dtype=torch.long)
        out = run_reference_loss(ref, inp, L, img_hw, shapes)
        flat = {"B": np.array(B), "L": np.array(L), "shapes": np.array(shapes), "img_hw": np.array(img_hw),
                "cls": inp["cls"].numpy(), "box": inp["box"].numpy(), "hs": inp["hs"].numpy(),
                "hs_t_last": inp["hs_t"][-1].numpy(), "keep": inp["keep"].numpy()}
        for i in range(len(shapes)):
            flat[f"feat_s{i}"], flat[f"feat_t{i}"] = inp["feats_s"][i].numpy(), inp["feats_t"][i].numpy()
        for b in range(B):
            flat[f"gt_b{b}"], flat[f"gt_l{b}"] = inp["gt_b"][b].numpy(), inp["gt_l"][b].numpy()
            flat[f"t_b{b}"], flat[f"t_l{b}"] = inp["t_b"][b].numpy(), inp["t_l"][b].numpy()
        flat.update(out)
        flat.pop("grad/cls")
        flat["grad/cls_sum_abs"] = np.abs(out["grad/cls"]).sum(axis=-1)
        np.savez_compressed(os.path.join(OUT, f"loss_ragged_{case}.npz"), **flat)
        print(case, {k: float(v) for k, v in out.items() if k in ("loss/loss_corr", "loss/loss_fg_feature", "loss/loss_cls")})


def main_two_rank():
    """tests/golden/loss_two_rank_r{0,1}.npz: the reference's loss() as TWO data-parallel ranks would evaluate it.
    The only cross-rank coupling inside loss() is ``reduce_mean`` (core/utils/dist_utils.py:68-74) of the
    per-layer normalisers (``num_total_pos``, ``cls_avg_factor``; gfl_deformable_detr_head_il.py
    ``loss_single_split``).  Pass 1 runs each rank's batch with a recording ``reduce_mean``; pass 2 replays
    both with ``reduce_mean`` returning the mean over the two ranks at the same call position -- exactly what
    the all-reduce delivers -- and stores each rank's losses and gradients."""
    ref = load_reference()
    H = ref["head"]
    B, L, shapes, img_hw, Q = 2, 70, [(9, 14), (5, 7)], [(72, 112), (70, 100)], 60
    inputs = [make_loss_inputs(B, L, 31, shapes, img_hw, n_t=4, n_gt=3, Q=Q),
              make_loss_inputs(B, L, 32, shapes, img_hw, n_t=2, n_gt=6, Q=Q)]
    recorded = []
    for inp in inputs:
        calls = []

        def rec(tns, calls=calls):
            calls.append(tns.clone())
            return tns
        H.reduce_mean = rec
        run_reference_loss(ref, inp, L, img_hw, shapes)
        recorded.append(calls)
    n = len(recorded[0]) // 3                     # run_reference_loss evaluates loss() three times
    assert len(recorded[0]) == len(recorded[1]) == 3 * n and n > 0
    means = [(a + b) / 2 for a, b in zip(recorded[0][:n], recorded[1][:n])]
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(recorded[0][:n], recorded[1][:n]))   # the ranks do differ
    for r, inp in enumerate(inputs):
        pos = [0]

        def feed(tns, pos=pos):
            m = means[pos[0] % n]
            pos[0] += 1
            assert m.shape == tns.shape
            return m.to(tns.dtype)
        H.reduce_mean = feed
        out = run_reference_loss(ref, inp, L, img_hw, shapes)
        flat = {"B": np.array(B), "L": np.array(L), "shapes": np.array(shapes), "img_hw": np.array(img_hw),
                "cls": inp["cls"].numpy(), "box": inp["box"].numpy(), "hs": inp["hs"].numpy(),
                "hs_t_last": inp["hs_t"][-1].numpy(), "keep": inp["keep"].numpy(),
                "reduce_mean_local": np.array([float(x.reshape(-1)[0]) for x in recorded[r][:n]]),
                "reduce_mean_global": np.array([float(x.reshape(-1)[0]) for x in means])}
        for i in range(len(shapes)):
            flat[f"feat_s{i}"], flat[f"feat_t{i}"] = inp["feats_s"][i].numpy(), inp["feats_t"][i].numpy()
        for b in range(B):
            flat[f"gt_b{b}"], flat[f"gt_l{b}"] = inp["gt_b"][b].numpy(), inp["gt_l"][b].numpy()
            flat[f"t_b{b}"], flat[f"t_l{b}"] = inp["t_b"][b].numpy(), inp["t_l"][b].numpy()
        flat.update(out)
        flat.pop("grad/cls")
        flat["grad/cls_sum_abs"] = np.abs(out["grad/cls"]).sum(axis=-1)
        np.savez_compressed(os.path.join(OUT, f"loss_two_rank_r{r}.npz"), **flat)
        print("rank", r, "calls per loss()", n, "local", flat["reduce_mean_local"][:4], "global", flat["reduce_mean_global"][:4],
              {k: float(v) for k, v in out.items() if k in ("loss/loss_cls", "loss/loss_corr")})
    H.reduce_mean = lambda t: t


def main_decode():
    """tests/golden/decode_cases.npz: the reference's teacher decode ``get_bboxes`` -> ``_get_bboxes_single``
    (gfl_deformable_detr_head_il.py:1535-1668) -> ``filter_scores_and_topk`` (core/utils/misc.py:119-165) on
    seeded head outputs: many candidates (top-100 cut, queries kept twice), a few, none, and rescale."""
    ref = load_reference()
    H = ref["head"]
    cls_ = H.GFLDeformableDETRHead_il
    self = types.SimpleNamespace()
    self.num_query, self.num_classes = 300, 80
    self.test_cfg = dict(max_per_img=100, score_thr=0.3)
    self.loss_cls = ref["losses"]["gfocal_loss"].QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=2.0)
    self.integral_average = H.Integral_average(16)
    for name in ("get_bboxes", "_get_bboxes_single"):
        setattr(self, name, types.MethodType(getattr(cls_, name), self))
    flat = {}
    g = torch.Generator().manual_seed(404)
    for tag, shift, rescale, cfg in (("many", -3.0, False, None), ("few", -6.5, False, None), ("none", -12.0, False, None),
                                     ("rescale", -4.0, True, None), ("cfg", -3.0, False, dict(max_per_img=17, score_thr=0.45))):
        B = 2
        cls = torch.randn(1, B, 300, 80, generator=g) * 2 + shift       # get_bboxes reads the last layer only
        box = torch.rand(1, B, 300, 70, generator=g) * 0.9 + 0.05
        metas = [dict(img_shape=(72, 112, 3), scale_factor=np.array([1.25, 1.5, 1.25, 1.5], dtype=np.float32)),
                 dict(img_shape=(640, 427, 3), scale_factor=np.array([0.8, 0.8, 0.8, 0.8], dtype=np.float32))]
        with torch.no_grad():
            out = self.get_bboxes(cls, box, None, None, img_metas=metas, rescale=rescale, cfg=cfg, need_logits=True)
        flat[f"{tag}/cls"], flat[f"{tag}/box"] = cls.numpy(), box.numpy()
        flat[f"{tag}/rescale"] = np.array(int(rescale))
        flat[f"{tag}/cfg"] = np.array([cfg["max_per_img"], cfg["score_thr"]] if cfg else [100, 0.3], dtype=np.float64)
        for i, (db, dl, dlog, keep) in enumerate(out):
            flat[f"{tag}/{i}/bboxes"], flat[f"{tag}/{i}/labels"] = db.numpy(), dl.numpy()
            flat[f"{tag}/{i}/logits"], flat[f"{tag}/{i}/keepid"] = dlog.numpy(), keep.numpy()
            flat[f"{tag}/{i}/img_shape"] = np.array(metas[i]["img_shape"])
            flat[f"{tag}/{i}/scale_factor"] = metas[i]["scale_factor"]
        print(tag, [tuple(o[0].shape) for o in out], "queries kept twice:",
              [int(len(o[3]) - len(torch.unique(o[3]))) for o in out])
    np.savez_compressed(os.path.join(OUT, "decode_cases.npz"), **flat)


def main_head_forward():
    """tests/golden/head_forward_cases.npz: the reference's ``GFLDeformableDETRHead_il.forward``
    (gfl_deformable_detr_head_il.py:196-281) with its own ``SinePositionalEncoding``
    (models/utils/positional_encoding.py:11-100) around a STUB transformer that returns seeded tensors
    and records what it was handed: padding masks (nearest interpolation of the image mask), sine
    encodings, query embedding; and the per-layer class / box branches with the reference-point
    shift on the first two box channels and the sigmoid on all of them."""
    ref = load_reference()
    _mod("mmcv.cnn.bricks", )
    _mod("mmcv.cnn.bricks.transformer", POSITIONAL_ENCODING=_Registry("pe"))
    pe_mod = _load("mmdet.models.utils.positional_encoding", "mmdet/models/utils/positional_encoding.py")
    H = ref["head"]
    D, Q, nl, C, RC = 32, 40, 6, 80, 70          # the arithmetic does not depend on 300 queries; keeps the fixture small
    flat = {}
    for tag, B, bis, shapes_img, feat_hw in (("full", 2, (64, 96), [(64, 96), (64, 96)], [(8, 12), (4, 6), (2, 3)]),
                                             ("padded", 3, (72, 112), [(72, 112), (50, 112), (72, 61)], [(9, 14), (5, 7), (3, 4)])):
        g = torch.Generator().manual_seed({"full": 31, "padded": 32}[tag])
        cls_b = nn.Linear(D, C)
        reg_b = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, D), nn.ReLU(), nn.Linear(D, RC))
        with torch.no_grad():
            for prm in list(cls_b.parameters()) + list(reg_b.parameters()):
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)
        emb = nn.Embedding(Q, 2 * D)
        with torch.no_grad():
            emb.weight.copy_(torch.randn(Q, 2 * D, generator=g))
        N = sum(h * w for h, w in feat_hw)
        ret = dict(hs=torch.randn(nl, Q, B, D, generator=g), init=torch.rand(B, Q, 2, generator=g) * 0.98 + 0.01,
                   memory=torch.randn(N, B, D, generator=g))
        # without box refinement the decoder hands back the SAME reference points for every layer
        # (models/utils/transformer.py:686-703: they only move when reg_branches is given)
        ret["inter"] = ret["init"].unsqueeze(0).expand(nl, -1, -1, -1).clone()
        seen = {}

        def transformer(mlvl_feats, mlvl_masks, query_embeds, mlvl_pos, reg_branches=None, cls_branches=None, **kw):
            seen.update(masks=mlvl_masks, pos=mlvl_pos, query=query_embeds, reg=reg_branches, cls=cls_branches)
            return ret["hs"], ret["init"], ret["inter"], ret["memory"], None, None

        self = types.SimpleNamespace(as_two_stage=False, with_box_refine=False, transformer=transformer,
                                     positional_encoding=pe_mod.SinePositionalEncoding(num_feats=D // 2, normalize=True, offset=-0.5),
                                     query_embedding=emb, cls_branches=nn.ModuleList([cls_b] * nl),
                                     reg_branches=nn.ModuleList([reg_b] * nl))
        feats = [torch.randn(B, D, h, w, generator=g) for h, w in feat_hw]
        metas = [dict(img_shape=(h, w, 3), batch_input_shape=bis) for h, w in shapes_img]
        with torch.no_grad():
            out = H.GFLDeformableDETRHead_il.forward(self, feats, metas)
        assert seen["reg"] is None and seen["cls"] is None and seen["query"] is emb.weight
        flat[f"{tag}/batch_input_shape"] = np.array(bis)
        flat[f"{tag}/img_shapes"] = np.array(shapes_img)
        flat[f"{tag}/feat_hw"] = np.array(feat_hw)
        for k, v in ret.items():
            flat[f"{tag}/ret/{k}"] = v.numpy()
        for k, v in list(cls_b.state_dict().items()):
            flat[f"{tag}/cls_branch/{k}"] = v.numpy()
        for k, v in list(reg_b.state_dict().items()):
            flat[f"{tag}/reg_branch/{k}"] = v.numpy()
        flat[f"{tag}/query_embedding"] = emb.weight.detach().numpy()
        for i, (m, pe) in enumerate(zip(seen["masks"], seen["pos"])):
            flat[f"{tag}/mask{i}"], flat[f"{tag}/pos{i}"] = m.numpy(), pe.numpy()
        flat[f"{tag}/out/cls"], flat[f"{tag}/out/box"] = out[0].numpy(), out[1].numpy()
        assert torch.equal(out[3], ret["hs"].permute(0, 2, 1, 3)) and out[2] is ret["memory"]   # 4th output: hs, batch-first
        print(tag, tuple(out[0].shape), tuple(out[1].shape), [int(m.sum()) for m in seen["masks"]])
    np.savez_compressed(os.path.join(OUT, "head_forward_cases.npz"), **flat)


def load_reference_transformer():
    """The reference's mmdet/models/utils/transformer.py, loaded with dummy classes for the ext-mmcv names
    it imports (layers, attention op: none of them is run here -- encoder / decoder are stubbed)."""
    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    if "mmcv" not in sys.modules:
        load_reference()
    cnn = sys.modules["mmcv.cnn"]
    for k in ("build_activation_layer", "build_conv_layer", "build_norm_layer", "xavier_init"):
        setattr(cnn, k, None)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", TRANSFORMER_LAYER=_Registry("tl"), TRANSFORMER_LAYER_SEQUENCE=_Registry("tls"))
    _mod("mmcv.cnn.bricks.transformer", BaseTransformerLayer=_Dummy, TransformerLayerSequence=_Dummy,
         build_transformer_layer_sequence=None, POSITIONAL_ENCODING=_Registry("pe"))
    _mod("mmcv.runner.base_module", BaseModule=_Dummy)
    sys.modules["mmcv.utils"].to_2tuple = lambda x: (x, x)
    _mod("mmcv.ops.multi_scale_deform_attn", MultiScaleDeformableAttention=_Dummy)
    _load("mmdet.models.utils.builder", "mmdet/models/utils/builder.py")
    return _load("mmdet.models.utils.ref_transformer", "mmdet/models/utils/transformer.py")


def main_transformer_forward():
    """tests/golden/transformer_forward_cases.npz: the reference's ``DeformableDetrTransformer.forward``
    (models/utils/transformer.py:875-1055, with its ``get_valid_ratio`` :865-873 and
    ``get_reference_points`` :830-863) around STUB encoder / decoder that return seeded tensors and record
    what they are handed: flattened features, level-embedded positional encodings, padding mask, valid
    ratios, encoder reference points, level start indices; query / query_pos split, decoder reference
    points ``sigmoid(Linear(query_pos))``."""
    tr = load_reference_transformer()
    cls_ = tr.DeformableDetrTransformer
    D, Q, nl = 32, 20, 6
    flat = {}
    for tag, B, canvas, img_hw, feat_hw in (("full", 2, (64, 96), [(64, 96), (64, 96)], [(8, 12), (4, 6), (2, 3), (1, 2)]),
                                            ("padded", 3, (72, 112), [(72, 112), (50, 112), (72, 61)],
                                             [(9, 14), (5, 7), (3, 4), (2, 2)])):
        g = torch.Generator().manual_seed({"full": 41, "padded": 42}[tag])
        N = sum(h * w for h, w in feat_hw)
        feats = [torch.randn(B, D, h, w, generator=g) for h, w in feat_hw]
        img_mask = torch.ones(B, *canvas)
        for i, (h, w) in enumerate(img_hw):
            img_mask[i, :h, :w] = 0
        masks = [torch.nn.functional.interpolate(img_mask[None], size=hw).to(torch.bool).squeeze(0) for hw in feat_hw]
        pos = [torch.randn(B, D, h, w, generator=g) for h, w in feat_hw]
        query_embed = torch.randn(Q, 2 * D, generator=g)
        ref_lin = nn.Linear(D, 2)
        with torch.no_grad():
            ref_lin.weight.copy_(torch.randn(2, D, generator=g) * 0.3)
            ref_lin.bias.copy_(torch.randn(2, generator=g) * 0.3)
        level_embeds = torch.randn(len(feat_hw), D, generator=g)
        enc_ret = torch.randn(N, B, D, generator=g)
        dec_ret = (torch.randn(nl, Q, B, D, generator=g), torch.rand(nl, B, Q, 2, generator=g))
        seen = {}

        def encoder(**kw):
            seen["enc"] = kw
            return enc_ret

        def decoder(**kw):
            seen["dec"] = kw
            return dec_ret

        self = types.SimpleNamespace(as_two_stage=False, encoder=encoder, decoder=decoder, level_embeds=level_embeds,
                                     reference_points=ref_lin, get_reference_points=cls_.get_reference_points)
        self.get_valid_ratio = types.MethodType(cls_.get_valid_ratio, self)
        with torch.no_grad():
            out = cls_.forward(self, feats, masks, query_embed, pos, reg_branches=None, cls_branches=None)
        flat[f"{tag}/canvas"], flat[f"{tag}/img_hw"], flat[f"{tag}/feat_hw"] = np.array(canvas), np.array(img_hw), np.array(feat_hw)
        for i in range(len(feat_hw)):
            flat[f"{tag}/feat{i}"], flat[f"{tag}/mask{i}"], flat[f"{tag}/pos{i}"] = feats[i].numpy(), masks[i].numpy(), pos[i].numpy()
        flat[f"{tag}/query_embed"], flat[f"{tag}/level_embeds"] = query_embed.numpy(), level_embeds.numpy()
        flat[f"{tag}/ref_w"], flat[f"{tag}/ref_b"] = ref_lin.weight.detach().numpy(), ref_lin.bias.detach().numpy()
        flat[f"{tag}/enc_ret"], flat[f"{tag}/dec_ret0"], flat[f"{tag}/dec_ret1"] = enc_ret.numpy(), dec_ret[0].numpy(), dec_ret[1].numpy()
        e, d = seen["enc"], seen["dec"]
        assert e["key"] is None and e["value"] is None and d["key"] is None and d["reg_branches"] is None
        for k in ("query", "query_pos", "query_key_padding_mask", "spatial_shapes", "reference_points", "level_start_index",
                  "valid_ratios"):
            flat[f"{tag}/enc/{k}"] = e[k].numpy()
        for k in ("query", "value", "query_pos", "key_padding_mask", "reference_points", "spatial_shapes", "level_start_index",
                  "valid_ratios"):
            flat[f"{tag}/dec/{k}"] = d[k].numpy()
        inter_states, init_ref, inter_refs, info_all, a, b = out
        assert a is None and b is None and inter_states is dec_ret[0] and inter_refs is dec_ret[1]
        flat[f"{tag}/out/init_reference"] = init_ref.numpy()
        flat[f"{tag}/out/memory"], flat[f"{tag}/out/spatial_shapes"] = info_all[0].numpy(), info_all[1].numpy()
        print(tag, {k: tuple(v.shape) for k, v in e.items() if torch.is_tensor(v)}, float(e["valid_ratios"].min()))
    np.savez_compressed(os.path.join(OUT, "transformer_forward_cases.npz"), **flat)


def main_decoder_loop():
    """tests/golden/decoder_loop_cases.npz: the reference's ``DeformableDetrTransformerDecoder.forward``
    (models/utils/transformer.py:639-709) over STUB layers that return seeded tensors and record the
    reference points they are handed (``reference_points[:, :, None] * valid_ratios[:, None]``), without
    and with ``reg_branches`` (iterative refinement of the reference points, detached)."""
    tr = load_reference_transformer()
    cls_ = tr.DeformableDetrTransformerDecoder
    D, Q, B, nl, L = 16, 12, 2, 3, 4
    flat = {}
    for tag, refine in (("plain", False), ("refine", True)):
        g = torch.Generator().manual_seed({"plain": 51, "refine": 52}[tag])
        query = torch.randn(Q, B, D, generator=g)
        ref = torch.rand(B, Q, 2, generator=g) * 0.9 + 0.05
        vr = torch.rand(B, L, 2, generator=g) * 0.5 + 0.5
        outs = [torch.randn(Q, B, D, generator=g) for _ in range(nl)]
        regs = None
        if refine:
            regs = nn.ModuleList([nn.Linear(D, 2) for _ in range(nl)])
            with torch.no_grad():
                for m in regs:
                    m.weight.copy_(torch.randn(2, D, generator=g) * 0.3)
                    m.bias.copy_(torch.randn(2, generator=g) * 0.3)
        seen = []

        def make_layer(i):
            def layer(output, *a, reference_points=None, **kw):
                seen.append(dict(inp=output, ref=reference_points, kw=kw))
                return outs[i]
            return layer
        self = types.SimpleNamespace(layers=[make_layer(i) for i in range(nl)], return_intermediate=True)
        with torch.no_grad():
            inter, inter_ref = cls_.forward(self, query, reference_points=ref, valid_ratios=vr, reg_branches=regs,
                                            key=None, value=None, spatial_shapes="passed-through")
        assert all(c["kw"]["spatial_shapes"] == "passed-through" for c in seen)
        flat[f"{tag}/query"], flat[f"{tag}/ref"], flat[f"{tag}/valid_ratios"] = query.numpy(), ref.numpy(), vr.numpy()
        for i in range(nl):
            flat[f"{tag}/layer_out{i}"] = outs[i].numpy()
            flat[f"{tag}/layer_in{i}"], flat[f"{tag}/layer_ref{i}"] = seen[i]["inp"].numpy(), seen[i]["ref"].numpy()
            if refine:
                flat[f"{tag}/reg_w{i}"], flat[f"{tag}/reg_b{i}"] = regs[i].weight.detach().numpy(), regs[i].bias.detach().numpy()
        flat[f"{tag}/inter"], flat[f"{tag}/inter_ref"] = inter.numpy(), inter_ref.numpy()
        print(tag, tuple(inter.shape), tuple(inter_ref.shape))
    np.savez_compressed(os.path.join(OUT, "decoder_loop_cases.npz"), **flat)


def main_bbox2result():
    """tests/golden/bbox2result_cases.npz: the reference's ``bbox2result`` (core/bbox/transforms.py:116-133)."""
    ref = load_reference()
    g = torch.Generator().manual_seed(61)
    flat = {}
    for tag, n in (("some", 23), ("empty", 0)):
        b = torch.rand(n, 5, generator=g) * 100
        l = torch.randint(0, 7, (n,), generator=g)
        out = ref["tr"].bbox2result(b, l, 7)
        flat[f"{tag}/bboxes"], flat[f"{tag}/labels"] = b.numpy(), l.numpy()
        for c, a in enumerate(out):
            assert a.dtype == np.float32
            flat[f"{tag}/out{c}"] = a
    np.savez_compressed(os.path.join(OUT, "bbox2result_cases.npz"), **flat)
    print("bbox2result_cases.npz")


def main_datasplit():
    """tests/golden/data_split_cases.json: the reference's class table and ``split_data_category``
    (mmdet/datasets/data_split.py, loaded by path: it imports nothing of mmdet) on a set of protocols."""
    import contextlib
    import io
    import json
    import random
    spec = importlib.util.spec_from_file_location("ref_data_split", os.path.join(REF, "mmdet/datasets/data_split.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = []
    for split, order, valpart, catofset, seed in [((40, 40), "pingyin", "prev-cur", "train|val|fine", None),
                                                  ((70, 10), "pingyin", "prev-only", "train|val|fine", None),
                                                  ("40-20-20", "pingyin", "cur-only", "train|val", None),
                                                  ((20, 20, 20, 20), "pingyin", "prev-cur", "fine", None),
                                                  ((50, 30), "pingyin", "prev-cur", "val", None),
                                                  ((2, 3, 1), "pingyin", "prev-only", "train", None),
                                                  ((40, 40), "shuffle", "prev-cur", "train|val|fine", 7),
                                                  ((60, 20), "shuffle", "cur-only", "train", 123)]:
        if seed is not None:
            random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            out = mod.split_data_category(dataname="CocoDataset", split=split, order=order, catofset=catofset,
                                          valpart=valpart)
        groups = out if isinstance(out, tuple) else (out,)
        cases.append({"split": split, "order": order, "valpart": valpart, "catofset": catofset, "seed": seed,
                      "out": [[list(d.items()) for d in grp] for grp in groups]})
    with open(os.path.join(OUT, "data_split_cases.json"), "w") as f:
        json.dump({"coco_cats_ids": list(mod.COCO_CATS_IDS.items()), "cases": cases}, f)
    print("data_split_cases.json:", len(cases), "cases")


def load_reference_gfl():
    """The reference's stock GFL pieces (BASELINE configs[4]): AnchorGenerator, anchor utils, ATSSAssigner,
    DistancePointBBoxCoder, AnchorHead.get_anchors and GFLHead's target / loss code, loaded by path on top of
    ``load_reference()``; ConvModule / Scale / BaseDenseHead are dummies (the layers are not run: the fixtures feed
    cls_scores / bbox_preds)."""
    ref = load_reference()

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    sys.modules["mmcv"].is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(x, t) for x in seq)
    cnn = sys.modules["mmcv.cnn"]
    cnn.ConvModule, cnn.Scale = _Dummy, _Dummy
    _pkg("mmdet.core.anchor", os.path.join(REF, "mmdet/core/anchor"))
    _load("mmdet.core.anchor.builder", "mmdet/core/anchor/builder.py")
    ag = _load("mmdet.core.anchor.anchor_generator", "mmdet/core/anchor/anchor_generator.py")
    au = _load("mmdet.core.anchor.utils", "mmdet/core/anchor/utils.py")
    _pkg("mmdet.core.bbox.coder", os.path.join(REF, "mmdet/core/bbox/coder"))
    _load("mmdet.core.bbox.coder.base_bbox_coder", "mmdet/core/bbox/coder/base_bbox_coder.py")
    tr = ref["tr"]
    sys.modules["mmdet.core.bbox.transforms"] = tr
    dp = _load("mmdet.core.bbox.coder.distance_point_bbox_coder", "mmdet/core/bbox/coder/distance_point_bbox_coder.py")
    sys.modules["mmdet.core.bbox.iou_calculators"].build_iou_calculator = lambda cfg: ref["iou"].BboxOverlaps2D()
    atss = _load("mmdet.core.bbox.assigners.atss_assigner", "mmdet/core/bbox/assigners/atss_assigner.py")
    core, misc = sys.modules["mmdet.core"], ref["misc"]
    bb = sys.modules["mmdet.core.bbox.builder"]
    core.anchor_inside_flags, core.images_to_levels, core.unmap = au.anchor_inside_flags, au.images_to_levels, misc.unmap
    core.build_bbox_coder = bb.build_bbox_coder
    core.build_prior_generator = sys.modules["mmdet.core.anchor.builder"].build_prior_generator
    _mod("mmdet.models.dense_heads.base_dense_head", BaseDenseHead=_Dummy)
    _mod("mmdet.models.dense_heads.dense_test_mixins", BBoxTestMixin=type("BBoxTestMixin", (), {}))
    ah = _load("mmdet.models.dense_heads.anchor_head", "mmdet/models/dense_heads/anchor_head.py")
    gh = _load("mmdet.models.dense_heads.gfl_head", "mmdet/models/dense_heads/gfl_head.py")
    ref.update(ag=ag, au=au, dp=dp, atss=atss, ah=ah, gh=gh)
    return ref


def main_gfl():
    """gfl_cases.npz: for two batches (one with a padded image and an image without ground truth) the reference's
    anchors and valid flags, ATSS assignment of image 0, the per-level targets and ``GFLHead.loss`` with gradients."""
    ref = load_reference_gfl()
    GH, AH, L_ = ref["gh"].GFLHead, ref["ah"].AnchorHead, ref["losses"]
    strides = [8, 16, 32, 64, 128]
    out = {}
    for tag, (H, W), shapes_img, n_gts, seed in [("a", (128, 160), [(128, 160), (128, 160)], [3, 2], 1),
                                                  ("b", (160, 224), [(160, 224), (120, 200)], [4, 0], 2)]:
        g = torch.Generator().manual_seed(seed)
        B = len(shapes_img)
        self = types.SimpleNamespace()
        self.num_classes = self.cls_out_channels = 80
        self.reg_max, self.use_sigmoid_cls, self.sampling = 16, True, False
        self.prior_generator = ref["ag"].AnchorGenerator(strides=strides, ratios=[1.0], octave_base_scale=8, scales_per_octave=1)
        self.bbox_coder = ref["dp"].DistancePointBBoxCoder()
        self.train_cfg = types.SimpleNamespace(allowed_border=-1, pos_weight=-1, debug=False)
        self.assigner = ref["atss"].ATSSAssigner(topk=9)
        self.sampler = ref["ps"].PseudoSampler()
        self.integral = GH.__dict__ and ref["gh"].Integral(16)
        self.loss_cls = L_["gfocal_loss"].QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=1.0)
        self.loss_dfl = L_["gfocal_loss"].DistributionFocalLoss(loss_weight=0.25)
        self.loss_bbox = L_["iou_loss"].GIoULoss(loss_weight=2.0)
        for name in ("anchor_center", "loss_single", "loss", "get_targets", "_get_target_single", "get_num_level_anchors_inside"):
            setattr(self, name, types.MethodType(getattr(GH, name), self))
        self.get_anchors = types.MethodType(AH.get_anchors, self)
        fsizes = [(-(-H // s), -(-W // s)) for s in strides]
        metas = [dict(img_shape=(h, w, 3), pad_shape=(H, W, 3)) for h, w in shapes_img]
        cls = [(torch.randn(B, 80, fh, fw, generator=g) * 1.5 - 3).requires_grad_(True) for fh, fw in fsizes]
        box = [torch.randn(B, 68, fh, fw, generator=g).requires_grad_(True) for fh, fw in fsizes]
        gt_b, gt_l = [], []
        for (h, w), n in zip(shapes_img, n_gts):
            xy = torch.rand(n, 2, generator=g) * torch.tensor([0.55 * w, 0.55 * h])
            sz = torch.rand(n, 2, generator=g) * torch.tensor([0.4 * w, 0.4 * h]) + 12
            gt_b.append(torch.cat([xy, xy + sz], 1))
            gt_l.append(torch.randint(0, 80, (n,), generator=g))
        anchors, flags = self.get_anchors(fsizes, metas, device="cpu")
        for lvl in range(5):
            out[f"{tag}/anchors{lvl}"] = anchors[0][lvl].numpy()
            out[f"{tag}/flags{lvl}_img1"] = flags[1][lvl].numpy()
        flat = torch.cat(anchors[0])
        nla = [a.shape[0] for a in anchors[0]]
        res = self.assigner.assign(flat, nla, gt_b[0], None, gt_l[0])
        out[f"{tag}/atss_gt_inds"], out[f"{tag}/atss_labels"] = res.gt_inds.numpy(), res.labels.numpy()
        out[f"{tag}/atss_max_overlaps"] = res.max_overlaps.numpy()
        losses = self.loss(cls, box, gt_b, gt_l, metas)
        total = sum(sum(v) for v in losses.values())
        total.backward()
        for k, v in losses.items():
            out[f"{tag}/loss/{k}"] = torch.stack([x.detach() for x in v]).numpy()
        for lvl in range(5):
            out[f"{tag}/cls{lvl}"], out[f"{tag}/box{lvl}"] = cls[lvl].detach().numpy(), box[lvl].detach().numpy()
            out[f"{tag}/gcls{lvl}"], out[f"{tag}/gbox{lvl}"] = cls[lvl].grad.numpy(), box[lvl].grad.numpy()
        for i in range(B):
            out[f"{tag}/gt_b{i}"], out[f"{tag}/gt_l{i}"] = gt_b[i].numpy(), gt_l[i].numpy()
        out[f"{tag}/img_shapes"] = np.array(shapes_img)
        out[f"{tag}/pad"] = np.array([H, W])
    # bbox coder known answers
    pts = torch.tensor([[10., 12.], [40., 8.]])
    dist = torch.tensor([[3., 4., 5., 6.], [50., 9., 2., 1.]])
    out["coder/decode"] = ref["dp"].DistancePointBBoxCoder().decode(pts, dist, max_shape=(30, 44)).numpy()
    out["coder/encode"] = ref["dp"].DistancePointBBoxCoder().encode(pts, torch.tensor([[2., 3., 30., 40.], [0., 0., 45., 20.]]), 16).numpy()
    np.savez_compressed(os.path.join(OUT, "gfl_cases.npz"), **out)
    print("wrote gfl_cases.npz:", {k: v.shape for k, v in out.items() if "loss/" in k})


if __name__ == "__main__":
    if "--gfl" in sys.argv:
        main_gfl()
        sys.exit(0)
    if "--datasplit" in sys.argv:
        main_datasplit()
    elif "--decode" in sys.argv:
        main_decode()
    elif "--head-forward" in sys.argv:
        main_head_forward()
    elif "--transformer-forward" in sys.argv:
        main_transformer_forward()
    elif "--decoder-loop" in sys.argv:
        main_decoder_loop()
    elif "--bbox2result" in sys.argv:
        main_bbox2result()
    elif "--ragged" in sys.argv:
        main_ragged()
    elif "--two-rank" in sys.argv:
        main_two_rank()
    elif "--variants" in sys.argv:
        main_variants()
    else:
        main()
