"""CPU restatement of the matching cost + Hungarian assignment (TEST INFRASTRUCTURE).

Follows, line by line in meaning (not in text):
  mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:120-158   (cost build, LSA, index fill)
  mmdet/core/bbox/match_costs/match_cost.py:34-51    BBoxL1Cost('xywh')
  mmdet/core/bbox/match_costs/match_cost.py:193-230  QualityFocalLossCost
  mmdet/core/bbox/match_costs/match_cost.py:460-476  IoUCost('giou')
  mmdet/core/bbox/iou_calculators/iou2d_calculator.py:190-261  bbox_overlaps
  mmdet/core/bbox/transforms.py:245-270              cxcywh <-> xyxy
Pinned by tests/golden/assign_*.npz (outputs of the reference's own classes) and by the
reference test known answer tests/test_metrics/test_box_overlap.py:92-106.
"""
import numpy as np
import torch
import torch.nn.functional as F


def cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def xyxy_to_cxcywh(b):
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


def overlaps(a, b, mode="iou", aligned=False, eps=1e-6):
    """bbox_overlaps restated (iou2d_calculator.py:190-261)."""
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    if aligned:
        lt, rb = torch.max(a[..., :2], b[..., :2]), torch.min(a[..., 2:], b[..., 2:])
        elt, erb = torch.min(a[..., :2], b[..., :2]), torch.max(a[..., 2:], b[..., 2:])
        union_base = area_a + area_b
    else:
        lt = torch.max(a[..., :, None, :2], b[..., None, :, :2])
        rb = torch.min(a[..., :, None, 2:], b[..., None, :, 2:])
        elt = torch.min(a[..., :, None, :2], b[..., None, :, :2])
        erb = torch.max(a[..., :, None, 2:], b[..., None, :, 2:])
        union_base = area_a[..., None] + area_b[..., None, :]
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    e = inter.new_tensor([eps])
    union = torch.max(union_base - inter, e)
    iou = inter / union
    if mode == "iou":
        return iou
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return iou - (earea - union) / earea


def cost_matrix(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_w, img_h, w_cls=2.0, w_reg=5.0, w_iou=2.0):
    """[Q,G] cost of GFLHungarianAssigner.assign (gfl_hungarian_assigner.py:120-140)."""
    factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h]).unsqueeze(0)
    gt_norm = gt_bboxes / factor
    reg = torch.cdist(bbox_pred, xyxy_to_cxcywh(gt_norm), p=1) * w_reg
    boxes_px = cxcywh_to_xyxy(bbox_pred) * factor
    iou_c = -overlaps(boxes_px, gt_bboxes, mode="giou") * w_iou
    score = overlaps(cxcywh_to_xyxy(bbox_pred), gt_norm, mode="iou")
    logits = cls_pred[:, gt_labels]
    cls = F.binary_cross_entropy_with_logits(logits, score, reduction="none") * \
        (score - logits.sigmoid()).abs().pow(2.0) * w_cls
    return cls + reg + iou_c


def assign(cost, lsa):
    """(row, col) of linear_sum_assignment on the detached CPU cost (…assigner.py:143-151)."""
    r, c = lsa(cost.detach().cpu().numpy())
    return np.asarray(r, dtype=np.int64), np.asarray(c, dtype=np.int64)
