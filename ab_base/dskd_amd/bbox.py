"""Box utilities, match costs, the GFL Hungarian assigner and the pseudo sampler, under the
reference's registry names.

Restated from:
  /root/reference/mmdet/core/bbox/transforms.py:245-270            cxcywh <-> xyxy
  /root/reference/mmdet/core/bbox/iou_calculators/iou2d_calculator.py:75-261  bbox_overlaps
  /root/reference/mmdet/core/bbox/match_costs/match_cost.py:10-51, :151-272, :437-476
  /root/reference/mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:16-160
  /root/reference/mmdet/core/bbox/assigners/assign_result.py:43-50
  /root/reference/mmdet/core/bbox/samplers/pseudo_sampler.py:24-41, sampling_result.py:26-52

The assigner has two entries: ``assign`` (the reference's per-image signature) and
``assign_batch`` -- all (decoder layer, image) problems of a step in two launches: the fused
HIP cost kernel and the batched on-device LSAP, with no device->host round trip (the
reference pays one per problem, gfl_hungarian_assigner.py:143-151).
"""
import torch

from . import native
from .utils import device_const
from .builder import BBOX_ASSIGNERS, BBOX_SAMPLERS, MATCH_COST, build_match_cost


def bbox_cxcywh_to_xyxy(bbox):
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def bbox_xyxy_to_cxcywh(bbox):
    x1, y1, x2, y2 = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1), (y2 - y1)], dim=-1)


def bbox2result(bboxes, labels, num_classes):
    """/root/reference/mmdet/core/bbox/transforms.py:116-133 -- detections [n, 5] + labels [n] to the evaluation
    format: one float32 numpy array [n_c, 5] per class."""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def bbox_overlaps(bboxes1, bboxes2, mode="iou", is_aligned=False, eps=1e-6):
    assert mode in ["iou", "iof", "giou"], f"Unsupported mode {mode}"
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    batch_shape = bboxes1.shape[:-2]
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(batch_shape + ((rows,) if is_aligned else (rows, cols)))
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    if is_aligned:
        lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap if mode in ["iou", "giou"] else area1
        if mode == "giou":
            enclosed_lt = torch.min(bboxes1[..., :2], bboxes2[..., :2])
            enclosed_rb = torch.max(bboxes1[..., 2:], bboxes2[..., 2:])
    else:
        lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap if mode in ["iou", "giou"] else area1[..., None]
        if mode == "giou":
            enclosed_lt = torch.min(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
            enclosed_rb = torch.max(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    union = union.clamp(min=eps)           # == torch.max(union, eps) of the reference, without a H2D copy
    ious = overlap / union
    if mode in ["iou", "iof"]:
        return ious
    enclose_wh = (enclosed_rb - enclosed_lt).clamp(min=0)
    enclose_area = (enclose_wh[..., 0] * enclose_wh[..., 1]).clamp(min=eps)
    return ious - (enclose_area - union) / enclose_area


# ------------------------------------------------------------------ match costs
@MATCH_COST.register_module()
class BBoxL1Cost:
    def __init__(self, weight=1., box_format="xyxy"):
        assert box_format in ["xyxy", "xywh"]
        self.weight, self.box_format = weight, box_format

    def __call__(self, bbox_pred, gt_bboxes):
        if self.box_format == "xywh":
            gt_bboxes = bbox_xyxy_to_cxcywh(gt_bboxes)
        else:
            bbox_pred = bbox_cxcywh_to_xyxy(bbox_pred)
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module()
class IoUCost:
    def __init__(self, iou_mode="giou", weight=1.):
        self.weight, self.iou_mode = weight, iou_mode

    def __call__(self, bboxes, gt_bboxes):
        return -bbox_overlaps(bboxes, gt_bboxes, mode=self.iou_mode, is_aligned=False) * self.weight


@MATCH_COST.register_module()
class QualityFocalLossCost:
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12, iou_mode="giou", beta=2.0, binary_input=False):
        assert not binary_input, "mask input is not on the DSKD path"
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps
        self.iou_mode, self.beta, self.binary_input = iou_mode, beta, binary_input

    def __call__(self, cls_pred, gt_labels, bboxes, gt_bboxes):
        """match_cost.py:193-230: BCE(logit, IoU) * |IoU - sigmoid|^beta at the GT classes."""
        import torch.nn.functional as F
        pred_sigmoid = cls_pred.sigmoid()
        score = bbox_overlaps(bboxes, gt_bboxes)
        scale_factor = score - pred_sigmoid[:, gt_labels]
        cls_cost = F.binary_cross_entropy_with_logits(cls_pred[:, gt_labels], score, reduction="none") \
            * scale_factor.abs().pow(self.beta)
        return cls_cost * self.weight


# ------------------------------------------------------------------ assign / sample
class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    """sampling_result.py:26-52."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, *args, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@BBOX_ASSIGNERS.register_module()
class GFLHungarianAssigner:
    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.),
                 reg_cost=dict(type="BBoxL1Cost", weight=1.0), iou_cost=dict(type="IoUCost", iou_mode="giou", weight=1.0),
                 dfl_cost=None, num_classes=80, reg_max=16):
        self.cls_cost = build_match_cost(dict(cls_cost))
        self.reg_cost = build_match_cost(dict(reg_cost))
        self.iou_cost = build_match_cost(dict(iou_cost))
        self.num_classes, self.reg_max = num_classes, reg_max

    def _fusable(self):
        return (isinstance(self.cls_cost, QualityFocalLossCost) and self.cls_cost.beta == 2.0
                and isinstance(self.reg_cost, BBoxL1Cost) and self.reg_cost.box_format == "xywh"
                and isinstance(self.iou_cost, IoUCost) and self.iou_cost.iou_mode == "giou")

    def cost_matrix(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta):
        """gfl_hungarian_assigner.py:120-140 (generic composition of the three costs)."""
        img_h, img_w, _ = img_meta["img_shape"]
        factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h]).unsqueeze(0)
        normalize_gt_bboxes = gt_bboxes / factor
        reg_cost = self.reg_cost(bbox_pred, normalize_gt_bboxes)
        iou_cost = self.iou_cost(bbox_cxcywh_to_xyxy(bbox_pred) * factor, gt_bboxes)
        cls_cost = self.cls_cost(cls_pred, gt_labels, bbox_cxcywh_to_xyxy(bbox_pred), normalize_gt_bboxes)
        return cls_cost + reg_cost + iou_cost

    def assign_batch(self, bbox_preds, cls_preds, gt_bboxes_list, gt_labels_list, img_metas):
        """All problems at once.  bbox_preds [P,Q,4], cls_preds [P,Q,C]; problem p uses image
        p % len(img_metas).  Returns (assigned_gt_inds [P,Q] (0 = background, k = 1-based gt),
        assigned_labels [P,Q] (-1 where unmatched), status) -- all on the device."""
        P, Q, _ = bbox_preds.shape
        nimg = len(img_metas)
        gts = [gt_bboxes_list[p % nimg] for p in range(P)]
        labs = [gt_labels_list[p % nimg] for p in range(P)]
        G = [int(g.shape[0]) for g in gts]
        gt_start = [0]
        for g in G:
            gt_start.append(gt_start[-1] + g)
        dev = bbox_preds.device
        gt_inds = torch.zeros((P, Q), dtype=torch.long, device=dev)
        labels = torch.full((P, Q), -1, dtype=torch.long, device=dev)
        if gt_start[-1] == 0 or Q == 0:
            return gt_inds, labels, None
        gt_cat = torch.cat([g.reshape(-1, 4) for g in gts], 0)
        lab_cat = torch.cat([l.reshape(-1) for l in labs], 0)
        wh = [(float(img_metas[p % nimg]["img_shape"][1]), float(img_metas[p % nimg]["img_shape"][0])) for p in range(P)]
        if self._fusable():
            cost = native.match_cost(bbox_preds, cls_preds, gt_cat, lab_cat, gt_start, wh, self.cls_cost.weight,
                                     self.reg_cost.weight, self.iou_cost.weight)
        else:
            parts = [self.cost_matrix(bbox_preds[p].detach(), cls_preds[p].detach(), gts[p], labs[p],
                                      img_metas[p % nimg]).reshape(-1) for p in range(P) if G[p] > 0]
            cost = torch.cat(parts)
        offsets = [Q * s for s in gt_start[:-1]]
        if cost.is_cuda:
            row, col, outs, status = native.lsap_batched(cost, [Q] * P, G, offsets)
        else:
            row, col, outs, status = _lsap_cpu_checker(cost, [Q] * P, G, offsets)
        # scatter the matches: one vectorised index_put for all problems
        n_match = [min(Q, g) for g in G]
        total = sum(n_match)
        prob_of = device_const([p for p in range(P) for _ in range(n_match[p])], torch.long, dev)
        start_of = device_const([gt_start[p] for p in range(P) for _ in range(n_match[p])], torch.long, dev)
        row, col = row[:total], col[:total]
        gt_inds[prob_of, row] = col + 1
        labels[prob_of, row] = lab_cat[start_of + col]
        return gt_inds, labels, status

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, bbox_lrtb, img_meta, gt_bboxes_ignore=None, eps=1e-7):
        """The reference's per-image entry (gfl_hungarian_assigner.py:59-160)."""
        assert gt_bboxes_ignore is None, "Only case when gt_bboxes_ignore is None is supported."
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        if num_gts == 0 or num_bboxes == 0:
            assigned_gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
            assigned_labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            return AssignResult(num_gts, assigned_gt_inds, None, labels=assigned_labels)
        gt_inds, labels, status = self.assign_batch(bbox_pred[None], cls_pred[None], [gt_bboxes], [gt_labels], [img_meta])
        if status is not None:
            native.raise_for_lsap_status(status)
        return AssignResult(num_gts, gt_inds[0], None, labels=labels[0])


def _lsap_cpu_checker(cost, nr, nc, offsets):
    """CPU tensors reach here only in tests / the CPU-baseline leg (checker injected)."""
    chk = native.cpu_checker()
    if chk is None:
        raise native.NativeError("assignment on CPU tensors: the HIP path needs GPU tensors (no CPU fallback)")
    rows, cols, outs, acc = [], [], [], 0
    for r, c, off in zip(nr, nc, offsets):
        outs.append(acc)
        if r and c:
            ri, ci = chk.lsap(cost[off:off + r * c].view(r, c))
            rows.append(ri)
            cols.append(ci)
            acc += min(r, c)
    row = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long)
    col = torch.cat(cols) if cols else torch.zeros(0, dtype=torch.long)
    return row, col, outs, None
