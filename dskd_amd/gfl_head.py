"""GFL CNN head + detector for BASELINE.json configs[4]: "GFL R50 (configs/gfl) 40+40 incremental with DSKD
feature-map loss only (CNN-head distillation path)".

The detection part restates the reference's stock GFL:
  ``GFLHead``           /root/reference/mmdet/models/dense_heads/gfl_head.py (Integral :16-50, layers :132-160,
                        forward :162-206, loss_single :220-318, loss :320-393, _get_bboxes_single :395-486,
                        get_targets / _get_target_single :488-640) on ``AnchorHead`` (anchor_head.py:14-130)
  ``AnchorGenerator``   core/anchor/anchor_generator.py:12-420 (the single-anchor case GFL uses: ratios [1.0],
                        octave_base_scale 8, scales_per_octave 1, centre offset 0), ``anchor_inside_flags``
                        (core/anchor/utils.py:21-50)
  ``ATSSAssigner``      core/bbox/assigners/atss_assigner.py:12-179
  ``DistancePointBBoxCoder`` core/bbox/coder/distance_point_bbox_coder.py:8-63 with ``bbox2distance`` /
                        ``distance2bbox`` (core/bbox/transforms.py)
  ``GFL`` detector      models/detectors/gfl.py + single_stage.py
and is pinned by goldens produced by the reference's own classes (tests/golden/gfl_cases.npz, gen_golden.py --gfl).

The reference has NO incremental GFL head (dense_heads/__init__.py:43-46 registers only the stock one), so the
distillation term has no reference implementation: it is the DSKD ``decode_v1`` term
(gfl_deformable_detr_head_il.py:664-718) -- the HIP kernel behind ``native.fgkd_loss`` -- applied to the FPN levels,
with the boxes of the teacher's detections followed by the ground truth, and, in place of the decoder's query
embeddings, each box's feature vector at its centre cell on the pyramid level of its size (teacher and student
pyramids respectively): ``m_j = softmax_c |f_t(box j) - f_s(box j)|`` painted over the box, KL between the masked
teacher and student maps.  The gradient reaches the student's pyramid through those per-box vectors, as it reaches
the decoder queries in the transformer head.  Checked against the CPU restatement (oracle) only."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .bbox import AssignResult, PseudoSampler, bbox_overlaps
from .builder import BBOX_ASSIGNERS, DETECTORS, HEADS, build_assigner, build_loss
from .deformable_detr_il import DeformableDETR_il
from .dist import reduce_mean
from .necks import ConvModule

INF = 100000000


class Scale(nn.Module):
    """ext-mmcv ``Scale``: a learnable scalar factor."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class Integral(nn.Module):
    """gfl_head.py:16-50: expectation of the softmax distribution over {0..reg_max} per box side."""

    def __init__(self, reg_max=16):
        super().__init__()
        self.reg_max = reg_max
        self.register_buffer("project", torch.linspace(0, self.reg_max, self.reg_max + 1))

    def forward(self, x):
        x = F.softmax(x.reshape(-1, self.reg_max + 1), dim=1)
        return F.linear(x, self.project.type_as(x)).reshape(-1, 4)


def distance2bbox(points, distance, max_shape=None):
    """transforms.py ``distance2bbox``: (left, top, right, bottom) distances from a point -> xyxy, clipped."""
    x1, y1 = points[..., 0] - distance[..., 0], points[..., 1] - distance[..., 1]
    x2, y2 = points[..., 0] + distance[..., 2], points[..., 1] + distance[..., 3]
    bboxes = torch.stack([x1, y1, x2, y2], -1)
    if max_shape is not None:
        h, w = float(max_shape[0]), float(max_shape[1])
        lo = bboxes.new_tensor(0.0)
        hi = bboxes.new_tensor([w, h, w, h])
        bboxes = torch.where(bboxes < lo, lo, bboxes)
        bboxes = torch.where(bboxes > hi, hi, bboxes)
    return bboxes


def bbox2distance(points, bbox, max_dis=None, eps=0.1):
    left, top = points[:, 0] - bbox[:, 0], points[:, 1] - bbox[:, 1]
    right, bottom = bbox[:, 2] - points[:, 0], bbox[:, 3] - points[:, 1]
    if max_dis is not None:
        left, top = left.clamp(min=0, max=max_dis - eps), top.clamp(min=0, max=max_dis - eps)
        right, bottom = right.clamp(min=0, max=max_dis - eps), bottom.clamp(min=0, max=max_dis - eps)
    return torch.stack([left, top, right, bottom], -1)


class AnchorGenerator:
    """anchor_generator.py, the configuration GFL uses: one square anchor of side ``octave_base_scale * stride`` per
    cell, centred on the cell's top-left corner (``center_offset = 0``)."""

    def __init__(self, strides, ratios=(1.0,), octave_base_scale=8, scales_per_octave=1, scales=None, base_sizes=None,
                 center_offset=0.0, **kwargs):
        assert list(ratios) == [1.0] and scales_per_octave == 1 and scales is None and center_offset == 0.0, \
            "only the single-anchor configuration of the GFL configs is implemented"
        self.strides = [(int(s), int(s)) if not isinstance(s, (tuple, list)) else tuple(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else list(base_sizes)
        self.scale = float(octave_base_scale)
        self.num_levels = len(self.strides)
        self.num_base_priors = [1] * self.num_levels

    def grid_priors(self, featmap_sizes, device="cpu", dtype=torch.float32):
        out = []
        for (h, w), (sw, sh), bs in zip(featmap_sizes, self.strides, self.base_sizes):
            half = 0.5 * bs * self.scale
            base = torch.tensor([-half, -half, half, half], dtype=dtype, device=device)
            sx = torch.arange(0, int(w), device=device).to(dtype) * sw
            sy = torch.arange(0, int(h), device=device).to(dtype) * sh
            xx, yy = sx.repeat(int(h)), sy.view(-1, 1).repeat(1, int(w)).view(-1)
            shifts = torch.stack([xx, yy, xx, yy], -1)
            out.append(shifts + base[None])
        return out

    def valid_flags(self, featmap_sizes, pad_shape, device="cpu"):
        out = []
        for (fh, fw), (sw, sh) in zip(featmap_sizes, self.strides):
            h, w = pad_shape[:2]
            vh, vw = min(int(math.ceil(h / sh)), int(fh)), min(int(math.ceil(w / sw)), int(fw))
            vx = torch.zeros(int(fw), dtype=torch.bool, device=device)
            vy = torch.zeros(int(fh), dtype=torch.bool, device=device)
            vx[:vw] = True
            vy[:vh] = True
            out.append((vy[:, None] & vx[None, :]).reshape(-1))
        return out


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


@BBOX_ASSIGNERS.register_module()
class ATSSAssigner:
    """atss_assigner.py:12-179: per ground truth the ``topk`` closest anchor centres of every level are candidates;
    positives are the candidates whose IoU reaches mean + std of the candidates' IoUs and whose centre lies inside
    the box; an anchor claimed by several boxes goes to the one with the highest IoU."""

    def __init__(self, topk, iou_calculator=None, ignore_iof_thr=-1):
        self.topk, self.ignore_iof_thr = topk, ignore_iof_thr

    def assign(self, bboxes, num_level_bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        bboxes = bboxes[:, :4]
        num_gt, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        overlaps = bbox_overlaps(bboxes, gt_bboxes)
        assigned_gt_inds = overlaps.new_full((num_bboxes,), 0, dtype=torch.long)
        if num_gt == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            return AssignResult(num_gt, assigned_gt_inds, max_overlaps, labels=labels)
        gt_points = torch.stack(((gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0, (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0), dim=1)
        bboxes_cx, bboxes_cy = (bboxes[:, 0] + bboxes[:, 2]) / 2.0, (bboxes[:, 1] + bboxes[:, 3]) / 2.0
        bboxes_points = torch.stack((bboxes_cx, bboxes_cy), dim=1)
        distances = (bboxes_points[:, None, :] - gt_points[None, :, :]).pow(2).sum(-1).sqrt()
        assert not (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0), \
            "ignore regions are not used by the DSKD configurations"
        candidate_idxs, start = [], 0
        for n in num_level_bboxes:
            end = start + n
            k = min(self.topk, n)
            _, idx = distances[start:end, :].topk(k, dim=0, largest=False)
            candidate_idxs.append(idx + start)
            start = end
        candidate_idxs = torch.cat(candidate_idxs, dim=0)
        candidate_overlaps = overlaps[candidate_idxs, torch.arange(num_gt, device=overlaps.device)]
        thr = candidate_overlaps.mean(0) + candidate_overlaps.std(0)
        is_pos = candidate_overlaps >= thr[None, :]
        candidate_idxs = candidate_idxs + torch.arange(num_gt, device=overlaps.device)[None, :] * num_bboxes
        ep_cx = bboxes_cx.view(1, -1).expand(num_gt, num_bboxes).contiguous().view(-1)
        ep_cy = bboxes_cy.view(1, -1).expand(num_gt, num_bboxes).contiguous().view(-1)
        flat = candidate_idxs.view(-1)
        l_ = ep_cx[flat].view(-1, num_gt) - gt_bboxes[:, 0]
        t_ = ep_cy[flat].view(-1, num_gt) - gt_bboxes[:, 1]
        r_ = gt_bboxes[:, 2] - ep_cx[flat].view(-1, num_gt)
        b_ = gt_bboxes[:, 3] - ep_cy[flat].view(-1, num_gt)
        is_pos = is_pos & (torch.stack([l_, t_, r_, b_], dim=1).min(dim=1)[0] > 0.01)
        overlaps_inf = torch.full_like(overlaps, -INF).t().contiguous().view(-1)
        index = flat[is_pos.view(-1)]
        overlaps_inf[index] = overlaps.t().contiguous().view(-1)[index]
        overlaps_inf = overlaps_inf.view(num_gt, -1).t()
        max_overlaps, argmax_overlaps = overlaps_inf.max(dim=1)
        assigned_gt_inds[max_overlaps != -INF] = argmax_overlaps[max_overlaps != -INF] + 1
        labels = None
        if gt_labels is not None:
            labels = assigned_gt_inds.new_full((num_bboxes,), -1)
            pos = torch.nonzero(assigned_gt_inds > 0, as_tuple=False).squeeze(1)
            if pos.numel() > 0:
                labels[pos] = gt_labels[assigned_gt_inds[pos] - 1]
        return AssignResult(num_gt, assigned_gt_inds, max_overlaps, labels=labels)


def nms(boxes, scores, iou_threshold):
    """Greedy NMS on [n, 4] xyxy boxes (ext-mmcv ``nms`` op): indices kept, by descending score.

    Greedy suppression is the unique solution of keep[i] = not any_{j < i}(keep[j] and iou[j, i] > thr) in score order
    (row i only looks at rows above it).  Iterating that map from all-ones fixes the first t entries after t rounds, so it
    reaches the greedy answer after as many rounds as the longest suppression chain (a handful) -- each round one
    [1, n] x [n, n] product on the device instead of a host loop over the n candidates."""
    if boxes.numel() == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    order = scores.argsort(descending=True)
    sup = (bbox_overlaps(boxes[order], boxes[order]) > iou_threshold).triu(1).float()      # sup[j, i]: j suppresses i
    n = sup.shape[0]
    keep = torch.ones(n, dtype=torch.float32, device=boxes.device)
    for it in range(n):
        new = ((keep[None] @ sup)[0] == 0).float()
        # entries [0, it] are final after round it; compare (one host sync) every fourth round only
        if it % 4 == 3 or it == n - 1:
            if torch.equal(new, keep):
                break
        keep = new
    return order[keep.bool()]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Class-aware NMS (ext-mmcv ``batched_nms``): boxes of different classes never suppress each other."""
    if boxes.numel() == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


@HEADS.register_module()
class GFLHead(nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, conv_cfg=None,
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True), anchor_generator=None,
                 loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0),
                 loss_dfl=dict(type="DistributionFocalLoss", loss_weight=0.25),
                 loss_bbox=dict(type="GIoULoss", loss_weight=2.0), bbox_coder=dict(type="DistancePointBBoxCoder"),
                 reg_max=16, train_cfg=None, test_cfg=None, init_cfg=None, feats_distill="", cates_distill="",
                 loss_fg_feature=None, has_teacher=False, **kwargs):
        super().__init__()
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs, self.reg_max = in_channels, feat_channels, stacked_convs, reg_max
        ag = dict(anchor_generator or dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                           strides=[8, 16, 32, 64, 128]))
        ag.pop("type", None)
        self.prior_generator = AnchorGenerator(**ag)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.assigner = build_assigner(dict(train_cfg["assigner"])) if train_cfg else None
        self.sampler = PseudoSampler()
        self.loss_cls, self.loss_bbox, self.loss_dfl = build_loss(dict(loss_cls)), build_loss(dict(loss_bbox)), build_loss(dict(loss_dfl))
        self.integral = Integral(reg_max)
        self.feats_distill, self.cates_distill, self.has_teacher = feats_distill, cates_distill, has_teacher
        self.loss_fg_feature = build_loss(dict(loss_fg_feature)) if loss_fg_feature else None
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(stacked_convs):
            chn = in_channels if i == 0 else feat_channels
            self.cls_convs.append(ConvModule(chn, feat_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=dict(type="ReLU")))
            self.reg_convs.append(ConvModule(chn, feat_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=dict(type="ReLU")))
        self.gfl_cls = nn.Conv2d(feat_channels, self.cls_out_channels, 3, padding=1)
        self.gfl_reg = nn.Conv2d(feat_channels, 4 * (reg_max + 1), 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.prior_generator.strides])

    def init_weights(self):
        """init_cfg of the reference: Normal(0, 0.01) on every Conv2d, cls bias for p = 0.01."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.gfl_cls.bias, float(-math.log((1 - 0.01) / 0.01)))

    # ------------------------------------------------------------------ forward
    def forward(self, feats, img_metas=None):
        cls_scores, bbox_preds = [], []
        for x, scale in zip(feats, self.scales):
            cls_feat = reg_feat = x
            for conv in self.cls_convs:
                cls_feat = conv(cls_feat)
            for conv in self.reg_convs:
                reg_feat = conv(reg_feat)
            cls_scores.append(self.gfl_cls(cls_feat))
            bbox_preds.append(scale(self.gfl_reg(reg_feat)).float())
        return cls_scores, bbox_preds

    @staticmethod
    def anchor_center(anchors):
        return torch.stack([(anchors[..., 2] + anchors[..., 0]) / 2, (anchors[..., 3] + anchors[..., 1]) / 2], dim=-1)

    def get_anchors(self, featmap_sizes, img_metas, device):
        multi = self.prior_generator.grid_priors(featmap_sizes, device=device)
        anchor_list = [[a for a in multi] for _ in img_metas]
        valid = [self.prior_generator.valid_flags(featmap_sizes, m.get("pad_shape", m["img_shape"]), device) for m in img_metas]
        return anchor_list, valid

    # ------------------------------------------------------------------ targets
    def _get_target_single(self, flat_anchors, valid_flags, num_level_anchors, gt_bboxes, gt_labels, img_meta):
        inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta["img_shape"][:2],
                                     (self.train_cfg or {}).get("allowed_border", -1))
        if not inside.any():
            return (None,) * 7
        anchors = flat_anchors[inside, :]
        inside_per_level = [int(f.sum()) for f in torch.split(inside, num_level_anchors)]
        assign = self.assigner.assign(anchors, inside_per_level, gt_bboxes, None, gt_labels)
        samp = self.sampler.sample(assign, anchors, gt_bboxes)
        n = anchors.shape[0]
        bbox_targets, bbox_weights = torch.zeros_like(anchors), torch.zeros_like(anchors)
        labels = anchors.new_full((n,), self.num_classes, dtype=torch.long)
        label_weights = anchors.new_zeros(n, dtype=torch.float)
        pos_inds, neg_inds = samp.pos_inds, samp.neg_inds
        if len(pos_inds) > 0:
            bbox_targets[pos_inds, :] = samp.pos_gt_bboxes
            bbox_weights[pos_inds, :] = 1.0
            labels[pos_inds] = 0 if gt_labels is None else gt_labels[samp.pos_assigned_gt_inds]
            pw = (self.train_cfg or {}).get("pos_weight", -1)
            label_weights[pos_inds] = 1.0 if pw <= 0 else pw
        if len(neg_inds) > 0:
            label_weights[neg_inds] = 1.0

        def unmap(data, fill=0):
            if data.dim() == 1:
                ret = data.new_full((flat_anchors.size(0),), fill)
                ret[inside] = data
            else:
                ret = data.new_full((flat_anchors.size(0),) + data.size()[1:], fill)
                ret[inside, :] = data
            return ret
        return (unmap(anchors), unmap(labels, self.num_classes), unmap(label_weights), unmap(bbox_targets),
                unmap(bbox_weights), pos_inds, neg_inds)

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas, gt_labels_list):
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        res = [self._get_target_single(torch.cat(anchor_list[i]), torch.cat(valid_flag_list[i]), num_level_anchors,
                                       gt_bboxes_list[i], gt_labels_list[i], img_metas[i]) for i in range(len(img_metas))]
        if any(r[1] is None for r in res):
            return None
        num_total_pos = sum(max(r[5].numel(), 1) for r in res)
        num_total_neg = sum(max(r[6].numel(), 1) for r in res)

        def to_levels(k):          # images_to_levels: [B, sum anchors, ...] -> per level [B, n_l, ...]
            t = torch.stack([r[k] for r in res], 0)
            out, start = [], 0
            for n in num_level_anchors:
                out.append(t[:, start:start + n])
                start += n
            return out
        return to_levels(0), to_levels(1), to_levels(2), to_levels(3), to_levels(4), num_total_pos, num_total_neg

    # ------------------------------------------------------------------ losses
    def loss_single(self, anchors, cls_score, bbox_pred, labels, label_weights, bbox_targets, stride, num_total_samples):
        assert stride[0] == stride[1], "h stride is not equal to w stride!"
        anchors = anchors.reshape(-1, 4)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4 * (self.reg_max + 1))
        bbox_targets, labels, label_weights = bbox_targets.reshape(-1, 4), labels.reshape(-1), label_weights.reshape(-1)
        pos_inds = ((labels >= 0) & (labels < self.num_classes)).nonzero().squeeze(1)
        score = label_weights.new_zeros(labels.shape)
        if len(pos_inds) > 0:
            pos_bbox_pred = bbox_pred[pos_inds]
            pos_centers = self.anchor_center(anchors[pos_inds]) / stride[0]
            weight_targets = cls_score.detach().sigmoid().max(dim=1)[0][pos_inds]
            pos_decode_pred = distance2bbox(pos_centers, self.integral(pos_bbox_pred))
            pos_decode_targets = bbox_targets[pos_inds] / stride[0]
            score[pos_inds] = bbox_overlaps(pos_decode_pred.detach(), pos_decode_targets, is_aligned=True)
            pred_corners = pos_bbox_pred.reshape(-1, self.reg_max + 1)
            target_corners = bbox2distance(pos_centers, pos_decode_targets, self.reg_max).reshape(-1)
            loss_bbox = self.loss_bbox(pos_decode_pred, pos_decode_targets, weight=weight_targets, avg_factor=1.0)
            loss_dfl = self.loss_dfl(pred_corners, target_corners, weight=weight_targets[:, None].expand(-1, 4).reshape(-1),
                                     avg_factor=4.0)
        else:
            loss_bbox = bbox_pred.sum() * 0
            loss_dfl = bbox_pred.sum() * 0
            weight_targets = bbox_pred.new_tensor(0)
        loss_cls = self.loss_cls(cls_score, (labels, score), weight=label_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox, loss_dfl, weight_targets.sum()

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        assert len(featmap_sizes) == self.prior_generator.num_levels
        device = cls_scores[0].device
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device)
        targets = self.get_targets(anchor_list, valid_flag_list, gt_bboxes, img_metas, gt_labels)
        if targets is None:
            return None
        anchors, labels, label_weights, bbox_targets, _, num_total_pos, _ = targets
        num_total_samples = max(reduce_mean(torch.tensor(num_total_pos, dtype=torch.float, device=device)).item(), 1.0)
        out = [self.loss_single(a, c, b, l, w, t, s, num_total_samples)
               for a, c, b, l, w, t, s in zip(anchors, cls_scores, bbox_preds, labels, label_weights, bbox_targets,
                                              self.prior_generator.strides)]
        avg_factor = reduce_mean(sum(o[3] for o in out)).clamp_(min=1).item()
        return dict(loss_cls=[o[0] for o in out], loss_bbox=[o[1] / avg_factor for o in out],
                    loss_dfl=[o[2] / avg_factor for o in out])

    # ------------------------------------------------------------------ the DSKD feature-map term on the pyramid
    def box_vectors(self, feats, boxes, img_metas):
        """One 256-vector per box: the pyramid feature at the box's centre cell on the level of its size (FPN's
        assignment rule floor(4 + log2(sqrt(wh) / 224)) mapped onto the levels at hand).  Differentiable w.r.t. feats."""
        nl = len(feats)
        out = []
        for i, bx in enumerate(boxes):
            if bx.shape[0] == 0:
                out.append(feats[0].new_zeros((0, feats[0].shape[1])))
                continue
            img_h, img_w = img_metas[i]["img_shape"][:2]
            wh = (bx[:, 2:] - bx[:, :2]).clamp(min=1.0)
            lvl = torch.floor(4 + torch.log2(torch.sqrt(wh[:, 0] * wh[:, 1]) / 224.0 + 1e-6)).clamp(2, 2 + nl - 1).long() - 2
            cx, cy = (bx[:, 0] + bx[:, 2]) * 0.5, (bx[:, 1] + bx[:, 3]) * 0.5
            vecs = feats[0].new_zeros((bx.shape[0], feats[0].shape[1]))
            for l in range(nl):
                sel = (lvl == l).nonzero().squeeze(1)
                if sel.numel() == 0:
                    continue
                H, W = feats[l].shape[-2:]
                xi = (cx[sel] / img_w * W).floor().clamp(0, W - 1).long()
                yi = (cy[sel] / img_h * H).floor().clamp(0, H - 1).long()
                vecs = vecs.index_put((sel,), feats[l][i, :, yi, xi].t().to(vecs.dtype))
            out.append(vecs)
        return torch.cat(out, 0)

    def fg_feature_loss(self, feats_s, teacher_info, gt_bboxes, img_metas):
        """``decode_v1`` on the pyramid (module docstring): boxes = teacher detections then ground truth."""
        feats_t = teacher_info["neck_feats"]
        boxes = [torch.cat([teacher_info["pred_bboxes"][i].reshape(-1, 4).to(gt_bboxes[i]), gt_bboxes[i].reshape(-1, 4)], 0)
                 for i in range(len(img_metas))]
        M = sum(int(b.shape[0]) for b in boxes)
        v_s = self.box_vectors(feats_s, boxes, img_metas).float()
        with torch.no_grad():
            v_t = self.box_vectors(feats_t, boxes, img_metas).float()
        dev = v_s.device
        # pair k of the kernel = (teacher row keepid[k], k-th student row labelled "previous"): row k with row k
        keep = torch.arange(M, device=dev)
        labels = torch.zeros(M, dtype=torch.long, device=dev)
        prev = torch.ones(1, dtype=torch.bool, device=dev)
        shapes = [tuple(int(v) for v in m["img_shape"][:2]) for m in img_metas]
        return native.fgkd_loss([f.float() for f in feats_s], [f.float() for f in feats_t], boxes, shapes, v_t, keep, v_s,
                                labels, prev, float(self.loss_fg_feature.T), float(self.loss_fg_feature.loss_weight))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      teacher_info=None, task_labels=None, **kwargs):
        outs = self.forward(x)
        losses = self.loss(*outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore)
        if losses is None:         # no valid anchor in some image (get_targets returned None, as in the reference's loss())
            return None
        if self.has_teacher and teacher_info and teacher_info.get("neck_feats") is not None and \
                "fg_info" in self.feats_distill and "decode_v1" in self.feats_distill:
            losses["loss_fg_feature"] = self.fg_feature_loss(x, teacher_info, gt_bboxes, img_metas)
        return losses

    # ------------------------------------------------------------------ decode
    def get_bboxes(self, cls_scores, bbox_preds, score_factors=None, img_metas=None, cfg=None, rescale=False, with_nms=True,
                   need_logits=False, **kwargs):
        """``BaseDenseHead.get_bboxes`` + ``_get_bboxes_single`` (:395-486) + ``_bbox_post_process``: per image
        (det_bboxes [n, 5], det_labels [n]) and, with ``need_logits``, the sigmoid rows and the flat prior index of
        the kept detections (what ``out_teacher`` hands to the distillation terms)."""
        from .gfl_deformable_detr_head_il import filter_scores_and_topk
        cfg = dict(self.test_cfg or {}) if cfg is None else dict(cfg)
        featmap_sizes = [c.shape[-2:] for c in cls_scores]
        priors = self.prior_generator.grid_priors(featmap_sizes, device=cls_scores[0].device)
        results = []
        for i, meta in enumerate(img_metas):
            bb, sc, lb, lg, ids = [], [], [], [], []
            start = 0
            for lvl, (cs, bp, stride, pri) in enumerate(zip(cls_scores, bbox_preds, self.prior_generator.strides, priors)):
                pred = self.integral(bp[i].permute(1, 2, 0)) * stride[0]
                scores = cs[i].permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
                s, l, keep_idx, filt = filter_scores_and_topk(scores, cfg.get("score_thr", 0.05), cfg.get("nms_pre", -1),
                                                              dict(bbox_pred=pred, priors=pri, rows=scores))
                bb.append(distance2bbox(self.anchor_center(filt["priors"]), filt["bbox_pred"], max_shape=meta["img_shape"]))
                sc.append(s)
                lb.append(l)
                lg.append(filt["rows"])
                ids.append(keep_idx + start)
                start += pri.shape[0]
            bb, sc, lb, lg, ids = torch.cat(bb), torch.cat(sc), torch.cat(lb), torch.cat(lg), torch.cat(ids)
            if rescale:
                bb = bb / bb.new_tensor(meta["scale_factor"])
            if with_nms and bb.numel() > 0:
                keep = batched_nms(bb, sc, lb, dict(cfg.get("nms") or {}).get("iou_threshold", 0.6))[:cfg.get("max_per_img", 100)]
                bb, sc, lb, lg, ids = bb[keep], sc[keep], lb[keep], lg[keep], ids[keep]
            det = torch.cat([bb, sc[:, None]], -1)
            results.append((det, lb, lg, ids) if need_logits else (det, lb))
        return results

    def simple_test(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats), img_metas=img_metas, rescale=rescale)


@DETECTORS.register_module()
class GFL(DeformableDETR_il):
    """``GFL`` (models/detectors/gfl.py: a plain single-stage detector) with the teacher plumbing of the incremental
    detector (``set_teacher`` / ``out_teacher`` / ``teacher_info``), so that tools/train_increment.py drives it like
    the transformer detector."""

    # out_teacher and the one-batch-ahead pipeline (TeacherAhead: the teacher forward as a hipGraph replay on a second stream,
    # its decode + NMS -- host round trips -- waiting for that stream only) are the transformer detector's; the head's kind
    # enters through these two
    def _teacher_heads(self, feats, img_metas):
        return self.teacher_model.bbox_head.forward(feats)

    @staticmethod
    def _keepid_stride(head_outs):
        return sum(int(c.shape[-2] * c.shape[-1]) for c in head_outs[0])      # priors per image (one per location)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, teacher_info=None):
        for m in img_metas:
            m.setdefault("batch_input_shape", tuple(img.size()[-2:]))
        if teacher_info is None and self.has_teacher:
            feats, outs, keepid, logits, labels, scores, bboxes = self.out_teacher(img, img_metas)
            teacher_info = {"neck_feats": feats, "head_outs": outs, "pred_keepid": keepid, "pred_logits": logits,
                            "pred_scores": scores, "pred_labels": labels, "pred_bboxes": bboxes}
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, teacher_info=teacher_info,
                                            task_labels=self.LableInPCNTask)

    def simple_test(self, img, img_metas, rescale=False):
        from .bbox import bbox2result
        res = self.bbox_head.simple_test(self.extract_feat(img), img_metas, rescale=rescale)
        return [bbox2result(b, l, self.bbox_head.num_classes) for b, l in res]
