"""BASELINE.json configs[4]: the stock GFL head (anchors, ATSS, targets, QFL / DFL / GIoU) against outputs of the
reference's own classes (tests/golden/gfl_cases.npz, gen_golden.py --gfl), the config against the reference's
configs/gfl/gfl_r50_fpn_1x_coco.py, and the DSKD feature-map term on the pyramid through a whole CPU step."""
import copy
import os

import numpy as np
import pytest
import torch

import dskd_amd  # noqa: F401
from dskd_amd import gfl_head as G
from dskd_amd.builder import build_detector
from dskd_amd.config import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z = np.load(os.path.join(ROOT, "tests", "golden", "gfl_cases.npz"))
CFG = os.path.join(ROOT, "configs", "dskd_gfl_r50_fpn_40_40.py")
REF_CFG = "/root/reference/configs/gfl/gfl_r50_fpn_1x_coco.py"
t = torch.from_numpy


def _head():
    cfg = Config.fromfile(CFG)
    hc = dict(cfg.model.bbox_head)
    hc.pop("type")
    return G.GFLHead(train_cfg=dict(cfg.model.train_cfg), test_cfg=dict(cfg.model.test_cfg), **hc)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_anchors_and_atss_vs_reference(tag):
    """AnchorGenerator.grid_priors / valid_flags (anchor_generator.py) and ATSSAssigner.assign (atss_assigner.py:40-179):
    equal anchors and flags, equal assigned ground truth / labels / overlaps."""
    H, W = (int(v) for v in Z[f"{tag}/pad"])
    strides = [8, 16, 32, 64, 128]
    ag = G.AnchorGenerator(strides=strides, ratios=[1.0], octave_base_scale=8, scales_per_octave=1)
    fsizes = [(-(-H // s), -(-W // s)) for s in strides]
    anchors = ag.grid_priors(fsizes)
    shp1 = tuple(int(v) for v in Z[f"{tag}/img_shapes"][1])
    flags = ag.valid_flags(fsizes, (H, W, 3))
    for lvl in range(5):
        assert torch.equal(anchors[lvl], t(Z[f"{tag}/anchors{lvl}"]))
        assert torch.equal(flags[lvl], t(Z[f"{tag}/flags{lvl}_img1"]))
    del shp1
    res = G.ATSSAssigner(topk=9).assign(torch.cat(anchors), [a.shape[0] for a in anchors], t(Z[f"{tag}/gt_b0"]), None,
                                        t(Z[f"{tag}/gt_l0"]))
    assert torch.equal(res.gt_inds, t(Z[f"{tag}/atss_gt_inds"])) and torch.equal(res.labels, t(Z[f"{tag}/atss_labels"]))
    torch.testing.assert_close(res.max_overlaps, t(Z[f"{tag}/atss_max_overlaps"]), rtol=1e-6, atol=1e-7)
    assert int((res.gt_inds > 0).sum()) > 0


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gfl_loss_vs_reference(tag):
    """GFLHead.loss (gfl_head.py:320-393 with get_targets / _get_target_single / loss_single): per-level loss_cls /
    loss_bbox / loss_dfl and the gradients w.r.t. every level's cls_score / bbox_pred, incl. a padded image whose
    anchors beyond the valid region are dropped and an image without ground truth."""
    head = _head()
    cls = [t(Z[f"{tag}/cls{l}"]).clone().requires_grad_(True) for l in range(5)]
    box = [t(Z[f"{tag}/box{l}"]).clone().requires_grad_(True) for l in range(5)]
    B = cls[0].shape[0]
    H, W = (int(v) for v in Z[f"{tag}/pad"])
    metas = [dict(img_shape=(int(h), int(w), 3), pad_shape=(H, W, 3)) for h, w in Z[f"{tag}/img_shapes"]]
    gt_b = [t(Z[f"{tag}/gt_b{i}"]) for i in range(B)]
    gt_l = [t(Z[f"{tag}/gt_l{i}"]) for i in range(B)]
    losses = head.loss(cls, box, gt_b, gt_l, metas)
    assert sorted(losses) == ["loss_bbox", "loss_cls", "loss_dfl"]
    for k, v in losses.items():
        torch.testing.assert_close(torch.stack([x.detach() for x in v]), t(Z[f"{tag}/loss/{k}"]), rtol=1e-5, atol=1e-7)
    sum(sum(v) for v in losses.values()).backward()
    for l in range(5):
        torch.testing.assert_close(cls[l].grad, t(Z[f"{tag}/gcls{l}"]), rtol=1e-4, atol=1e-8)
        torch.testing.assert_close(box[l].grad, t(Z[f"{tag}/gbox{l}"]), rtol=1e-4, atol=1e-8)


def test_distance_point_coder_vs_reference():
    pts = torch.tensor([[10., 12.], [40., 8.]])
    dist = torch.tensor([[3., 4., 5., 6.], [50., 9., 2., 1.]])
    torch.testing.assert_close(G.distance2bbox(pts, dist, max_shape=(30, 44)), t(Z["coder/decode"]))
    torch.testing.assert_close(G.bbox2distance(pts, torch.tensor([[2., 3., 30., 40.], [0., 0., 45., 20.]]), 16), t(Z["coder/encode"]))


def test_config_is_the_references_gfl_model():
    if not os.path.isfile(REF_CFG):
        pytest.skip("reference tree not present")

    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x
    ref, own = plain(Config.fromfile(REF_CFG).model), plain(Config.fromfile(CFG).model)
    ref["backbone"]["init_cfg"] = own["backbone"]["init_cfg"] = None
    for k in ("feats_distill", "loss_fg_feature"):           # this repo's distillation keys (no reference IL GFL head)
        own["bbox_head"].pop(k)
    own.pop("teacher_test_cfg")
    assert ref == own


def test_fpn_matches_its_definition_and_reference_names():
    """fpn.py:140-204 with add_extra_convs='on_output', start_level=1: five outputs, P6 / P7 from stride-2 convs on the
    previous OUTPUT; parameter names as in a reference checkpoint."""
    from dskd_amd.necks import FPN
    torch.manual_seed(0)
    fpn = FPN([8, 16, 32, 64], 12, num_outs=5, start_level=1, add_extra_convs="on_output")
    xs = [torch.randn(2, c, s, s + 2) for c, s in zip([8, 16, 32, 64], [32, 16, 8, 4])]
    outs = fpn(xs)
    assert [tuple(o.shape[1:]) for o in outs] == [(12, 16, 18), (12, 8, 10), (12, 4, 6), (12, 2, 3), (12, 1, 2)]
    import torch.nn.functional as F
    lat = [fpn.lateral_convs[i].conv(xs[i + 1]) for i in range(3)]
    lat[1] = lat[1] + F.interpolate(lat[2], size=lat[1].shape[2:], mode="nearest")
    lat[0] = lat[0] + F.interpolate(lat[1], size=lat[0].shape[2:], mode="nearest")
    want = [fpn.fpn_convs[i].conv(lat[i]) for i in range(3)]
    want.append(fpn.fpn_convs[3].conv(want[-1]))
    want.append(fpn.fpn_convs[4].conv(want[-1]))
    for a, b in zip(outs, want):
        torch.testing.assert_close(a, b)
    names = set(fpn.state_dict())
    assert {"lateral_convs.0.conv.weight", "lateral_convs.2.conv.bias", "fpn_convs.4.conv.weight"} <= names


def test_gfl_distillation_step_cpu(cpu_ops):
    """One teacher + student step of the GFL detector built from the config (reduced input): detection losses + the
    DSKD feature-map term on the five pyramid levels; the term's gradient reaches the student's pyramid through the
    per-box vectors and vanishes when teacher == student."""
    cfg = Config.fromfile(CFG)
    torch.manual_seed(0)
    m = build_detector(cfg.model)
    m.init_weights()
    m.set_teacher(model=copy.deepcopy(m))
    m.LableInPCNTask = {"prev": list(range(40)), "curr": list(range(40, 80)), "next": []}
    m.train()
    B, H, W = 2, 128, 160
    img = torch.randn(B, 3, H, W)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0) for _ in range(B)]
    gt_b = [torch.tensor([[10., 12., 90., 100.], [30., 20., 120., 110.]]), torch.tensor([[5., 5., 100., 90.]])]
    gt_l = [torch.tensor([45, 71]), torch.tensor([79])]
    feats, outs, *_ = m.out_teacher(img, metas)
    ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=None, pred_logits=None, pred_scores=None, pred_labels=None,
              pred_bboxes=[torch.tensor([[20., 20., 100., 90.]]), torch.tensor([[40., 40., 120., 100.], [0., 0., 50., 60.]])])
    lv = m.train_step(dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=ti))["log_vars"]
    assert {"loss_cls", "loss_bbox", "loss_dfl", "loss_fg_feature", "loss"} <= set(lv)
    base = lv["loss_fg_feature"]                  # identical teacher: only the fp32 noise of the CPU evaluation is left
    assert abs(base) < 1e-3
    with torch.no_grad():
        for p in m.teacher_model.parameters():
            p.add_(torch.randn_like(p) * 5e-2)
    feats, outs, *_ = m.out_teacher(img, metas)
    out = m.train_step(dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=dict(ti, neck_feats=feats)))
    assert out["log_vars"]["loss_fg_feature"] > 10 * abs(base) + 1e-4, (out["log_vars"]["loss_fg_feature"], base)
    x = m.extract_feat(img)
    term = m.bbox_head.fg_feature_loss(x, dict(ti, neck_feats=feats), gt_b, metas)
    g = torch.autograd.grad(term, m.neck.fpn_convs[0].conv.weight)[0]
    assert float(g.abs().sum()) > 0
    # inference surface: per image one [n_c, 5] array per class
    m.eval()
    res = m.simple_test(img, metas)
    assert len(res) == B and len(res[0]) == 80 and res[0][0].shape[1] == 5


def test_nms_fixed_point_equals_greedy_suppression():
    """``gfl_head.nms`` (device-side fixed-point rounds) against the sequential definition of greedy NMS (ext-mmcv ``nms``,
    called from mmdet/core/post_processing/bbox_nms.py:multiclass_nms): random clusters, two thresholds, and a chain
    box_k -> box_k+1 whose answer needs as many rounds as the chain is long."""
    from dskd_amd.gfl_head import nms, batched_nms, bbox_overlaps

    def greedy(boxes, scores, thr):
        order = scores.argsort(descending=True)
        iou = bbox_overlaps(boxes[order], boxes[order])
        keep = torch.ones(len(order), dtype=torch.bool)
        for i in range(len(order)):
            if keep[i]:
                keep[i + 1:] &= ~(iou[i, i + 1:] > thr)
        return order[keep]

    g = torch.Generator().manual_seed(0)
    for n in (1, 2, 7, 50, 300):
        for spread in (20.0, 200.0):
            xy = torch.rand(n, 2, generator=g) * spread
            b = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 30 + 5], 1)
            s = torch.rand(n, generator=g)
            for thr in (0.3, 0.6):
                assert torch.equal(nms(b, s, thr), greedy(b, s, thr))
    chain = torch.stack([torch.tensor([k * 6.0, 0.0, k * 6.0 + 10.0, 10.0]) for k in range(40)])
    s = torch.linspace(1.0, 0.1, 40)
    kept = nms(chain, s, 0.2)
    assert torch.equal(kept, greedy(chain, s, 0.2)) and len(kept) == 20
    assert nms(chain[:0], s[:0], 0.5).numel() == 0
    # class-aware: identical boxes of different classes never suppress each other
    same = chain[:1].repeat(3, 1)
    assert len(batched_nms(same, torch.tensor([0.9, 0.8, 0.7]), torch.tensor([0, 1, 0]), 0.5)) == 2
