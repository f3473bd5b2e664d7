"""Host-side mirror of the reference's plugin surface: registry, config loading (the
reference's own config files when /root/reference is present), detector state handling,
optimizer grouping, LR schedule, the runner, and BASELINE config #1 (CPU plumbing)."""
import copy
import os

import pytest
import torch

import dskd_amd
from dskd_amd import builder
from dskd_amd.config import Config
from dskd_amd.datasets import SyntheticILDataset, build_dataloader
from dskd_amd.runner import StepLrWarmup, TaskEpochBasedRunner, build_optimizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN_CFG = os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_70_10.py")
REF_CFG = "/root/reference/configs/deformable_detr/chaosuan_gfl_deformable_detr_40_r50_8x4_1x_qoqo_il.py"


def test_registry_names_of_the_reference_resolve():
    for reg, names in [(builder.DETECTORS, ["DeformableDETR_il"]), (builder.HEADS, ["GFLDeformableDETRHead_il"]),
                       (builder.BBOX_ASSIGNERS, ["GFLHungarianAssigner"]), (builder.BBOX_SAMPLERS, ["PseudoSampler"]),
                       (builder.MATCH_COST, ["QualityFocalLossCost", "BBoxL1Cost", "IoUCost"]),
                       (builder.LOSSES, ["QualityFocalLoss", "DistributionFocalLoss", "L1Loss", "GIoULoss",
                                         "KnowledgeDistillationKLDivLoss", "SmoothL1Loss", "MSELoss"]),
                       (builder.BACKBONES, ["ResNet"]), (builder.NECKS, ["ChannelMapper"]),
                       (builder.TRANSFORMER, ["DeformableDetrTransformer"]),
                       (builder.TRANSFORMER_LAYER_SEQUENCE, ["DetrTransformerEncoder", "DeformableDetrTransformerDecoder"]),
                       (builder.TRANSFORMER_LAYER, ["BaseTransformerLayer", "DetrTransformerDecoderLayer"]),
                       (builder.ATTENTION, ["MultiScaleDeformableAttention", "MultiheadAttention"]),
                       (builder.POSITIONAL_ENCODING, ["SinePositionalEncoding"])]:
        for n in names:
            assert reg.get(n) is not None, n
    with pytest.raises(KeyError):
        builder.build_loss(dict(type="NoSuchLoss"))
    with pytest.raises(KeyError):
        builder.DETECTORS.register_module()(builder.DETECTORS.get("DeformableDETR_il"))


def test_own_config_loads_and_builds():
    cfg = Config.fromfile(OWN_CFG)
    assert cfg.model.type == "DeformableDETR_il" and cfg.data.train.catsplit == (70, 10)
    assert isinstance(cfg.optimizer, list) and len(cfg.runner) == 2
    cfg.merge_from_dict({"data.samples_per_gpu": 2, "model.bbox_head.num_query": 50})
    assert cfg.data.samples_per_gpu == 2 and cfg.model.bbox_head.num_query == 50


@pytest.mark.skipif(not os.path.isfile(REF_CFG), reason="reference checkout not present (GPU box)")
def test_reference_config_files_load_unchanged():
    import glob
    files = sorted(glob.glob("/root/reference/configs/deformable_detr/*_il*.py"))
    assert len(files) >= 8
    for f in files:
        cfg = Config.fromfile(f)          # _base_ chain (dataset + runtime) included
        assert "model" in cfg and "data" in cfg, f
    cfg = Config.fromfile(REF_CFG)
    assert cfg.model.bbox_head.feats_distill == "corr + fg_info + decode_v1"
    assert cfg.model.bbox_head.cates_distill == "hard + teacher-first"
    assert cfg.dist_params["backend"] == "nccl" and cfg.checkpoint_config["interval"] == 1
    cfg.model.backbone.init_cfg = None
    model = builder.build_detector(cfg.model)
    assert sum(p.numel() for p in model.parameters()) == 40156376


def _tiny_model(num_query=20):
    cfg = Config.fromfile(OWN_CFG)
    cfg.model.bbox_head.num_query = num_query
    torch.manual_seed(0)
    m = builder.build_detector(cfg.model)
    m.init_weights()
    return cfg, m


def test_head_ctor_checks_loss_and_matcher_weights():
    cfg = Config.fromfile(OWN_CFG)
    bad = copy.deepcopy(cfg.model)
    bad.bbox_head.loss_bbox = dict(type="L1Loss", loss_weight=4.0)
    with pytest.raises(AssertionError, match="regression L1 weight"):
        builder.build_detector(bad)
    bad = copy.deepcopy(cfg.model)
    bad.bbox_head.positional_encoding = dict(type="SinePositionalEncoding", num_feats=64, normalize=True)
    with pytest.raises(AssertionError, match="embed_dims"):
        builder.build_detector(bad)


def test_teacher_is_a_plain_attribute():
    """deformable_detr_il.py:79-114, :467-496: teacher frozen, eval, not registered, not saved."""
    cfg, m = _tiny_model()
    n_params = sum(1 for _ in m.parameters())
    keys = set(m.state_dict().keys())
    t = copy.deepcopy(m)
    assert m.set_teacher(config=None, ckptfile=None, model=None) is None and not m.has_teacher
    m.set_teacher(model=t)
    assert m.has_teacher and m.bbox_head.has_teacher and not t.has_teacher and not t.bbox_head.has_teacher
    assert sum(1 for _ in m.parameters()) == n_params and set(m.state_dict().keys()) == keys
    assert all(not p.requires_grad for p in t.parameters())
    m.train()
    assert m.training and not t.training                       # eval_teacher
    # a second hand-over drops the nested teacher
    m2 = copy.deepcopy(m)
    m.set_teacher(model=m2)
    assert getattr(m.teacher_model, "teacher_model", None) is None
    m.set_datainfo(cat2id={"a": 1, "b": 2, "c": 3}, cat2label={1: 0, 2: 1, 3: 2}, pred_cat=["a", "b"], load_cat=["b"],
                   task_cat=[["a"], ["b"], ["c"]])
    assert m.LableInPCNTask == {"prev": [0], "curr": [1], "next": [2]}
    # unused-in-forward parameter exists (reason for find_unused_parameters=True)
    assert "bbox_head.prototype.weight" in keys
    # frozen stem / stage 1 and every BN
    assert not m.backbone.conv1.weight.requires_grad and not m.backbone.layer1[0].conv1.weight.requires_grad
    assert m.backbone.layer2[0].conv1.weight.requires_grad and not m.backbone.layer2[0].bn1.weight.requires_grad
    assert not m.backbone.layer2[0].bn1.training


def test_optimizer_param_groups_follow_custom_keys():
    cfg, m = _tiny_model()
    opt = build_optimizer(m, cfg.optimizer[0])
    lrs = sorted({g["lr"] for g in opt.param_groups})
    assert lrs == [pytest.approx(2e-5), pytest.approx(2e-4)]
    small = next(g for g in opt.param_groups if g["lr"] < 1e-4)
    ids = {id(p) for p in small["params"]}
    for n, p in m.named_parameters():
        if p.requires_grad:
            expect_small = ("backbone" in n) or ("sampling_offsets" in n) or ("reference_points" in n)
            assert (id(p) in ids) == expect_small, n
    assert all(p.requires_grad for g in opt.param_groups for p in g["params"])


def test_step_lr_with_linear_warmup():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = StepLrWarmup(opt, step=[8, 11], warmup="linear", warmup_iters=10, warmup_ratio=0.01)
    sch.set(0, 0)
    assert opt.param_groups[0]["lr"] == pytest.approx(0.01)
    sch.set(0, 5)
    assert opt.param_groups[0]["lr"] == pytest.approx(1 - 0.5 * 0.99)
    sch.set(0, 10)
    assert opt.param_groups[0]["lr"] == pytest.approx(1.0)
    sch.set(8, 100)
    assert opt.param_groups[0]["lr"] == pytest.approx(0.1)
    sch.set(11, 100)
    assert opt.param_groups[0]["lr"] == pytest.approx(0.01)


def test_split_data_category_matches_reference_goldens():
    """Class table and task split (reference mmdet/datasets/data_split.py:62-80, :100-158) against the
    outputs of the reference's own function (tests/golden/gen_golden.py --datasplit), incl. the
    'shuffle' order under a seeded ``random`` and the string form of ``split``."""
    import json
    import os
    import random

    from dskd_amd.datasets import COCO_CATS_IDS, split_data_category
    with open(os.path.join(os.path.dirname(__file__), "golden", "data_split_cases.json")) as f:
        gold = json.load(f)
    assert [list(kv) for kv in COCO_CATS_IDS.items()] == gold["coco_cats_ids"]
    assert len(gold["cases"]) >= 8
    for c in gold["cases"]:
        if c["seed"] is not None:
            random.seed(c["seed"])
        split = c["split"] if isinstance(c["split"], str) else tuple(c["split"])
        out = split_data_category(split=split, order=c["order"], catofset=c["catofset"], valpart=c["valpart"])
        groups = out if isinstance(out, tuple) else (out,)
        got = [[[list(kv) for kv in d.items()] for d in grp] for grp in groups]
        assert got == c["out"], (c["split"], c["order"], c["valpart"], c["catofset"])
    with pytest.raises(NotImplementedError):
        split_data_category(dataname="VOCDataset", split=(10, 10), valpart="prev-cur")
    with pytest.raises(ValueError):
        split_data_category(split=(40, 40), order="random", valpart="prev-cur")
    with pytest.raises(AssertionError):
        split_data_category(split=(40, 40))             # the default valpart is a menu, not a mode (reference :101, :131)


def test_synthetic_il_dataset_uses_protocol_classes():
    ds = SyntheticILDataset(catsplit=(70, 10), catload=(0, 1), num_images=2, img_size=(32, 48), n_gt=2)
    assert ds.TASK_CLASSES[0][0] == "airplane" and len(ds.TASK_CLASSES[0]) == 70
    assert ds.TASK_CLASSES[1] == ["toilet", "toothbrush", "traffic light", "train", "truck", "tv", "umbrella", "vase",
                                  "wine glass", "zebra"]
    assert ds.ALL_CLASSES_IDS["person"] == 1 and ds.cat2label[ds.ALL_CLASSES_IDS["airplane"]] == 0
    assert ds.cat2label[ds.ALL_CLASSES_IDS["zebra"]] == 79
    assert sorted(ds.cat2label[ds.ALL_CLASSES_IDS[c]] for c in ds.LOAD_CLASSES) == list(range(70, 80))
    model_labels = [ds.cat2label[ds.ALL_CLASSES_IDS[c]] for c in set(ds.PRED_CLASSES) - set(ds.LOAD_CLASSES)]
    assert sorted(model_labels) == list(range(70))          # what set_datainfo turns into LableInPCNTask['prev']


def test_synthetic_il_dataset_surface():
    ds = SyntheticILDataset(catsplit=(40, 40), catload=(0, 1), num_images=5, img_size=(64, 96), n_gt=3)
    assert len(ds.TASK_CLASSES) == 2 and len(ds.LOAD_CLASSES) == 40 and len(ds.PRED_CLASSES) == 80
    assert ds.cat2label[ds.ALL_CLASSES_IDS[ds.LOAD_CLASSES[0]]] == 40
    item = ds[3]
    assert item["img"].shape == (3, 64, 96) and item["gt_bboxes"].shape == (3, 4)
    assert int(item["gt_labels"].min()) >= 40
    assert torch.equal(ds[3]["img"], item["img"])                # deterministic per index
    b = item["gt_bboxes"]
    assert (b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all() and b[:, 2].max() <= 96 and b[:, 3].max() <= 64
    loader = build_dataloader(ds, 2, 0, seed=1)
    batch = next(iter(loader))
    assert batch["img"].shape == (2, 3, 64, 96) and len(batch["img_metas"]) == 2


def test_incremental_training_plumbing_cpu(cpu_ops, tmp_path):
    """BASELINE.json configs[0]: two tasks on CPU through the driver, synthetic tensors; the
    second task distils from the frozen copy of the first; checkpoints hold the student only."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_increment", os.path.join(ROOT, "tools", "train_increment.py"))
    ti = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ti)
    cfg_file = REF_CFG if os.path.isfile(REF_CFG) else OWN_CFG
    runners = ti.main([cfg_file, "--device", "cpu", "--max-iters", "2", "--max-epochs", "1", "--work-dir", str(tmp_path),
                       "--cfg-options", "data.samples_per_gpu=1", "data.workers_per_gpu=0", "data.train.num_images=2",
                       "data.train.img_size=(64,96)", "data.train.n_gt=2", "model.bbox_head.num_query=30"])
    assert len(runners) == 2
    keys1, keys2 = set(runners[0].history[-1]), set(runners[1].history[-1])
    det = {"loss_cls", "loss_bbox", "loss_iou", "loss_dfl"} | {f"d{i}.{k}" for i in range(5)
                                                             for k in ("loss_cls", "loss_bbox", "loss_iou", "loss_dfl")}
    assert det | {"loss"} <= keys1 and "loss_corr" not in keys1
    assert det | {"loss", "loss_corr", "loss_fg_feature"} <= keys2
    for r in runners:
        assert all(torch.isfinite(torch.tensor(h["loss"])) for h in r.history)
    ck = torch.load(os.path.join(str(tmp_path), "task_2_epoch_1.pth"), map_location="cpu")
    assert not any(k.startswith("teacher") for k in ck["state_dict"])
    assert ck["meta"] == dict(task=2, epoch=1, iter=2)
    # resume restores epoch / iter
    model = runners[1].module
    r2 = TaskEpochBasedRunner(model, runners[1].optimizer, max_epochs=1)
    assert r2.resume(os.path.join(str(tmp_path), "task_2_epoch_1.pth")) == dict(task=2, epoch=1, iter=2)


def test_checkpoint_hand_over_in_reference_layout(tmp_path):
    """checkpoints in the reference's (mmcv) layout -- ``{'meta', 'state_dict'}``
    with keys ``backbone.* / neck.* / bbox_head.*`` -- load into the student (``set_student``) and,
    with a config path, build the frozen teacher (``set_teacher(config=, ckptfile=)``,
    deformable_detr_il.py:79-114)."""
    cfg, src = _tiny_model()
    sd = src.state_dict()
    # the names a reference checkpoint carries for this architecture
    for k in ("backbone.conv1.weight", "backbone.bn1.running_var", "backbone.layer3.5.conv3.weight",
              "backbone.layer2.0.downsample.0.weight", "neck.convs.0.conv.weight", "neck.convs.2.gn.bias",
              "neck.extra_convs.0.conv.weight", "bbox_head.query_embedding.weight", "bbox_head.prototype.weight",
              "bbox_head.cls_branches.0.weight", "bbox_head.reg_branches.0.4.bias",
              "bbox_head.transformer.level_embeds", "bbox_head.transformer.reference_points.weight",
              "bbox_head.transformer.encoder.layers.5.attentions.0.sampling_offsets.weight",
              "bbox_head.transformer.encoder.layers.0.attentions.0.attention_weights.bias",
              "bbox_head.transformer.encoder.layers.0.ffns.0.layers.0.0.weight",
              "bbox_head.transformer.encoder.layers.0.ffns.0.layers.1.bias",
              "bbox_head.transformer.encoder.layers.0.norms.1.weight",
              "bbox_head.transformer.decoder.layers.0.attentions.0.attn.in_proj_weight",
              "bbox_head.transformer.decoder.layers.0.attentions.0.attn.out_proj.bias",
              "bbox_head.transformer.decoder.layers.0.attentions.1.value_proj.weight",
              "bbox_head.transformer.decoder.layers.5.norms.2.bias"):
        assert k in sd, k
    ckpt = tmp_path / "task_1_epoch_12.pth"
    # a reference checkpoint repeats the shared heads under every decoder-layer index
    extra = {f"bbox_head.cls_branches.{i}.weight": sd["bbox_head.cls_branches.0.weight"] for i in range(1, 6)}
    torch.save({"meta": {"epoch": 12}, "state_dict": {**sd, **extra}}, ckpt)

    torch.manual_seed(123)
    dst = builder.build_detector(copy.deepcopy(cfg.model))
    dst.init_weights()
    assert not torch.equal(dst.bbox_head.cls_branches[0].weight, src.bbox_head.cls_branches[0].weight)
    dst.set_student(ckptfile=str(ckpt))
    for (n, a), (_, b) in zip(dst.state_dict().items(), src.state_dict().items()):
        assert torch.equal(a, b), n
    dst.set_teacher(config=cfg, ckptfile=str(ckpt))            # a Config object or a config file path
    t = dst.teacher_model
    assert dst.has_teacher and not t.training and all(not p.requires_grad for p in t.parameters())
    assert torch.equal(t.backbone.layer4[2].conv3.weight, src.backbone.layer4[2].conv3.weight)
    assert "teacher_model.backbone.conv1.weight" not in dst.state_dict()
    dst.load_student(str(ckpt))                                   # drops the teacher (:116-121 of the reference flow)
    assert not dst.has_teacher


def test_lean_self_attention_equals_nn_multihead_attention():
    """The mask-free GPU path of the MultiheadAttention wrapper (shared q/k projection + fused SDPA)
    is the same function as ``nn.MultiheadAttention`` -- checked on the CPU by calling it directly."""
    from dskd_amd.transformer import MultiheadAttention
    torch.manual_seed(4)
    m = MultiheadAttention(embed_dims=256, num_heads=8, dropout=0.0).eval()
    q, pos = torch.randn(30, 2, 256), torch.randn(30, 2, 256)
    ref = m.attn(q + pos, q + pos, q, need_weights=False)[0]
    qp = q + pos
    torch.testing.assert_close(m._attend(qp, qp, q), ref, rtol=1e-5, atol=1e-5)            # shared q/k
    k2 = torch.randn(17, 2, 256)
    ref2 = m.attn(qp, k2, k2, need_weights=False)[0]
    torch.testing.assert_close(m._attend(qp, k2, k2), ref2, rtol=1e-5, atol=1e-5)          # cross attention
    # and through forward(): identity + attention (dropout 0)
    out = m(q, query_pos=pos)
    torch.testing.assert_close(out, q + ref, rtol=1e-5, atol=1e-5)


def test_deepcopy_drops_runtime_accelerator_state():
    """The incremental driver deep-copies the trained student into the next teacher
    (train_increment.py:250-251): hipGraph / stream holders cached on the modules must not be
    copied (they cannot be) and must not survive in the copy."""
    import threading
    cfg, m = _tiny_model()
    m.__dict__["_teacher_ahead"] = threading.Lock()                 # stands in for streams + CUDAGraphs
    m.bbox_head.__dict__["_dense_graphs"] = {"sig": threading.Lock()}
    m.bbox_head.__dict__["_dense_seen"] = {"sig": 3}
    c = copy.deepcopy(m)
    assert "_teacher_ahead" not in c.__dict__ and "_dense_graphs" not in c.bbox_head.__dict__
    assert "_teacher_ahead" in m.__dict__ and "_dense_graphs" in m.bbox_head.__dict__
    for (n, a), (_, b) in zip(m.state_dict().items(), c.state_dict().items()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), n
    assert c.bbox_head is not m.bbox_head and c.bbox_head.transformer is not m.bbox_head.transformer


def test_inference_surface_returns_the_reference_format(cpu_ops):
    """``model(return_loss=False, img=[...], img_metas=[[...]])`` (reference ``BaseDetector.forward_test`` base.py:112-154
    -> ``simple_test`` deformable_detr_il.py:365-387): double-nested inputs, ``batch_input_shape`` filled in,
    per image one [n_c, 5] float32 array per class; more than one augmentation is refused."""
    import copy
    import os

    import numpy as np

    from dskd_amd.builder import build_detector
    from dskd_amd.config import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mc = copy.deepcopy(Config.fromfile(os.path.join(root, "configs", "dskd_gfl_deformable_detr_r50_70_10.py")).model)
    for part in ("encoder", "decoder"):
        mc["bbox_head"]["transformer"][part]["num_layers"] = 1
    mc["bbox_head"]["num_query"] = 20
    torch.manual_seed(0)
    model = build_detector(mc)
    model.init_weights()
    model.eval()
    img = torch.randn(2, 3, 64, 96)
    metas = [dict(img_shape=(64, 96, 3), scale_factor=1.0), dict(img_shape=(50, 96, 3), scale_factor=1.0)]
    with torch.no_grad():
        res = model(return_loss=False, img=[img], img_metas=[metas], rescale=False)
    assert metas[0]["batch_input_shape"] == (64, 96)
    assert len(res) == 2 and all(len(r) == 80 for r in res)
    for r in res:
        for a in r:
            assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 5
        assert sum(len(a) for a in r) <= 100                                  # test_cfg max_per_img
    with pytest.raises(TypeError):
        model(return_loss=False, img=img, img_metas=[metas])
    with pytest.raises(NotImplementedError):
        model(return_loss=False, img=[img, img], img_metas=[metas, metas])


def test_gfl_hungarian_assigner_like_the_reference_test(cpu_ops):
    """The reference's own assigner test (tests/test_utils/test_assigner.py:385-428), applied to the assigner of
    this path (``GFLHungarianAssigner``): no ground truth -> everything background / unlabeled; with ground truth
    every gt is matched exactly once; the plain-IoU cost mode; ``gt_bboxes_ignore`` is refused."""
    from dskd_amd import bbox as pbbox
    asg = pbbox.GFLHungarianAssigner(cls_cost=dict(type="QualityFocalLossCost", weight=2.0),
                                     reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
                                     iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))
    assert asg.iou_cost.iou_mode == "giou"
    g = torch.Generator().manual_seed(0)
    bbox_pred, cls_pred = torch.rand((10, 4), generator=g), torch.rand((10, 80), generator=g)
    img_meta = dict(img_shape=(10, 8, 3))
    res = asg.assign(bbox_pred, cls_pred, torch.empty((0, 4)).float(), torch.empty((0,)).long(), None, img_meta)
    assert res.num_gts == 0 and torch.all(res.gt_inds == 0) and torch.all(res.labels == -1)
    gt_bboxes, gt_labels = torch.FloatTensor([[0, 0, 5, 7], [3, 5, 7, 8]]), torch.LongTensor([1, 20])
    for cfg in (dict(type="IoUCost", iou_mode="giou", weight=2.0), dict(type="IoUCost", iou_mode="iou", weight=1.0)):
        asg = pbbox.GFLHungarianAssigner(cls_cost=dict(type="QualityFocalLossCost", weight=2.0),
                                         reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"), iou_cost=cfg)
        res = asg.assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, None, img_meta)
        assert torch.all(res.gt_inds > -1)
        assert (res.gt_inds > 0).sum() == gt_bboxes.size(0) and (res.labels > -1).sum() == gt_bboxes.size(0)
        assert sorted(res.gt_inds[res.gt_inds > 0].tolist()) == [1, 2]
        assert sorted(res.labels[res.labels > -1].tolist()) == [1, 20]
    with pytest.raises(AssertionError):
        asg.assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, None, img_meta, gt_bboxes_ignore=torch.zeros(1, 4))
    # no predictions at all
    res = asg.assign(torch.empty((0, 4)), torch.empty((0, 80)), gt_bboxes, gt_labels, None, img_meta)
    assert res.num_gts == 2 and len(res.gt_inds) == 0


def test_own_40_40_config_describes_the_reference_model():
    """configs/dskd_gfl_deformable_detr_r50_40_40.py (the copy that travels to the GPU box) against the reference's
    chaosuan_gfl_deformable_detr_40_r50_8x4_1x_qoqo_il.py: same model dict (but for the pretrained-checkpoint path),
    same optimizer / clip / lr schedule / runner lists, same class split."""
    if not os.path.isfile(REF_CFG):
        pytest.skip("reference tree not present")
    from dskd_amd.config import Config

    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x
    ref = Config.fromfile(REF_CFG)
    own = Config.fromfile(os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_40_40.py"))
    mr, mo = plain(ref.model), plain(own.model)
    mr["backbone"]["init_cfg"] = mo["backbone"]["init_cfg"] = None
    assert mr == mo
    for k in ("optimizer", "optimizer_config", "lr_config", "runner"):
        assert plain(ref[k]) == plain(own[k]), k
    assert tuple(ref.data.train.catsplit) == tuple(own.data.train.catsplit) == (40, 40)


def test_train_increment_command_line_is_the_reference_drivers(cpu_ops, tmp_path):
    """/root/reference/tools/dist_train_increment.sh:22-28 calls ``train_increment.py --config=... --work-dir=...
    --resume-from=$CHECKPOINT --launcher=pytorch``; the same spelling drives ours (CONFIG may also be positional),
    an empty --resume-from means none, and a checkpoint given there restores student, optimizer and counters."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_increment", os.path.join(ROOT, "tools", "train_increment.py"))
    ti = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ti)
    a = ti.parse_args([f"--config={OWN_CFG}", f"--work-dir={tmp_path}", "--resume-from=", "--launcher=none"])
    assert a.config == OWN_CFG and a.resume_from == "" and not a.auto_resume
    assert ti.parse_args([OWN_CFG]).config == OWN_CFG
    for bad in ([], [OWN_CFG, f"--config={OWN_CFG}"]):
        with pytest.raises(SystemExit):
            ti.parse_args(bad)
    small = ["--device", "cpu", "--max-iters", "1", "--cfg-options", "data.samples_per_gpu=1", "data.workers_per_gpu=0",
             "data.train.num_images=2", "data.train.img_size=(64,96)", "data.train.n_gt=2", "model.bbox_head.num_query=30",
             "data.train.catsplit=(80,)", "data.train.catload=(1,)"]
    r1 = ti.main([f"--config={OWN_CFG}", f"--work-dir={tmp_path}", "--resume-from=", "--max-epochs", "1"] + small)
    ck = os.path.join(str(tmp_path), "task_1_epoch_1.pth")
    assert len(r1) == 1 and os.path.isfile(ck) and ti.find_latest_checkpoint(str(tmp_path)) == ck
    # resume: epoch 1 is done, so a 2-epoch run executes exactly one more epoch and starts from the saved weights
    r2 = ti.main([f"--config={OWN_CFG}", f"--work-dir={tmp_path}", f"--resume-from={ck}", "--max-epochs", "2"] + small)
    assert r2[0].epoch == 2 and r2[0].iter == 2 and [h["epoch"] for h in r2[0].history] == [2]
    # --auto-resume picks the newest checkpoint of the work dir (now epoch 2): one more epoch of three is left
    r3 = ti.main([f"--config={OWN_CFG}", f"--work-dir={tmp_path}", "--auto-resume", "--max-epochs", "3"] + small)
    assert r3[0].epoch == 3 and r3[0].iter == 3 and [h["epoch"] for h in r3[0].history] == [3]
    with pytest.raises(FileNotFoundError):
        ti.main([f"--config={OWN_CFG}", f"--work-dir={tmp_path}", "--resume-from=/nonexistent.pth"] + small)
