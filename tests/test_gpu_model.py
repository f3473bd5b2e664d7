"""Model-level GPU tests (-m gpu): the full distillation step through the HIP kernels against
the same step on the CPU with the oracle injected, and the hipGraph replay against eager."""
import copy
import os

import pytest
import torch

import dskd_amd  # noqa: F401
from dskd_amd import native
from dskd_amd.builder import build_detector
from dskd_amd.config import Config
from dskd_amd.graph_step import GraphedDistillStep
from dskd_amd.runner import build_optimizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_70_10.py")
CFG_40 = os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_40_40.py")      # BASELINE configs[0] / [2] model
CFG_SWIN = os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_swin_t_40_40.py")   # BASELINE configs[3]: Swin-T 40+40
CFG_SWIN_70 = os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_swin_t_70_10.py")  # the same trunk on the headline split
SWIN_CFGS = (CFG_SWIN, CFG_SWIN_70)


def _build(seed=0, num_query=300, cfg_file=CFG):
    cfg = Config.fromfile(cfg_file)
    cfg.model.bbox_head.num_query = num_query
    torch.manual_seed(seed)
    m = build_detector(cfg.model)
    m.init_weights()
    for mod in m.modules():                       # deterministic: no dropout
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    t = copy.deepcopy(m)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in t.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 1e-3)
    m.set_teacher(model=t)
    m.LableInPCNTask = {"prev": list(range(cfg.num_prev)), "curr": list(range(cfg.num_prev, 80)), "next": []}
    return cfg, m


def _batch(dev, B=2, H=192, W=256):
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, H, W, generator=g).to(dev)
    metas = [dict(img_shape=(H, W, 3), batch_input_shape=(H, W), scale_factor=1.0) for _ in range(B)]
    gt_b = [torch.tensor([[10., 12., 90., 100.], [30., 20., 200., 150.]]).to(dev), torch.tensor([[5., 5., 120., 90.]]).to(dev)]
    gt_l = [torch.tensor([75, 71]).to(dev), torch.tensor([79]).to(dev)]
    inject = dict(pred_bboxes=[torch.tensor([[20., 20., 120., 110.], [0., 0., 60., 70.]]).to(dev),
                               torch.tensor([[40., 40., 200., 160.]]).to(dev)],
                  pred_labels=[torch.tensor([1, 7]).to(dev), torch.tensor([3]).to(dev)],
                  pred_keepid=torch.tensor([3, 17, 305]).to(dev))
    return dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l), inject


@pytest.mark.parametrize("cfg_file", [CFG, CFG_40, CFG_SWIN, CFG_SWIN_70],
                         ids=["r50_70_10", "r50_40_40", "swin_t_40_40", "swin_t_70_10"])
def test_full_step_gpu_matches_cpu_oracle_fp32(oracle_checker, cfg_file):
    """Same weights, same batch, fp32: loss dict on the GPU (HIP kernels, device LSAP) vs on the
    CPU (oracle kernels, oracle LSAP).  Also the gradient of a few parameters.  The three detector
    configurations BASELINE.json names: R50 70+10 (configs[1]), R50 40+40 (the model of configs[0] / [2]; the
    config is the reference's chaosuan_..._40_..._il.py, see tests/test_host_logic.py) and Swin-T (configs[3])."""
    cfg, m_cpu = _build(cfg_file=cfg_file)
    m_gpu = copy.deepcopy(m_cpu)
    m_gpu.to("cuda:0").train()
    m_cpu.train()
    data_c, inj_c = _batch(torch.device("cpu"))
    data_g, inj_g = _batch(torch.device("cuda:0"))

    def run(model, data, inj):
        feats, outs, *_ = model.out_teacher(data["img"], data["img_metas"])
        ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=inj["pred_keepid"], pred_logits=None, pred_scores=None,
                  pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
        out = model.train_step(dict(data, teacher_info=ti))
        out["loss"].backward()
        return out["log_vars"]

    native.install_cpu_checker(oracle_checker)
    try:
        lv_c = run(m_cpu, data_c, inj_c)
    finally:
        native.install_cpu_checker(None)
    lv_g = run(m_gpu, data_g, inj_g)
    assert set(lv_c) == set(lv_g)
    # An untrained detector sits on assignment near-ties: the three teacher boxes of this batch are matched to three
    # of 300 almost identical queries, and loss_corr is the distance matrix of exactly those three embeddings.  With
    # the Swin trunk the GPU / CPU fp32 difference (fused attention) is enough to move one of them, so for that
    # configuration loss_corr is compared in the two-stage form below (same head inputs on both devices) instead, and the
    # detection losses (a few of 600 assignments move) end to end at 5 %.
    loose = {"loss_corr"} if cfg_file in SWIN_CFGS else set()
    bad = []
    for k in lv_c:
        rtol = 5e-2 if k == "loss_fg_feature" else 2e-3       # fp32 reference noise of decode_v1, see kernel tests
        atol = 1e-5
        if cfg_file in SWIN_CFGS:                              # end to end only a coarse band; the strict form follows
            rtol, atol = 0.15, 5e-4
        if k not in loose and lv_g[k] != pytest.approx(lv_c[k], rel=rtol, abs=atol):
            bad.append((k, lv_g[k], lv_c[k]))
    assert not bad, bad
    # two-stage form: trunk outputs GPU vs CPU, then the head's loss() on the GPU against loss() on the CPU
    # (oracle ops) fed with the SAME head inputs (the GPU's, copied)
    dev, cpu = torch.device("cuda:0"), torch.device("cpu")
    with torch.no_grad():
        xg = m_gpu.extract_feat(data_g["img"])
        og = m_gpu.bbox_head.forward(xg, data_g["img_metas"])
        feats_t, outs_t, *_ = m_gpu.out_teacher(data_g["img"], data_g["img_metas"])

    def to(x, d):
        if torch.is_tensor(x):
            return x.to(d)
        if isinstance(x, (list, tuple)):
            return type(x)(to(y, d) for y in x)
        if isinstance(x, dict):
            return {k: to(v, d) for k, v in x.items()}
        return x
    tl = m_gpu.LableInPCNTask
    ti_g = dict(neck_feats=feats_t, head_outs=outs_t, pred_keepid=inj_g["pred_keepid"], pred_logits=None, pred_scores=None,
                pred_labels=inj_g["pred_labels"], pred_bboxes=inj_g["pred_bboxes"])
    lg = m_gpu.bbox_head.loss(*og, data_g["gt_bboxes"], data_g["gt_labels"], data_g["img_metas"], student_feat=xg,
                              teacher_info=ti_g, task_labels=tl)
    native.install_cpu_checker(oracle_checker)
    try:
        lc = m_cpu.bbox_head.loss(*to(og, cpu), data_c["gt_bboxes"], data_c["gt_labels"], data_c["img_metas"],
                                  student_feat=to(xg, cpu), teacher_info=to(ti_g, cpu), task_labels=tl)
    finally:
        native.install_cpu_checker(None)
    assert set(lg) == set(lc)
    for k in lc:
        # decode_v1 with a teacher 1e-3 away from the student is ~1e-5: the fp32 oracle's cancellation noise is of that
        # size (the kernel is pinned against the fp64 oracle in tests/test_gpu_kernels.py::test_fgkd_vs_oracle)
        tol = dict(rtol=5e-2, atol=2e-5) if k == "loss_fg_feature" else dict(rtol=1e-3, atol=1e-6)
        torch.testing.assert_close(lg[k].detach().cpu(), lc[k].detach(), msg=lambda m: f"{k}: {m}", **tol)
    names = ["bbox_head.cls_branches.0.weight", "bbox_head.transformer.decoder.layers.5.ffns.0.layers.1.weight",
             "bbox_head.transformer.encoder.layers.0.attentions.0.value_proj.weight", "neck.convs.0.conv.weight"]
    names.append("backbone.stages.3.blocks.1.attn.w_msa.qkv.weight" if cfg_file in SWIN_CFGS else "backbone.layer4.2.conv3.weight")
    if cfg_file in SWIN_CFGS:          # a few of the 600 assignments differ between the devices (above): the end-to-end
        names = []                    # gradients are those of two slightly different matchings
    for name in names:
        gc = dict(m_cpu.named_parameters())[name].grad
        gg = dict(m_gpu.named_parameters())[name].grad.cpu()
        rel = (gc - gg).norm() / (gc.norm() + 1e-12)
        assert rel < 2e-2, (name, float(rel))
    native.raise_for_lsap_status(m_gpu.bbox_head.last_lsap_status)


@pytest.mark.parametrize("channels_last,amp", [(False, None), (True, torch.bfloat16)])
def test_graph_replay_equals_eager(channels_last, amp):
    """hipGraph capture/replay of the step produces the same training trajectory as eager
    (also in the benchmark's configuration: channels_last model, bf16 autocast)."""
    cfg, m1 = _build(seed=3)
    m2 = copy.deepcopy(m1)
    dev = torch.device("cuda:0")
    losses = []
    for m, use_graphs in ((m1, False), (m2, True)):
        m.to(dev).train()
        if channels_last:
            m.to(memory_format=torch.channels_last)
            m.teacher_model.to(memory_format=torch.channels_last)
        opt = build_optimizer(m, cfg.optimizer[0], capturable=True)
        data, inject = _batch(dev)
        stepper = GraphedDistillStep(m, opt, amp_dtype=amp, max_norm=0.1, use_graphs=use_graphs, warmup=2)
        seq = []
        for _ in range(6):
            loss = stepper.step(data, inject)
            seq.append(float(loss))
        losses.append(seq)
        if use_graphs:
            assert len(stepper._graphs) == 1
            logs = stepper.logs()
            assert "loss_corr" in logs and "loss_fg_feature" in logs and logs["loss"] == pytest.approx(seq[-1], rel=1e-5)
    for a, b in zip(*losses):
        assert b == pytest.approx(a, rel=2e-3 if amp is None else 3e-2), losses
    assert losses[0][-1] != losses[0][0]                      # the weights actually move


def test_teacher_ahead_matches_inline_teacher():
    """The teacher run one batch ahead on a second stream hands over exactly what the inline
    ``out_teacher`` computes (frozen teacher: same kernels, same inputs), and a student step fed
    from it gives the same losses."""
    cfg, m = _build()
    m.to("cuda:0").train()
    data, inj = _batch(torch.device("cuda:0"))
    feats, outs, keepid, logits, labels, scores, bboxes = m.out_teacher(data["img"], data["img_metas"])
    ahead = m.teacher_ahead()
    ahead.launch(data["img"], data["img_metas"])
    # main-stream work queued between launch and finish, as in a training step
    junk = torch.randn(2048, 2048, device="cuda:0")
    for _ in range(8):
        junk = junk @ junk.t() * 1e-3
    ti = ahead.finish()
    assert ahead.pending is None
    # same kernels on the same inputs; MIOpen may pick another algorithm for the second call, so
    # "same" means to rounding, and the decode is compared through its sizes
    for a, b in zip(ti["neck_feats"], feats):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    for a, b in zip(ti["head_outs"][:2], outs[:2]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    assert ti["pred_keepid"].shape == keepid.shape
    assert [t.shape for t in ti["pred_bboxes"]] == [t.shape for t in bboxes]
    # nothing pending: finish() falls back to the inline teacher
    ti2 = ahead.finish(data["img"], data["img_metas"])
    assert ti2["pred_keepid"].shape == keepid.shape

    def losses(teacher_info):
        info = dict(teacher_info, pred_keepid=inj["pred_keepid"], pred_labels=inj["pred_labels"],
                    pred_bboxes=inj["pred_bboxes"], pred_logits=None, pred_scores=None)
        return m.train_step(dict(data, teacher_info=info))["log_vars"]
    inline = dict(neck_feats=feats, head_outs=outs)
    la, lb = losses(ti), losses(inline)
    assert set(la) == set(lb)
    for k in la:
        assert la[k] == pytest.approx(lb[k], rel=1e-3, abs=1e-4), k


def test_bf16_tall_step_tracks_fp32_step():
    """The benchmark's execution mode (bf16 autocast, channels_last) at a size where the tall-token
    paths are active (3 x 7 140 tokens >= 16 384: split-K weight gradients, fused FFN activation,
    column-sum bias gradients, bf16 residual streams), against the same step in fp32 on the GPU:
    every loss within bf16 tolerance, every trainable parameter with a finite gradient, and the
    gradients pointing the same way."""
    cfg, m32 = _build(seed=7, num_query=100)
    m16 = copy.deepcopy(m32)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, H, W = 3, 512, 672
    img = torch.randn(B, 3, H, W, generator=g).to(dev)
    metas = [dict(img_shape=(H, W, 3), batch_input_shape=(H, W), scale_factor=1.0) for _ in range(B)]
    gt_b = [torch.tensor([[30., 40., 300., 280.], [200., 100., 600., 400.]]).to(dev) for _ in range(B)]
    gt_l = [torch.tensor([75, 71]).to(dev) for _ in range(B)]
    inj = dict(pred_bboxes=[torch.tensor([[50., 60., 320., 300.]]).to(dev) for _ in range(B)],
               pred_labels=[torch.tensor([5]).to(dev) for _ in range(B)],
               pred_keepid=torch.tensor([3, 117, 205]).to(dev))

    def run(model, amp):
        model.to(dev).train()
        x = img
        if amp is not None:
            model.to(memory_format=torch.channels_last)
            model.teacher_model.to(memory_format=torch.channels_last)
            x = img.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            feats, outs, *_ = model.out_teacher(x, metas)
            ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=inj["pred_keepid"], pred_logits=None,
                      pred_scores=None, pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
            out = model.train_step(dict(img=x, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=ti))
        out["loss"].backward()
        return out["log_vars"]

    lv32, lv16 = run(m32, None), run(m16, torch.bfloat16)
    assert set(lv32) == set(lv16)
    print({k: (round(lv16[k], 4), round(lv32[k], 4)) for k in lv32})
    # an untrained model sits on assignment ties: a bf16-sized change of the matching cost can move
    # a query to another target, which moves the classification terms by several per cent
    assert lv16["loss"] == pytest.approx(lv32["loss"], rel=3e-2)
    for k in lv32:
        assert lv16[k] == pytest.approx(lv32[k], rel=0.2 if "cls" in k else 6e-2, abs=5e-3), (k, lv16[k], lv32[k])
    p32 = dict(m32.named_parameters())
    cosines = []
    for name, p in m16.named_parameters():
        if not p.requires_grad or name.startswith("teacher_model"):
            continue
        if p32[name].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        a, b = p.grad.flatten().double(), p32[name].grad.flatten().double()
        if float(b.norm()) > 1e-6 and a.numel() >= 256:
            cosines.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), name))
    cosines.sort()
    assert len(cosines) > 100
    assert cosines[0][0] > 0.5, cosines[:5]                       # nothing points the wrong way
    assert cosines[len(cosines) // 10][0] > 0.9, cosines[:30]     # 90 % of the tensors agree to > 0.9
    assert cosines[len(cosines) // 2][0] > 0.97


def _rel(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_benched_mode_step_at_800x1333_vs_fp32():
    """BASELINE configs[1] exactly as bench.py runs it -- 800 x 1333 synthetic COCO-shaped batch (bench.make_batch: 7 ground
    truth + 10 injected teacher detections per image), 70 + 10 classes, bf16 autocast, channels_last, the student head as
    two hipGraph replays and the teacher's forward as a hipGraph replay on its side stream (TeacherAhead) -- against the
    same step in fp32 (eager, no graphs) on the GPU from the same weights.  B = 2, dropout off.  This is the only test in
    which the bf16-only kernels (gemm_nt / gemm_big / gemm_tn, the fused FFN, lin256, the matrix-core grad_value, the fused
    Bottleneck node, the pre-packed weight images) and the size-gated tall paths run TOGETHER at the benchmark's size.
      (i)   trunk outputs -- neck features, encoder memory, every decoder layer's query embedding, logits and boxes --
            within a relative Frobenius bound per tensor (bf16 residual streams: ~0.4 % per rounding, six layers deep);
      (ii)  ``loss()`` on IDENTICAL head inputs (the bf16 run's outputs handed to both heads): every term to rtol 1e-3;
      (iii) every trainable parameter's gradient against fp32 in the relative Frobenius norm, exceptions named."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda:0")
    cfg, m32 = _build(seed=3)
    m16 = copy.deepcopy(m32)
    B = 2
    data, synth = bench.make_batch(B, cfg.num_prev, 111, dev)
    metas = data["img_metas"]

    def info(feats, outs):
        return dict(neck_feats=feats, head_outs=outs, pred_keepid=synth["keep"], pred_logits=None, pred_scores=None,
                    pred_labels=synth["t_l"], pred_bboxes=synth["t_b"])

    # ---- fp32 reference: eager, NCHW, no graphs
    m32.to(dev).train()
    m32.bbox_head.graph_head = False
    with torch.no_grad():
        tf32, to32, *_ = m32.out_teacher(data["img"], metas)
    feats32 = m32.extract_feat(data["img"])
    outs32 = m32.bbox_head(feats32, metas)
    out = m32.train_step(dict(data, teacher_info=info(tf32, to32)))
    out["loss"].backward()
    lv32 = out["log_vars"]

    # ---- the benched mode
    m16.to(dev).train()
    m16.to(memory_format=torch.channels_last)
    m16.teacher_model.to(memory_format=torch.channels_last)
    img16 = data["img"].contiguous(memory_format=torch.channels_last)
    m16.bbox_head.graph_head = True
    ahead = m16.teacher_ahead()
    ahead.use_graphs = True
    lv16 = outs16 = feats16 = ti16 = None
    for step in range(5):                     # head graphs: two eager calls, capture, replays; teacher graph likewise
        for p in m16.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ti = ahead.finish(img16, metas)
            ti16 = info(ti["neck_feats"], ti["head_outs"])
            out = m16.train_step(dict(data, img=img16, teacher_info=ti16))
            ahead.launch(img16, metas, amp_dtype=torch.bfloat16)
        out["loss"].backward()
        lv16 = out["log_vars"]
    # the trunk outputs of the same mode, in an autocast region of their own (a no-grad forward inside the training step's
    # region would leave detached weight casts in autocast's cache and the step after it without those gradients)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        feats16 = m16.extract_feat(img16)
        outs16 = m16.bbox_head(feats16, metas)
    torch.cuda.synchronize()
    hg = m16.bbox_head.__dict__.get("_head_graphs", {})
    assert len(hg) == 1 and all(v not in (None, False) for v in hg.values()), hg        # the head really was replayed
    assert any(ahead._graphs.values()), ahead._graphs                                      # ... and the teacher

    # ---- (i) trunk outputs
    rep = {}
    for i, (a, b) in enumerate(zip(feats16, feats32)):
        rep[f"neck{i}"] = _rel(a.float(), b)
    for i, (a, b) in enumerate(zip(ti16["neck_feats"], tf32)):
        rep[f"teacher_neck{i}"] = _rel(a.float(), b)
    cls16, box16, info16, hs16 = outs16
    cls32, box32, info32_, hs32 = outs32
    rep["memory"] = _rel(info16[0].float(), info32_[0])
    for l in range(hs32.shape[0]):
        rep[f"hs{l}"] = _rel(hs16[l].float(), hs32[l])
    rep["cls"] = _rel(cls16.float(), cls32)
    rep["box"] = _rel(box16.float(), box32)
    print("trunk outputs, relative Frobenius distance bf16 (benched mode) vs fp32:", {k: round(v, 4) for k, v in rep.items()})
    # measured 0.13 % (boxes) .. 0.96 % (teacher's coarse neck levels): bf16 activations and residual streams
    for k, v in rep.items():
        assert v <= 2e-2, (k, v)

    # ---- (ii) loss() on identical head inputs
    def loss_on(model, amp):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            ls = model.bbox_head.loss(cls16.float(), box16.float(), (info16[0].float(), info16[1]), hs16.float(),
                                      data["gt_bboxes"], data["gt_labels"], metas,
                                      student_feat=[f.float() for f in feats16],
                                      teacher_info=dict(ti16, neck_feats=[f.float() for f in ti16["neck_feats"]]),
                                      task_labels=m32.LableInPCNTask)
        return {k: float(v) for k, v in ls.items() if "loss" in k}
    la, lb = loss_on(m16, True), loss_on(m32, False)
    worst_loss = max(abs(la[k] - lb[k]) / max(abs(lb[k]), 1e-6) for k in lb)
    print("loss() on identical head inputs, worst relative difference:", worst_loss)
    assert set(la) == set(lb) and worst_loss <= 1e-3, (worst_loss, la, lb)       # measured: bit-equal (the loss stage is f32)

    # ---- (iii) gradients
    p32 = dict(m32.named_parameters())
    rows = []
    for name, p in m16.named_parameters():
        if not p.requires_grad or name.startswith("teacher_model") or p32[name].grad is None:
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        if float(p32[name].grad.norm()) > 1e-9:
            rows.append((_rel(p.grad, p32[name].grad), name, p.numel()))
    rows.sort(reverse=True)
    print("gradients, relative Frobenius distance: worst 25:", [(round(r, 3), n) for r, n, _ in rows[:25]])
    print("median", rows[len(rows) // 2][0], "p90", rows[len(rows) // 10][0], "n", len(rows))
    print("losses bf16 / fp32:", {k: (round(lv16[k], 4), round(lv32[k], 4)) for k in lv32})
    assert set(lv16) == set(lv32)
    for k in lv32:       # measured: <= 4e-3 on the classification terms (a logit moved by bf16), <= 3e-4 elsewhere
        assert lv16[k] == pytest.approx(lv32[k], rel=1.5e-2 if "cls" in k else 3e-3, abs=1e-4), (k, lv16[k], lv32[k])
    # Gradients.  Measured: median 5 %, 90th percentile 18 %; the tail is ONE family -- parameters whose gradient arrives
    # through the sampling LOCATIONS of the deformable attention (sampling_offsets / attention_weights, the decoder's
    # reference_points and the query embedding behind them): d(out)/d(loc) is a difference of neighbouring value pixels,
    # and a difference of bf16-rounded values carries several per cent of noise that the sums over 22 223 x 128 samples do
    # not average out completely (38 % worst).  Every other parameter -- all of the ResNet, the neck, every projection, FFN
    # and norm of the transformer, the branches -- agrees to 30 %, half of them to 5 %.
    loc_path = ("sampling_offsets", "attention_weights", "reference_points", "query_embedding")
    assert len(rows) >= 240
    for r, name, _ in rows:
        assert r <= (0.6 if any(t in name for t in loc_path) else 0.3), (name, r)
    assert rows[len(rows) // 2][0] <= 0.1 and rows[len(rows) // 10][0] <= 0.3, (rows[len(rows) // 2], rows[len(rows) // 10])


def test_teacher_ahead_graph_replay_tracks_eager_teacher():
    """After the batch signature has repeated, the ahead-of-time teacher forward is a hipGraph
    replay (two alternating graphs): with a NEW image every step its outputs must keep tracking
    the inline teacher, and the previous batch's tensors must stay intact (double buffer)."""
    cfg, m = _build(seed=5)
    m.to("cuda:0").train()
    data, _ = _batch(torch.device("cuda:0"))
    ahead = m.teacher_ahead()
    ahead.use_graphs, ahead.graph_warmup = True, 2
    g = torch.Generator().manual_seed(21)
    prev = None
    for step in range(7):
        img = torch.randn(data["img"].shape, generator=g).to("cuda:0")
        ahead.launch(img, data["img_metas"])
        ti = ahead.finish()
        feats, outs, *_ = m.out_teacher(img, data["img_metas"])
        for a, b in zip(ti["neck_feats"], feats):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(ti["head_outs"][0], outs[0], rtol=1e-3, atol=1e-3)
        if prev is not None:                                  # batch i-1's tensors were not overwritten
            torch.testing.assert_close(prev[0], prev[1], rtol=0, atol=0)
        prev = (ti["neck_feats"][0], ti["neck_feats"][0].clone())
    entries = [v for v in ahead._graphs.values() if v]
    assert len(entries) == 1 and len(entries[0]) == 2        # captured, verified, double-buffered


def test_graphed_dense_losses_equal_eager():
    """The dense detection losses in their PyTorch formulation (what runs where csrc/denseloss.hip does not apply: other
    loss types / beta) replayed as hipGraphs (forward + backward) against the eager computation: new inputs every call,
    non-unit upstream gradients."""
    cfg, m = _build(seed=9)
    head = m.bbox_head.to("cuda:0")
    nl, N, C = 6, 2 * 300, 80
    g = torch.Generator().manual_seed(12)

    def inputs():
        cls = torch.randn(nl, N, C, generator=g).to("cuda:0").requires_grad_(True)
        cx = torch.rand(nl, N, 4, generator=g).mul(0.5).add(0.2).to("cuda:0").requires_grad_(True)
        lr = torch.rand(nl, N, 4 * 17, generator=g).to("cuda:0").requires_grad_(True)
        labels = torch.randint(0, 81, (nl, N), generator=g).to("cuda:0")
        tgt = torch.rand(nl, N, 4, generator=g).mul(0.5).add(0.2).to("cuda:0")
        factors = torch.tensor([[256., 192., 256., 192.]]).repeat(N, 1).to("cuda:0")
        return cls, cx, lr, labels, tgt, labels < 80, factors

    head.graph_dense_losses = True
    head.fused_dense_losses = False           # the PyTorch formulation (the fused kernels of csrc/denseloss.hip have their own test)
    for step in range(6):
        a = inputs()
        w = [torch.rand(nl, generator=g).to("cuda:0") + 0.5 for _ in range(4)]
        avg = float(3 + step)
        out = head._dense_losses(*a, avg)
        sum((o * wi).sum() for o, wi in zip(out, w)).backward()
        got = [t.detach().clone() for t in out] + [a[0].grad.clone(), a[1].grad.clone(), a[2].grad.clone()]
        b = [t.detach().clone().requires_grad_(t.requires_grad) if t.is_floating_point() else t for t in a]
        ref = head.loss_layers_dense(*b, avg)
        sum((o * wi).sum() for o, wi in zip(ref, w)).backward()
        exp = [t.detach() for t in ref] + [b[0].grad, b[1].grad, b[2].grad]
        for x, y in zip(got, exp):
            torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    graphs = head.__dict__["_dense_graphs"]
    assert len(graphs) == 1 and all(v is not False for v in graphs.values())       # captured after two eager calls


@pytest.mark.parametrize("name", ["loss_b1_l40.npz", "loss_b2_l70.npz", "loss_ragged_no_teacher_boxes.npz",
                                  "loss_ragged_no_gt.npz", "loss_ragged_empty.npz"])
def test_head_loss_on_gpu_vs_reference_goldens(name):
    """The head's ``loss`` on the GPU (fused cost + batched device LSAP, HIP DSKD losses) against the outputs of the
    reference's own ``loss`` -- including RAGGED batches whose second image has no teacher detection / no ground
    truth / neither, i.e. empty matching problems inside the batched launches."""
    from test_golden_reference import _load_loss_case, _make_head, t
    dev = torch.device("cuda:0")
    d = _load_loss_case(name)
    z = d["z"]
    head = _make_head(d["L"])
    cls = d["cls"].to(dev).requires_grad_(True)
    box = d["box"].to(dev).requires_grad_(True)
    hs = d["hs"].to(dev).requires_grad_(True)
    fs = [f.to(dev) for f in d["feats_s"]]
    metas = [dict(img_shape=(d["img_hw"][b][0], d["img_hw"][b][1], 3)) for b in range(d["B"])]
    tinfo = dict(neck_feats=[f.to(dev) for f in d["feats_t"]], head_outs=(None, None, None, d["hs_t"][None].to(dev)),
                 pred_keepid=d["keep"].to(dev), pred_labels=[x.to(dev) for x in d["t_l"]],
                 pred_bboxes=[x.to(dev) for x in d["t_b"]])
    losses = head.loss(cls, box, (None, torch.tensor(d["shapes"])), hs, [x.to(dev) for x in d["gt_b"]],
                       [x.to(dev) for x in d["gt_l"]], metas, student_feat=fs, teacher_info=tinfo,
                       task_labels={"prev": list(range(d["L"])), "curr": [], "next": []})
    ref_keys = [k[5:] for k in z.files if k.startswith("loss/")]
    assert sorted(losses.keys()) == sorted(ref_keys)
    for k in ref_keys:
        rtol = 5e-2 if k == "loss_fg_feature" else 2e-4       # decode_v1: fp32 noise of the reference itself (kernel tests)
        torch.testing.assert_close(losses[k].detach().cpu(), t(z[f"loss/{k}"]), rtol=rtol, atol=1e-6, msg=lambda m: f"{k}: {m}")
    sum(v for k, v in losses.items() if "loss" in k).backward()
    torch.testing.assert_close(box.grad.cpu(), t(z["grad/box"]), rtol=2e-3, atol=1e-5)
    torch.testing.assert_close(hs.grad.cpu(), t(z["grad/hs"]), rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(cls.grad.abs().sum(-1).cpu(), t(z["grad/cls_sum_abs"]), rtol=2e-3, atol=1e-5)
    if head.last_lsap_status is not None:
        native.raise_for_lsap_status(head.last_lsap_status)


@pytest.mark.parametrize("tag,feats_distill,memory_distill,key", [
    ("decode_v2", "corr + fg_info + decode_v2", "", "loss_fg_feature"),
    ("kldv", "corr + kldv", "", "loss_fd"),
    ("memory", "corr", "memory", "loss_memory"),
    ("sg_out", "corr + fg_info + sg_out", "", "loss_fg_feature"),
    ("fg_only", "corr + fg_info + fg_only", "", "loss_fg_feature")])
def test_other_distill_variants_on_gpu_vs_reference_goldens(tag, feats_distill, memory_distill, key):
    """SURVEY.md 8f row 4 on the GPU: the other feature / memory distillation branches of the reference's ``loss()``
    (gfl_deformable_detr_head_il.py:646-661, :721-772, :860-925, :1082-1129) with every tensor on cuda:0 (HIP cost /
    LSAP / loss_corr kernels underneath), against the goldens produced by the reference itself."""
    from test_golden_reference import _distill_variant_case
    # The goldens are the reference's fp32 CPU evaluation of KL terms between almost identical distributions, ~0.5 % off
    # the exact value (float64: 5.897e-4 against 5.926e-4 for decode_v2); the GPU evaluates them in float64
    # (dskd_amd/losses.py), so the comparison allows that noise of the reference.
    _distill_variant_case(tag, feats_distill, memory_distill, key, torch.device("cuda:0"), rtol=2e-2, grad_rtol=2e-2)


@pytest.mark.parametrize("tag,cates_distill,locat_distill,keys", [
    ("soft", "hard + soft + teacher-first", "", ("loss_kd",)),
    ("ld", "hard + teacher-first", "bbox + logit", ("loss_ld_bbox", "loss_ld_logit"))])
def test_logit_and_localisation_distillation_on_gpu_vs_reference_goldens(tag, cates_distill, locat_distill, keys):
    """The remaining live branches of the reference's ``loss()`` on the GPU (VERDICT r2 missing #3): 'soft'
    classification distillation (gfl_deformable_detr_head_il.py:590-622) and 'bbox' / 'logit' localisation distillation
    (:624-645) with every tensor on cuda:0 (HIP cost / LSAP / loss_corr kernels underneath), values and gradients
    against the reference's own outputs (tests/golden/loss_variants_b2_l70.npz)."""
    from test_golden_reference import _logit_ld_case
    _logit_ld_case(tag, cates_distill, locat_distill, keys, torch.device("cuda:0"), rtol=1e-3, grad_rtol=5e-3)


@pytest.mark.parametrize("tag", ["many", "few", "none", "rescale", "cfg"])
def test_teacher_decode_values_on_gpu_vs_reference_goldens(tag):
    """Row A6 on the GPU: ``get_bboxes`` -> ``_get_bboxes_single`` -> ``filter_scores_and_topk`` on cuda:0 against
    the outputs of the reference's own methods (gfl_deformable_detr_head_il.py:1535-1668, core/utils/misc.py:119-165):
    kept (query, class) pairs, their order, boxes and logits -- VALUES, not shapes."""
    from test_golden_reference import _teacher_decode_case
    _teacher_decode_case(tag, torch.device("cuda:0"))


@pytest.mark.parametrize("fused_clip", [True, False], ids=["dskd_clip_adamw", "torch_fused"])
def test_adamw_clip_update_gpu_matches_cpu(fused_clip):
    """Row A13: three optimizer updates (global-norm clip max_norm=0.1, AdamW with the config's parameter groups /
    lr multipliers, warm-up lr) on the GPU -- the two-launch ``dskd_clip_adamw`` (dskd_amd.optim.FusedClipAdamW, the
    default) and PyTorch's fused multi-tensor AdamW behind ``clip_grad_norm_`` -- against the same updates on the CPU
    (``torch.optim.AdamW``, ``clip_grad_norm_``) from identical gradients: every parameter agrees to rounding, and so does
    the clip's norm; channels_last convolution weights included."""
    from dskd_amd.optim import FusedClipAdamW
    from dskd_amd.runner import StepLrWarmup
    cfg, m_cpu = _build(seed=4, num_query=50)
    m_gpu = copy.deepcopy(m_cpu).to("cuda:0").to(memory_format=torch.channels_last)
    opts, lrs = [], []
    for m in (m_cpu, m_gpu):
        o = build_optimizer(m, cfg.optimizer[0], fused_clip=fused_clip)
        opts.append(o)
        lrs.append(StepLrWarmup(o, **{k: v for k, v in dict(cfg.lr_config[0]).items() if k != "policy"}))
    assert isinstance(opts[1], FusedClipAdamW) == fused_clip and not isinstance(opts[0], FusedClipAdamW)
    assert fused_clip or opts[1].defaults.get("fused")
    g = torch.Generator().manual_seed(99)
    pc, pg = dict(m_cpu.named_parameters()), dict(m_gpu.named_parameters())
    train = [n for n, p in pc.items() if p.requires_grad]
    assert len(train) > 150
    for it in range(3):
        for n in train:
            gr = torch.randn(pc[n].shape, generator=g) * (10.0 ** (it - 1))      # clip active, very different scales
            pc[n].grad = gr.clone()
            pg[n].grad = torch.empty_like(pg[n]).copy_(gr)                        # the parameter's layout, as autograd makes it
        keep = pg[train[0]].grad.clone()
        norms = []
        for lr, o, params in ((lrs[0], opts[0], pc), (lrs[1], opts[1], pg)):
            lr.set(0, it)
            if hasattr(o, "clip_and_step"):
                norms.append(float(o.clip_and_step(0.1)))
                assert float(o.last_norm[1]) == pytest.approx(min(1.0, 0.1 / (norms[-1] + 1e-6)), rel=1e-5)
            else:
                norms.append(float(torch.nn.utils.clip_grad_norm_([params[n] for n in train], max_norm=0.1, norm_type=2)))
                o.step()
        assert norms[1] == pytest.approx(norms[0], rel=1e-4)          # fp32 sum of 40 M squares, two summation orders
        assert [gr["lr"] for gr in opts[0].param_groups] == [gr["lr"] for gr in opts[1].param_groups]
        if fused_clip:
            assert torch.equal(pg[train[0]].grad, keep)               # the gradients are not rewritten
    for n in train:
        torch.testing.assert_close(pg[n].detach().cpu(), pc[n].detach(), rtol=2e-5, atol=2e-7, msg=lambda m: f"{n}: {m}")
    # the parameters did move (lr_mult 0.1 groups included)
    assert float((pc["backbone.layer4.2.conv3.weight"] - dict(_build(seed=4, num_query=50)[1].named_parameters())
                  ["backbone.layer4.2.conv3.weight"]).abs().max()) > 0
    if fused_clip:      # state layout of torch.optim.AdamW: a checkpoint written by one loads into the other
        sd = opts[1].state_dict()
        st0 = sd["state"][0]
        assert set(st0) == {"step", "exp_avg", "exp_avg_sq"} and int(st0["step"]) == 3
        ref = torch.optim.AdamW([{"params": list(gr["params"])} for gr in opts[1].param_groups], lr=1e-4)
        ref.load_state_dict(sd)
        opts[1].load_state_dict(ref.state_dict())
        for n in train[:3]:
            pg[n].grad = torch.zeros_like(pg[n])
        kept = opts[1].clip_and_step(0.1)
        assert int(opts[1].state[pg[train[0]]]["step"]) == 4
        # a parameter whose step count differs is refused BEFORE any state moves, and the returned norm is
        # the caller's own tensor, not the buffer the next step rewrites
        st1 = opts[1].state[pg[train[1]]]
        st1["step"] = st1["step"] + 5
        before = [int(opts[1].state[pg[n]]["step"]) for n in train[:3]]
        with pytest.raises(native.NativeError, match="different step counts"):
            opts[1].clip_and_step(0.1)
        assert [int(opts[1].state[pg[n]]["step"]) for n in train[:3]] == before
        st1["step"] = st1["step"] - 5
        v = float(kept)
        for n in train[:3]:
            pg[n].grad = torch.ones_like(pg[n])
        opts[1].clip_and_step(0.1)
        assert float(kept) == v and float(opts[1].last_norm[0]) != v


def test_teacher_ahead_is_invalidated_by_set_teacher():
    """graphs captured for one teacher must not be replayed for the next one (task t+1 swaps the
    teacher in; same batch signature), and a batch launched for another image must not be consumed."""
    cfg, m = _build(seed=6, num_query=50)
    m.to("cuda:0").train()
    data, _ = _batch(torch.device("cuda:0"))
    ahead = m.teacher_ahead()
    ahead.use_graphs, ahead.graph_warmup = True, 1
    for _ in range(4):                                     # capture happens here
        ahead.launch(data["img"], data["img_metas"])
        ti_old = ahead.finish(data["img"], data["img_metas"])
    assert any(ahead._graphs.values())
    old_feat = ti_old["neck_feats"][0].clone()
    # new teacher with clearly different weights, same batch signature
    t2 = copy.deepcopy(m.teacher_model)
    with torch.no_grad():
        for p in t2.parameters():
            p.mul_(1.5)
    m.set_teacher(model=t2)
    assert m.__dict__.get("_teacher_ahead") is None
    ahead2 = m.teacher_ahead()
    assert ahead2 is not ahead and not ahead2._graphs
    ahead2.use_graphs, ahead2.graph_warmup = True, 1
    feats_inline, *_ = m.out_teacher(data["img"], data["img_metas"])
    for _ in range(4):
        ahead2.launch(data["img"], data["img_metas"])
        ti_new = ahead2.finish(data["img"], data["img_metas"])
        torch.testing.assert_close(ti_new["neck_feats"][0], feats_inline[0], rtol=1e-3, atol=1e-3)
    assert float((ti_new["neck_feats"][0] - old_feat).abs().max()) > 1e-3
    # a pending batch that is not the one being trained on is discarded (inline teacher on the given image)
    other = torch.randn_like(data["img"])
    ahead2.launch(data["img"], data["img_metas"])
    ti_other = ahead2.finish(other, data["img_metas"])
    feats_other, *_ = m.out_teacher(other, data["img_metas"])
    torch.testing.assert_close(ti_other["neck_feats"][0], feats_other[0], rtol=1e-4, atol=1e-4)
    assert ahead2.pending is None


def _head_only_run(m, feats, data, ti, graphed):
    """Student head (transformer + branches + loss) forward and backward on GIVEN neck features: log_vars, the gradient of
    every head parameter and of the features."""
    m.bbox_head.graph_head = graphed
    for p in m.parameters():
        p.grad = None
    x = [f.detach().clone().requires_grad_(True) for f in feats]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses = m.bbox_head.forward_train(x, data["img_metas"], data["gt_bboxes"], data["gt_labels"], None,
                                           proposal_cfg=None, teacher_info=ti, task_labels=m.LableInPCNTask)
        loss, log_vars = m._parse_losses(losses)
    loss.backward()
    grads = {n: p.grad.detach().float().clone() for n, p in m.bbox_head.named_parameters() if p.grad is not None}
    grads.update({f"feat{i}": t.grad.detach().float().clone() for i, t in enumerate(x)})
    return log_vars, grads


def test_graphed_student_head_equals_eager_on_identical_features():
    """The graphed region is the student HEAD (transformer + branches: two hipGraph replays, utils.GraphedFunction with the
    parameters read from persistent low-precision buffers), so the comparison is made on the head alone: ``extract_feat`` and
    the teacher run ONCE per step and the SAME neck features go to the graphed and to the eager head.  The head's forward
    kernels are deterministic (no float atomics), so every loss term must agree to 1e-3 (bit-equal in practice); the backward
    has float atomics (MSDA grad_value, bias-gradient column sums): every gradient's cosine >= 0.9999.  Two un-padded
    shapes are captured and then ALTERNATED: a replay of the first signature must not read tensors a cache freed when
    the second shape arrived (positional encodings, reference points, ones rows, the MSDA workspace)."""
    dev = torch.device("cuda:0")
    cfg, m = _build(seed=13)                  # 300 queries: _batch()'s injected keepid (305) addresses 2 x 300 rows
    m.to(dev).train()
    g = torch.Generator().manual_seed(31)
    shapes = [(192, 256), (160, 224)]
    batches = {hw: _batch(dev, H=hw[0], W=hw[1]) for hw in shapes}
    order = [0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1]        # per shape: two eager calls, capture, replay; then alternate
    worst = 0.0
    for step, si in enumerate(order):
        hw = shapes[si]
        data, inj = batches[hw]
        img = torch.randn(2, 3, *hw, generator=g).to(dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            tfeats, touts, *_ = m.out_teacher(img, data["img_metas"])
            ti = dict(neck_feats=tfeats, head_outs=touts, pred_keepid=inj["pred_keepid"], pred_logits=None,
                      pred_scores=None, pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
            feats = [f.detach() for f in m.extract_feat(img)]
        lg, gg = _head_only_run(m, feats, data, ti, True)
        le, ge = _head_only_run(m, feats, data, ti, False)
        assert set(lg) == set(le) and set(gg) == set(ge)
        for k in le:
            assert lg[k] == pytest.approx(le[k], rel=1e-3, abs=1e-5), (step, hw, k, le[k], lg[k])
            worst = max(worst, abs(lg[k] - le[k]) / max(abs(le[k]), 1e-5))
        for n in ge:
            if float(ge[n].norm()) > 1e-8 and ge[n].numel() >= 64:
                c = float(torch.dot(ge[n].flatten(), gg[n].flatten()) / (ge[n].norm() * gg[n].norm() + 1e-30))
                assert c >= 0.9999, (step, hw, n, c)
    hg = m.bbox_head.__dict__.get("_head_graphs", {})
    assert len(hg) == 2 and all(v not in (None, False) for v in hg.values()), hg
    print("worst relative loss-term difference graphed vs eager:", worst)


def test_graphed_student_head_equals_eager(monkeypatch):
    """Whole-step companion of the head-only test above (backbone, neck and teacher run twice, once per path).  Two EAGER
    whole steps on the same image are not bit-equal by default: at this small image MIOpen runs the trunk's convolutions on
    its split-K ``igemm_fwd_gtcx35_nhwc_bf16`` kernels, which accumulate through float atomics in an f32 workspace
    (profiles/r03_determinism_convs.log: 7 of 7 repeated calls differ), and an untrained detector sits on assignment
    near-ties, so the rounding flips an assignment in one decoder layer now and then (the 2-5 % per-term spread seen in
    round 2).  ``torch.backends.cudnn.deterministic`` pins the library to its deterministic solvers
    (profiles/r03_determinism_flag.log: 0 of 213 module outputs differ), so the comparison is tight again: every loss term
    to 2e-3, every gradient's cosine > 0.999.  With dropout on, a replay draws new masks every step."""
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    dev = torch.device("cuda:0")
    cfg, m = _build(seed=13)
    m.to(dev).train()
    g = torch.Generator().manual_seed(31)
    data, inj = _batch(dev)

    def run(img, graphed):
        m.bbox_head.graph_head = graphed
        for p in m.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats, outs, *_ = m.out_teacher(img, data["img_metas"])
            ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=inj["pred_keepid"], pred_logits=None,
                      pred_scores=None, pred_labels=inj["pred_labels"], pred_bboxes=inj["pred_bboxes"])
            out = m.train_step(dict(data, img=img, teacher_info=ti))
        out["loss"].backward()
        return out["log_vars"], {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}

    run(torch.randn(2, 3, 192, 256, generator=g).to(dev), False)      # the libraries pick their algorithms on the first call
    for step in range(6):
        img = torch.randn(2, 3, 192, 256, generator=g).to(dev)
        lg, gg = run(img, True)               # eager for the first two calls, then captured and replayed
        le, ge = run(img, False)
        assert set(lg) == set(le) and set(gg) == set(ge)
        for k in le:
            assert lg[k] == pytest.approx(le[k], rel=2e-3, abs=1e-5), (step, k, le[k], lg[k])
        cos = sorted(float(torch.dot(ge[n].flatten(), gg[n].flatten()) / (ge[n].norm() * gg[n].norm() + 1e-30))
                     for n in ge if float(ge[n].norm()) > 1e-8 and ge[n].numel() >= 64)
        assert cos[0] > 0.999, (step, cos[:5])
    hg = m.bbox_head.__dict__.get("_head_graphs", {})
    assert len(hg) == 1 and all(v not in (None, False) for v in hg.values()), hg

    # dropout on: replays of the same input differ (new masks), and the loss stays finite
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
    img = torch.randn(2, 3, 192, 256, generator=g).to(dev)
    vals = [run(img, True)[0]["loss"] for _ in range(5)]
    assert len(m.bbox_head.__dict__["_head_graphs"]) == 2           # a second signature (dropout p) was captured
    assert all(v == v and abs(v) < 1e4 for v in vals)
    assert len(set(round(v, 6) for v in vals[3:])) == 2, vals       # replays (calls 4, 5) draw different masks


@pytest.mark.parametrize("case", ["random", "ties_and_misses", "no_positive"])
def test_fused_dense_losses_match_the_pytorch_formulation(case):
    """csrc/denseloss.hip (QFL / L1 / GIoU / DFL of all decoder layers in two launches, gradients in one) against
    ``loss_layers_dense`` written with PyTorch ops (the formulation the reference goldens pin on the CPU): the four
    per-layer losses and the gradients w.r.t. logits, boxes and distribution logits under random upstream weights.
    Cases: random boxes; predictions equal to / disjoint from their targets (max / min ties, clamped overlaps and
    enclosing boxes); a batch with no positive at all."""
    dev = torch.device("cuda:0")
    cfg, m = _build(seed=3)
    head = m.bbox_head.to(dev)
    g = torch.Generator().manual_seed(17)
    nl, N, C, R1 = 6, 2 * 300, head.cls_out_channels, head.reg_max + 1
    cls = (torch.randn(nl, N, C, generator=g) * 2).to(dev)
    cxy = torch.rand(nl, N, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(nl, N, 2, generator=g) * 0.3 + 0.05
    box = torch.cat([cxy, wh], -1).to(dev)
    lrtb = (torch.randn(nl, N, 4 * R1, generator=g) * 1.5).to(dev)
    pos = (torch.rand(nl, N, generator=g) < (0.0 if case == "no_positive" else 0.08)).to(dev)
    labels = torch.where(pos, torch.randint(0, C, (nl, N), generator=g).to(dev), torch.full((nl, N), C, device=dev))
    tgt = torch.cat([cxy + (torch.rand(nl, N, 2, generator=g) - 0.5) * 0.2, wh * (0.6 + 0.8 * torch.rand(nl, N, 2, generator=g))], -1).to(dev)
    if case == "ties_and_misses":
        sel = pos.nonzero()
        tgt[sel[0::3, 0], sel[0::3, 1]] = box[sel[0::3, 0], sel[0::3, 1]]                       # identical boxes: ties everywhere
        tgt[sel[1::3, 0], sel[1::3, 1], :2] = box[sel[1::3, 0], sel[1::3, 1], :2] + 0.5           # disjoint: overlap clamped to 0
    tgt = torch.where(pos[..., None], tgt, torch.zeros_like(tgt))
    factors = torch.tensor([[1333., 800., 1333., 800.]] * 300 + [[1200., 750., 1200., 750.]] * 300, device=dev)
    avg_pos = torch.tensor(max(float(pos.sum()) / nl, 1.0), device=dev)
    up = [torch.rand(nl, generator=g).to(dev) + 0.5 for _ in range(4)]
    res = []
    for fused in (True, False):
        head.fused_dense_losses = fused
        head.graph_dense_losses = False
        ins = [t.clone().requires_grad_(True) for t in (cls, box, lrtb)]
        out = head._dense_losses(ins[0], ins[1], ins[2], labels, tgt, pos, factors, avg_pos)
        total = sum((o * u).sum() for o, u in zip(out, up))
        grads = torch.autograd.grad(total, ins)
        res.append(([o.detach() for o in out], grads))
    for a, b in zip(res[0][0], res[1][0]):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-6)
    for name, a, b in zip(("cls", "box", "lrtb"), res[0][1], res[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-7 + 2e-5 * float(b.abs().max()), msg=lambda s_: f"{name}: {s_}")


def test_gfl_distillation_step_on_gpu_vs_cpu_oracle(oracle_checker):
    """BASELINE.json configs[4] (GFL R50-FPN, CNN-head distillation path) on the GPU: the trunk's outputs against the
    CPU run of the same weights, then -- on identical head inputs -- the stock GFL losses (ATSS targets, QFL / DFL /
    GIoU; pinned to the reference on the CPU by tests/test_gfl.py) and the DSKD feature-map term through the HIP
    kernel (``dskd_fgkd_fwd`` on the five pyramid levels) against the CPU oracle, values and the gradient that
    reaches the student's pyramid."""
    from dskd_amd.gfl_head import GFLHead
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "dskd_gfl_r50_fpn_40_40.py"))
    torch.manual_seed(2)
    m_cpu = build_detector(cfg.model)
    m_cpu.init_weights()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():                                  # features far from the N(0, 0.01) initialisation
        for p in m_cpu.neck.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
    t = copy.deepcopy(m_cpu)
    with torch.no_grad():
        for p in t.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 2e-2)
    m_cpu.set_teacher(model=t)
    m_cpu.LableInPCNTask = {"prev": list(range(40)), "curr": list(range(40, 80)), "next": []}
    m_gpu = copy.deepcopy(m_cpu).to("cuda:0").train()
    m_cpu.train()
    dev, cpu = torch.device("cuda:0"), torch.device("cpu")
    B, H, W = 2, 192, 256
    img = torch.randn(B, 3, H, W, generator=g)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0) for _ in range(B)]
    gt_b = [torch.tensor([[10., 12., 90., 100.], [30., 20., 200., 150.]]), torch.tensor([[5., 5., 120., 90.]])]
    gt_l = [torch.tensor([45, 71]), torch.tensor([79])]
    t_b = [torch.tensor([[20., 20., 120., 110.], [0., 0., 60., 70.]]), torch.tensor([[40., 40., 200., 160.]])]

    def to(x, d):
        if torch.is_tensor(x):
            return x.detach().to(d)
        if isinstance(x, (list, tuple)):
            return type(x)(to(y, d) for y in x)
        if isinstance(x, dict):
            return {k: to(v, d) for k, v in x.items()}
        return x
    # trunk: GPU vs CPU
    with torch.no_grad():
        xc = m_cpu.extract_feat(img)
        oc = m_cpu.bbox_head.forward(xc)
    xg = m_gpu.extract_feat(img.to(dev))
    og = m_gpu.bbox_head.forward(xg)
    for a, b in zip(list(xg) + list(og[0]) + list(og[1]), list(xc) + list(oc[0]) + list(oc[1])):
        torch.testing.assert_close(a.detach().cpu(), b, rtol=2e-3, atol=2e-4)
    with torch.no_grad():
        ft = m_gpu.teacher_model.extract_feat(img.to(dev))
    ti = dict(neck_feats=ft, pred_bboxes=to(t_b, dev))
    assert isinstance(m_gpu.bbox_head, GFLHead)
    # losses on identical head inputs
    lg = m_gpu.bbox_head.loss(*og, to(gt_b, dev), to(gt_l, dev), metas)
    fg_g = m_gpu.bbox_head.fg_feature_loss(xg, ti, to(gt_b, dev), metas)
    xs_c = [x.detach().cpu().requires_grad_(True) for x in xg]
    native.install_cpu_checker(oracle_checker)
    try:
        lc = m_cpu.bbox_head.loss(*to(og, cpu), gt_b, gt_l, metas)
        fg_c = m_cpu.bbox_head.fg_feature_loss(xs_c, to(ti, cpu), gt_b, metas)
        gc = torch.autograd.grad(fg_c, xs_c, allow_unused=True)
    finally:
        native.install_cpu_checker(None)
    for k in lc:
        torch.testing.assert_close(torch.stack([v.detach().cpu() for v in lg[k]]), torch.stack([v.detach() for v in lc[k]]),
                                   rtol=1e-3, atol=1e-6, msg=lambda s: f"{k}: {s}")
    assert float(fg_c) > 1e-4
    torch.testing.assert_close(fg_g.detach().cpu(), fg_c.detach(), rtol=5e-2, atol=1e-6)      # fp32 noise of the CPU evaluation
    gg = torch.autograd.grad(fg_g, list(xg), allow_unused=True)
    for a, b in zip(gg, gc):
        assert (a is None) == (b is None)
        if a is not None:
            rel = (a.cpu() - b).norm() / (b.norm() + 1e-12)
            assert rel < 5e-2, float(rel)
    # and a whole training step runs on the GPU
    out = m_gpu.train_step(dict(img=img.to(dev), img_metas=metas, gt_bboxes=to(gt_b, dev), gt_labels=to(gt_l, dev),
                                teacher_info=dict(ti, head_outs=None, pred_keepid=None, pred_logits=None, pred_scores=None,
                                                  pred_labels=None)))
    out["loss"].backward()
    assert out["log_vars"]["loss_fg_feature"] > 0 and all(v == v for v in out["log_vars"].values())


def test_gfl_teacher_one_batch_ahead_matches_inline_teacher():
    """BASELINE.json configs[4]: the GFL detector's frozen teacher run one batch ahead on a second stream -- eagerly first,
    as hipGraph replays once the batch signature has repeated -- hands over what the inline ``out_teacher`` computes
    (pyramid features and head outputs to rounding: MIOpen may pick another algorithm; the decode + NMS through its sizes
    and the per-image offsets of ``pred_keepid``), with a NEW image every step."""
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "dskd_gfl_r50_fpn_40_40.py"))
    torch.manual_seed(4)
    m = build_detector(cfg.model)
    m.init_weights()
    t = copy.deepcopy(m)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p in t.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 2e-2)
    m.set_teacher(model=t)
    m.to("cuda:0").train()
    B, H, W = 2, 192, 256
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0, batch_input_shape=(H, W)) for _ in range(B)]
    ahead = m.teacher_ahead()
    ahead.use_graphs, ahead.graph_warmup = True, 2
    for step in range(6):
        img = torch.randn(B, 3, H, W, generator=g).to("cuda:0")
        feats, outs, keepid, logits, labels, scores, bboxes = m.out_teacher(img, metas)
        ahead.launch(img, metas)
        ti = ahead.finish(img, metas)
        for a, b in zip(ti["neck_feats"], feats):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-3)
        for a, b in zip(list(ti["head_outs"][0]) + list(ti["head_outs"][1]), list(outs[0]) + list(outs[1])):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-3)
        assert [x.shape[1] for x in ti["pred_bboxes"]] == [4] * B and len(ti["pred_labels"]) == B
        n_prior = sum(int(c.shape[-2] * c.shape[-1]) for c in outs[0])
        assert m._keepid_stride(outs) == n_prior
        if ti["pred_keepid"].numel():
            assert int(ti["pred_keepid"].max()) < B * n_prior
    entries = [v for v in ahead._graphs.values() if v]
    assert len(entries) == 1 and len(entries[0]) == 2        # captured, verified against eager, double-buffered
