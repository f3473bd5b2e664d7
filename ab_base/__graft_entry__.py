"""Driver entry points: build() compiles every native piece; smoke() runs one small
invocation of the hot path on cuda:0 and checks it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """hipcc --offload-arch=gfx950 for the HIP library (dskd_amd/_C/libdskd_hip.so, in-tree) and
    gcc for the oracle's C restatement (building the checker is not using it).  The reference
    is pure Python, so there is no oracle/_ref build."""
    subprocess.run(["bash", os.path.join(ROOT, "dskd_amd", "csrc", "build.sh")], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    import dskd_amd  # noqa: F401
    from dskd_amd import native
    lib = native.load()
    for sym in native.EXPORTED_SYMBOLS:
        getattr(lib, sym)


def smoke() -> None:
    """One tiny teacher+student distillation step (forward + backward) on cuda:0 through the
    HIP kernels, plus a direct kernel-vs-oracle check of every entry point."""
    import copy

    import numpy as np
    import torch

    import dskd_amd  # noqa: F401
    from dskd_amd import native
    from dskd_amd.builder import build_detector
    from dskd_amd.config import Config
    from oracle import msda_ref
    from oracle.lsap_ref import linear_sum_assignment as oracle_lsa

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda:0")
    native.load()
    # --- kernel vs oracle: MSDA forward/backward
    shapes = [(12, 17), (6, 9), (3, 5), (2, 3)]
    g = torch.Generator().manual_seed(0)
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(2, Nv, 8, 32, generator=g)
    loc = torch.rand(2, 50, 8, 4, 4, 2, generator=g) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(2, 50, 8, 16, generator=g), -1).view(2, 50, 8, 4, 4)
    vr, lr, ar = value.clone().requires_grad_(True), loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    ref = msda_ref.msda_grid_sample(vr, shapes, lr, ar)
    ref.square().sum().backward()
    vd, ld, ad = (t.to(dev).requires_grad_(True) for t in (value, loc, attn))
    out = native.ms_deform_attn(vd, shapes, ld, ad)
    out.square().sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(vd.grad.cpu(), vr.grad, atol=1e-4, rtol=1e-3)
    # --- LSAP bit-exact vs the oracle
    rng = np.random.default_rng(0)
    mats = [rng.random((300, 17)).astype(np.float32), rng.integers(0, 4, (300, 9)).astype(np.float32)]
    flat = torch.cat([torch.from_numpy(m).reshape(-1) for m in mats]).to(dev)
    row, col, outs, status = native.lsap_batched(flat, [300, 300], [17, 9], [0, 300 * 17])
    row, col = row.cpu().numpy(), col.cpu().numpy()
    for p, m in enumerate(mats):
        r, c = oracle_lsa(m)
        n = min(m.shape)
        assert np.array_equal(row[outs[p]:outs[p] + n], r) and np.array_equal(col[outs[p]:outs[p] + n], c)
    # --- fused sub-layer tail (add + LayerNorm + query_pos) and conv epilogue vs the oracle chain
    from oracle.checker import OracleChecker
    chk = OracleChecker()
    h, res_ = torch.randn(2, 37, 256, generator=g), torch.randn(2, 37, 256, generator=g)
    pos = torch.randn(1, 37, 256, generator=g)
    norm = torch.nn.LayerNorm(256)
    y_ref, q_ref = chk.add_layer_norm(h, res_, norm, 0.0, pos, True)
    y, q = native.add_layer_norm(h.to(dev), res_.to(dev), copy.deepcopy(norm).to(dev), 0.0, pos.to(dev), True)
    torch.testing.assert_close(y.cpu(), y_ref.detach(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(q.cpu(), q_ref.detach(), atol=2e-5, rtol=1e-5)
    x4 = torch.randn(2, 64, 5, 7, generator=g).contiguous(memory_format=torch.channels_last)
    b4 = torch.randn(64, generator=g)
    torch.testing.assert_close(native.bias_act(x4.to(dev), b4.to(dev), x4.to(dev), True).cpu(),
                               chk.bias_act(x4, b4, x4, True), atol=1e-6, rtol=1e-6)
    # --- one tiny end-to-end distillation step
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_70_10.py"))
    torch.manual_seed(0)
    model = build_detector(cfg.model)
    model.init_weights()
    teacher = copy.deepcopy(model)
    gp = torch.Generator().manual_seed(1)
    with torch.no_grad():                      # a teacher that differs from the student, so that both DSKD losses are non-zero
        for prm in teacher.parameters():
            prm.add_(torch.randn(prm.shape, generator=gp) * 1e-2)
    model.set_teacher(model=teacher)
    model.LableInPCNTask = {"prev": list(range(70)), "curr": list(range(70, 80)), "next": []}
    model.to(dev).train()
    B, H, W = 2, 160, 224
    img = torch.randn(B, 3, H, W, device=dev)
    metas = [dict(img_shape=(H, W, 3), batch_input_shape=(H, W), scale_factor=1.0) for _ in range(B)]
    gt_b = [torch.tensor([[10., 12., 60., 70.], [30., 20., 120., 100.]], device=dev), torch.tensor([[5., 5., 50., 40.]], device=dev)]
    gt_l = [torch.tensor([75, 71], device=dev), torch.tensor([79], device=dev)]
    feats, outs_t, *_ = model.out_teacher(img, metas)
    ti = dict(neck_feats=feats, head_outs=outs_t, pred_keepid=torch.tensor([3, 17, 305], device=dev), pred_logits=None,
              pred_scores=None, pred_labels=[torch.tensor([1, 7], device=dev), torch.tensor([3], device=dev)],
              pred_bboxes=[torch.tensor([[20., 20., 80., 90.], [0., 0., 30., 30.]], device=dev),
                           torch.tensor([[40., 40., 100., 120.]], device=dev)])
    res = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=ti))
    res["loss"].backward()
    lv = res["log_vars"]
    assert np.isfinite(lv["loss"]) and lv["loss_corr"] > 0 and lv["loss_fg_feature"] > 0, lv
    # the head's loss() on the GPU (HIP cost / LSAP / DSKD kernels) against loss() on the CPU with the oracle injected as
    # the checker, fed with the SAME head inputs (the GPU's, copied): an untrained detector sits on assignment
    # near-ties, so two trunks that differ by rounding would be matched differently
    from oracle.checker import OracleChecker as _OC
    cpu = torch.device("cpu")

    def tc(x):
        if torch.is_tensor(x):
            return x.detach().to(cpu)
        if isinstance(x, (list, tuple)):
            return type(x)(tc(y) for y in x)
        if isinstance(x, dict):
            return {k: tc(v) for k, v in x.items()}
        return x
    with torch.no_grad():
        xg = model.extract_feat(img)
        og = model.bbox_head.forward(xg, metas)
    lg = model.bbox_head.loss(*og, gt_b, gt_l, metas, student_feat=xg, teacher_info=ti, task_labels=model.LableInPCNTask)
    model_c = copy.deepcopy(model).to(cpu).train()
    native.install_cpu_checker(_OC())
    try:
        lc = model_c.bbox_head.loss(*tc(og), tc(gt_b), tc(gt_l), metas, student_feat=tc(xg), teacher_info=tc(ti),
                                    task_labels=model.LableInPCNTask)
    finally:
        native.install_cpu_checker(None)
    assert set(lg) == set(lc)
    for k in lc:
        a_, b_ = float(lg[k]), float(lc[k])
        tol, atol = (5e-2, 2e-5) if k == "loss_fg_feature" else (2e-3, 1e-6)
        assert abs(a_ - b_) <= tol * abs(b_) + atol, (k, a_, b_)
    torch.cuda.synchronize()
    print("smoke ok:", {k: round(v, 4) for k, v in res["log_vars"].items() if not k.startswith("d")})


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
