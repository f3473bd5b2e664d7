"""N>1 path on CPU: two ranks over gloo run one DDP step of the distillation loss; gradients
agree across ranks, the normaliser is the cross-rank mean, log scalars come from one
coalesced all-reduce (SURVEY.md sections 2.3 and 8e)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import copy
    import dskd_amd  # noqa: F401
    from dskd_amd import native
    from dskd_amd.builder import build_detector
    from dskd_amd.config import Config
    from dskd_amd.dist import allreduce_scalars, init_dist, reduce_mean, wrap_ddp
    from oracle.checker import OracleChecker
    native.install_cpu_checker(OracleChecker())
    init_dist("pytorch", backend="gloo")
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_70_10.py"))
    cfg.model.bbox_head.num_query = 20
    torch.manual_seed(0)                                  # identical init on both ranks
    model = build_detector(cfg.model)
    model.init_weights()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    model.set_teacher(model=copy.deepcopy(model))
    model.LableInPCNTask = {"prev": list(range(70)), "curr": list(range(70, 80)), "next": []}
    model.train()
    ddp = wrap_ddp(model)
    g = torch.Generator().manual_seed(100 + rank)          # different data per rank
    H, W = 64, 96
    img = torch.randn(1, 3, H, W, generator=g)
    metas = [dict(img_shape=(H, W, 3), batch_input_shape=(H, W), scale_factor=1.0)]
    n_gt = 1 + rank                                         # rank-dependent positives
    gt_b = [torch.tensor([[4., 4., 40., 30.], [20., 10., 80., 60.]])[:n_gt]]
    gt_l = [torch.tensor([75, 71])[:n_gt]]
    feats, outs, *_ = model.out_teacher(img, metas)
    ti = dict(neck_feats=feats, head_outs=outs, pred_keepid=torch.tensor([3]), pred_logits=None, pred_scores=None,
              pred_labels=[torch.tensor([5])], pred_bboxes=[torch.tensor([[10., 10., 50., 40.]])])
    losses = ddp(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l, teacher_info=ti)
    loss, log_vars = model._parse_losses(losses)
    loss.backward()
    gsum = torch.stack([p.grad.double().sum() for p in model.parameters() if p.grad is not None]).sum()
    gathered = [torch.zeros_like(gsum) for _ in range(world)]
    dist.all_gather(gathered, gsum)
    # ---- the graph-step driver's data-parallel exchange (eager on CPU): ONE flat gradient
    # all-reduce, no DDP wrapper; parameters must stay identical across ranks
    from dskd_amd.graph_step import GraphedDistillStep
    from dskd_amd.runner import build_optimizer
    torch.manual_seed(1)
    m2 = build_detector(cfg.model)
    m2.init_weights()
    for mm in m2.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
        if isinstance(mm, torch.nn.MultiheadAttention):
            mm.dropout = 0.0
    m2.set_teacher(model=copy.deepcopy(m2))
    m2.LableInPCNTask = {"prev": list(range(70)), "curr": list(range(70, 80)), "next": []}
    m2.train()
    opt2 = build_optimizer(m2, cfg.optimizer[0])
    stepper = GraphedDistillStep(m2, opt2, amp_dtype=None, max_norm=0.1)
    data = dict(img=img, img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l)
    inject = dict(pred_bboxes=ti["pred_bboxes"], pred_labels=ti["pred_labels"], pred_keepid=ti["pred_keepid"])
    for _ in range(2):
        stepper.step(data, inject)
    psum = torch.stack([p.detach().double().sum() for p in m2.parameters()]).sum()
    pg = [torch.zeros_like(psum) for _ in range(world)]
    dist.all_gather(pg, psum)
    step_logs = stepper.logs()
    rm = reduce_mean(torch.tensor([float(rank + 1)]))
    sc = allreduce_scalars([torch.tensor(float(rank)), torch.tensor(2.0)])
    if rank == 0:
        out.put(dict(psums=[float(x) for x in pg], step_keys=sorted(step_logs.keys()),
                     gsums=[float(x) for x in gathered], reduce_mean=float(rm), scalars=sc.tolist(),
                     keys=sorted(log_vars.keys()), loss=float(log_vars["loss"]), local_loss=float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_step_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res["gsums"][0] == pytest.approx(res["gsums"][1], rel=1e-9)        # DDP averaged the gradients
    assert res["psums"][0] == pytest.approx(res["psums"][1], rel=1e-12)        # stepper keeps ranks in sync
    assert "loss" in res["step_keys"] and "loss_fg_feature" in res["step_keys"]
    assert res["reduce_mean"] == pytest.approx(1.5)
    assert res["scalars"] == [pytest.approx(0.5), pytest.approx(2.0)]
    assert "loss_corr" in res["keys"] and "loss_fg_feature" in res["keys"] and "d4.loss_dfl" in res["keys"]
    assert res["loss"] != pytest.approx(res["local_loss"], rel=1e-6)          # logged value is the cross-rank mean
