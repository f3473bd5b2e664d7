#!/usr/bin/env python
"""Headline benchmark: images/sec of one full teacher+student DSKD distillation training step
(BASELINE.json metric) on synthetic COCO-shaped 800x1333 batches.

  python bench.py --gpus N --steps K --warmup W

N>1 either way: under a launcher (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or WITHOUT one -- ``python bench.py --gpus N`` finds no
WORLD_SIZE, starts N fresh rank processes itself (one per GPU, before this process has touched the GPU; the
reference launches the same way, tools/dist_train_increment.sh:22-28), relays rank 0's JSON line and exits non-zero
if any rank failed.

Workload (config.workload): BASELINE.json configs[1] -- Deformable-DETR R50 70+10 incremental,
bf16 autocast for conv/GEMM (fp32 losses, costs, LSAP, MSDA accumulation), batch 4 per GPU.
A step = teacher forward (no grad) + student forward + Hungarian targets + detection losses +
both DSKD losses + backward + gradient all-reduce (N>1) + grad-clip + AdamW update; nothing is
skipped or cached.  Inputs are resident in HBM before the timed region.  An untrained teacher
emits no score > 0.3, so 10 synthetic teacher detections per image are injected through the
same ``teacher_info`` dict after the real teacher forward + decode.

The JSON line carries ``roofline`` (dominant hand-written kernel: MSDeformAttn, algorithmic
bytes of SURVEY.md section 8d / measured HIP-event time of that launch) and ``cpu_baseline``
(the CPU oracle restatement of the same step, B=1, timed on this host; rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MIOpen's default find mode (dynamic hybrid) answers a find-db miss with heuristics and only the
# NEXT process gets the measured choice: on a fresh box the first run was 55 ms/step and erratic,
# every later one 43 ms.  NORMAL makes torch.backends.cudnn.benchmark really measure, inside the
# untimed warm-up (about +50 s of start-up).  Must be set before MIOpen initialises.
os.environ.setdefault("MIOPEN_FIND_MODE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import dskd_amd  # noqa: E402,F401
from dskd_amd import native  # noqa: E402
from dskd_amd.builder import build_detector  # noqa: E402
from dskd_amd.config import Config  # noqa: E402
from dskd_amd.graph_step import GraphedDistillStep  # noqa: E402
from dskd_amd.runner import build_optimizer  # noqa: E402

CONFIGS = {"r50": os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_r50_70_10.py"),           # BASELINE configs[1]
           "swin_t": os.path.join(ROOT, "configs", "dskd_gfl_deformable_detr_swin_t_40_40.py"),      # BASELINE configs[3]
           "gfl_r50": os.path.join(ROOT, "configs", "dskd_gfl_r50_fpn_40_40.py")}                   # BASELINE configs[4]
CONFIG = CONFIGS["r50"]
IMG_H, IMG_W = 800, 1333
LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]
NV = sum(h * w for h, w in LEVELS)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s float4-copy measured)


def msda_algorithmic_bytes(kind, B, Nq, esz):
    """compulsory unique bytes of one launch (esz = value/out element
    size; loc 1024 B and attn 512 B per query are fp32)."""
    fwd = B * (NV * 256 * esz + Nq * 1024 + Nq * 512 + Nq * 256 * esz)
    if kind == "fwd_fused":      # prologue folded in: the projection output (384 values) + reference points replace loc/attn
        return B * (NV * 256 * esz + Nq * (384 * esz + 4 * 2 * 4) + Nq * 256 * esz)
    if kind == "fwd":
        return fwd
    # bwd = fwd - out + grad_out + grad_value(fp32) + grad_loc + grad_attn
    return fwd + B * (NV * 256 * 4 + Nq * 1024 + Nq * 512)


def build_models(device, seed, dropout, config=None):
    cfg = Config.fromfile(config or CONFIG)
    torch.manual_seed(seed)
    model = build_detector(cfg.model)
    model.init_weights()
    if dropout is not None:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = dropout
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = dropout
    teacher = copy.deepcopy(model)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in teacher.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 1e-3)     # so that hs_t != hs_s
    model.set_teacher(model=teacher)
    prev = list(range(cfg.num_prev))
    model.LableInPCNTask = {"prev": prev, "curr": list(range(cfg.num_prev, 80)), "next": []}
    model.to(device)
    model.train()
    return cfg, model


def make_batch(B, num_prev, seed, device, n_gt=7, n_t=10):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, IMG_H, IMG_W, generator=g)

    def boxes(n):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([0.6 * IMG_W, 0.6 * IMG_H])
        lo, hi = torch.tensor([8.0, 8.0]), torch.tensor([0.35 * IMG_W, 0.35 * IMG_H])
        wh = lo + torch.rand(n, 2, generator=g) * (hi - lo)
        return torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(IMG_W), float(IMG_H)]))], 1)
    gt_b = [boxes(n_gt).to(device) for _ in range(B)]
    gt_l = [torch.randint(num_prev, 80, (n_gt,), generator=g).to(device) for _ in range(B)]
    t_b = [boxes(n_t).to(device) for _ in range(B)]
    t_l = [torch.randint(0, num_prev, (n_t,), generator=g).to(device) for _ in range(B)]
    keep = torch.cat([b * 300 + torch.randperm(300, generator=g)[:n_t] for b in range(B)]).to(device)
    metas = [dict(img_shape=(IMG_H, IMG_W, 3), batch_input_shape=(IMG_H, IMG_W), scale_factor=1.0) for _ in range(B)]
    return dict(img=img.to(device), img_metas=metas, gt_bboxes=gt_b, gt_labels=gt_l), dict(t_b=t_b, t_l=t_l, keep=keep)


def train_step(model, wrapped, optimizer, data, synth, amp_dtype, max_norm=0.1, ahead=None, gsync=None):
    """One distillation step.  ``ahead`` (a ``TeacherAhead``): the teacher of the next batch runs
    on a second stream behind the student's backward and its decode no longer drains the main
    stream; every step still contains one teacher forward + decode, one student
    forward/backward and one optimizer update."""
    module = model
    dev = data["img"].device
    optimizer.zero_grad(set_to_none=True)
    with torch.autocast(device_type=dev.type, dtype=amp_dtype, enabled=amp_dtype is not None):
        if ahead is not None:
            ti = ahead.finish(data["img"], data["img_metas"])
            feats, outs = ti["neck_feats"], ti["head_outs"]
        else:
            feats, outs, keepid, logits, labels, scores, bboxes = module.out_teacher(data["img"], data["img_metas"])
        teacher_info = {"neck_feats": feats, "head_outs": outs, "pred_keepid": synth["keep"], "pred_logits": None,
                        "pred_scores": None, "pred_labels": synth["t_l"], "pred_bboxes": synth["t_b"]}
        losses = wrapped(img=data["img"], img_metas=data["img_metas"], gt_bboxes=data["gt_bboxes"],
                         gt_labels=data["gt_labels"], teacher_info=teacher_info)
        loss, log_vars = module._parse_losses(losses)
    if ahead is not None:       # the next batch (synthetic: the same tensors), enqueued behind the backward
        ahead.launch(data["img"], data["img_metas"], amp_dtype=amp_dtype)
    loss.backward()
    if gsync is not None:       # data parallel: the buckets' all-reduces were started by the backward's hooks
        gsync.finish()
    if hasattr(optimizer, "clip_and_step"):       # dskd_amd.optim.FusedClipAdamW: clip + AdamW of every tensor, two launches
        optimizer.clip_and_step(max_norm)
    else:
        params = [p for gr in optimizer.param_groups for p in gr["params"] if p.grad is not None]
        torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=2)
        optimizer.step()
    return loss, log_vars


def cpu_baseline(seed, num_prev, batch=1, timed=3):
    """The CPU oracle path of the same step (PyTorch CPU kernels + grid_sample MSDA + oracle LSAP + loop DSKD
    losses) on this host's cores: 1 warm-up + ``timed`` timed steps at ``batch`` images (SURVEY.md section 8d
    protocol), 800x1333, fp32.  Bounded sample: (1 + timed) image-steps at B=1 are ~30-40 s of CPU work."""
    from oracle.checker import OracleChecker
    native.install_cpu_checker(OracleChecker())
    try:
        # the GPU box gives one GPU a 16-core CPU share; os.cpu_count() reports the whole host
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 16))
        torch.set_num_threads(cores)
        print(f"[bench] cpu_baseline: 1 warm-up + {timed} timed steps, B={batch}, {cores} threads ...", file=sys.stderr,
              flush=True)
        cfg, model = build_models(torch.device("cpu"), seed, dropout=None)
        opt = build_optimizer(model, cfg.optimizer[0])
        data, synth = make_batch(batch, num_prev, seed, torch.device("cpu"))
        times = []
        for i in range(1 + timed):
            t0 = time.time()
            train_step(model, model, opt, data, synth, None)
            times.append(time.time() - t0)
            print(f"[bench] cpu_baseline step {i}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
        dt = sum(times[1:]) / max(len(times) - 1, 1)
        return {"value": round(batch / dt, 5), "unit": "images/sec", "cores": cores, "kind": "port",
                "sample": f"1 warm-up + {timed} timed full distillation steps, B={batch}, 800x1333, fp32, oracle CPU path "
                          f"({dt:.1f} s/step; warm-up {times[0]:.1f} s)"}
    finally:
        native.install_cpu_checker(None)


GEMM_CONV_PATTERNS = ("Cijk_", "ck::", "_ZN2ck", "igemm_", "miopen", "MIOpen", "gemm_xdl", "xdlops", "wrw_", "naive_conv",
                      "attn_fwd", "bwd_kernel_dk_dv", "bwd_kernel_dq", "ffn_fused_kernel", "lin256_kernel", "gemm_nt_kernel",
                      "gemm_tn_kernel", "winattn_")
# the hand-written MFMA kernels on the path: csrc/ffn_mfma.hip, csrc/gemm_nt.hip (1x1 / 3x3 convolutions, weight gradients),
# csrc/winattn.hip (Swin window attention)
OWN_MFMA_KERNELS = ("ffn_fused_kernel", "lin256_kernel", "gemm_nt_kernel", "gemm_tn_kernel", "winattn_")
MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


def mfma_utilisation(step_fn, dtype):
    """north_star: 'MFMA utilisation vs gfx950 peak' of the dense part.  One eager step under FlopCounterMode counts the
    FLOPs of every GEMM / convolution / attention call (forward and backward, teacher and student); one eager step
    under the profiler sums the device time of the library kernels that execute them; utilisation = FLOPs / that
    time / dense bf16 peak."""
    from torch.profiler import ProfilerActivity, profile
    from torch.utils.flop_counter import FlopCounterMode

    def addmm_act_flop(self_shape, a_shape, b_shape, *args, out_shape=None, **kwargs):   # bias + [m,k] x [k,n] + activation
        return 2 * a_shape[0] * a_shape[1] * b_shape[1]
    own0 = native.ffn_flops_launched()
    with FlopCounterMode(display=False, custom_mapping={torch.ops.aten._addmm_activation: addmm_act_flop}) as fc:
        step_fn()
    torch.cuda.synchronize()
    own_flops = float(native.ffn_flops_launched() - own0)          # the fused FFN launches are not aten ops
    flops = float(fc.get_total_flops()) + own_flops
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step_fn()
        torch.cuda.synchronize()
    t_us, n, total_us, names, own_us, own_n = 0.0, 0, 0.0, {}, 0.0, 0
    for e in prof.key_averages():
        dt = float(getattr(e, "self_device_time_total", 0.0) or 0.0)
        if dt <= 0 or "Memcpy" in e.key or "Memset" in e.key or e.key.startswith("aten::") or \
                getattr(e, "device_type", None) != torch.autograd.DeviceType.CUDA:      # device kernels only, not the ops that own them
            continue
        total_us += dt
        if any(pat in e.key for pat in GEMM_CONV_PATTERNS):
            t_us += dt
            n += e.count
            names[e.key[:48]] = names.get(e.key[:48], 0.0) + dt
            if any(k in e.key for k in OWN_MFMA_KERNELS):
                own_us += dt
                own_n += e.count
    if t_us <= 0:
        return None
    own = None
    if own_us > 0:
        own = {"kernel": "dskd::ffn_fused_kernel (encoder FFN) + dskd::lin256_kernel (tall 256-input Linear layers and their "
                         "dX), csrc/ffn_mfma.hip; dskd::gemm_nt_kernel (1x1 / 3x3 convolutions forward and dX) + "
                         "dskd::gemm_tn_kernel (weight gradients), csrc/gemm_nt.hip", "launches": own_n,
               "flops_TFLOP": round(own_flops / 1e12, 3), "kernel_ms": round(own_us / 1e3, 2),
               "achieved_TFLOPs": round(own_flops / (own_us * 1e-6) / 1e12, 1),
               "frac": round(own_flops / (own_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
    tflops = flops / (t_us * 1e-6) / 1e12
    peak = MFMA_PEAK_TFLOPS if dtype == "bf16" else MFMA_PEAK_TFLOPS / 16      # fp32: no MFMA-rate claim, reported for scale
    return {"flops_per_step": round(flops / 1e12, 3), "unit": "TFLOP", "gemm_conv_kernel_ms": round(t_us / 1e3, 2),
            "gemm_conv_launches": n, "all_kernel_ms": round(total_us / 1e3, 2), "achieved_TFLOPs": round(tflops, 1),
            "peak_TFLOPs": peak, "frac": round(tflops / peak, 4),
            "counted": "aten mm/addmm/bmm/convolution/sdpa (FlopCounterMode) + the hand-written MFMA launches (fused FFN: 4 * "
                       "tokens * 256 * 1024 each; lin256 / gemm_nt / gemm_tn: 2 M N K; conv3x3: 2 * pixels * N * 9 C), forward + "
                       "backward, teacher + student; time = device time of the hipBLASLt / CK / MIOpen / attention kernels and of "
                       "the hand-written MFMA kernels in one profiled eager step",
            "top_kernels_ms": {k: round(v / 1e3, 2) for k, v in top}, "hand_written": own}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: this (parent) process has NOT initialised the GPU -- it starts N
    fresh children of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (never an exec of a
    process that has touched the GPU), lets them inherit stdout / stderr (rank 0 prints the one JSON line), and
    returns the first non-zero exit code (0 if every rank succeeded)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
    rc = 0
    try:
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending.values():         # a dead rank leaves the others blocked in a collective
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def launch_check(world, rank, local_rank):
    """``--launch-check``: the rank plumbing alone (rendezvous, backend, an all-reduced rank count), no model and no
    kernels -- runs on a CPU-only host over gloo (tests/test_bench_contract.py) and on a GPU box over RCCL."""
    use_gpu = torch.cuda.device_count() >= world and not os.environ.get("DSKD_BENCH_REHEARSE")
    backend = "nccl" if use_gpu else "gloo"
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    count = torch.ones(1, device=device)
    if world > 1:
        dist.all_reduce(count)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks" if use_gpu else "gloo_ranks": int(count.item()),
                          "backend": backend if world > 1 else None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (weak scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=None, help="override dropout p (default: config, 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-batch", type=int, default=1, help="images per CPU-baseline step (SURVEY 8d: B=1 and "
                    "B=4; B=4 takes ~2.5 min more)")
    ap.add_argument("--no-mfma-probe", action="store_true", help="skip the FLOP count / profiler pass behind `mfma`")
    ap.add_argument("--graph", action="store_true", help="EXPERIMENTAL: replay the step as hipGraphs "
                    "(dskd_amd/graph_step.py) instead of eager launches + DDP")
    ap.add_argument("--no-teacher-ahead", action="store_true", help="run the teacher inline on the main stream "
                    "(its decode then drains the stream mid-step) instead of one batch ahead on a second stream")
    ap.add_argument("--no-teacher-graph", action="store_true", help="enqueue the ahead-of-time teacher forward "
                    "eagerly instead of replaying it as a hipGraph")
    ap.add_argument("--probe-steps", type=int, default=3, help="eager steps after the timed region that bracket "
                    "every MSDeformAttn launch with HIP events (roofline)")
    ap.add_argument("--backbone", default="r50", choices=sorted(CONFIGS), help="r50 = BASELINE configs[1] (the headline "
                    "workload); swin_t = configs[3] (SURVEY.md 8f row 2) and gfl_r50 = configs[4] (GFL CNN head with the "
                    "DSKD feature-map term only), reported in DESIGN.md only")
    ap.add_argument("--seed", type=int, default=111)
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, all-reduce a rank count and print "
                    "it (no model, no kernels; works on a CPU-only host over gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher.  Nothing above this line initialises HIP (torch.cuda.device_count() does not).
        if not args.launch_check and not os.environ.get("DSKD_BENCH_REHEARSE") and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this host shows {torch.cuda.device_count()} GPU(s)")
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, "
                         "or without a launcher (bench.py starts the ranks itself)")
    if args.launch_check:
        return launch_check(world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    # Rehearsal switch (one-GPU box): DSKD_BENCH_REHEARSE=1 runs all ranks on cuda:0 over gloo, to
    # exercise the multi-process path (DDP, flat collectives, per-rank teacher stream / graphs).
    rehearse = bool(os.environ.get("DSKD_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # DSKD_BENCH_DDP1=1 (diagnostic, one GPU): a ONE-rank RCCL process group and the DDP wrap around the student, so that the
    # production combination -- backend nccl, DDP's bucketed all-reduce hooks, head graphs, teacher graph on its side
    # stream -- executes on a 1-GPU box (everything except the inter-GPU transfers themselves).
    ddp1 = world == 1 and bool(os.environ.get("DSKD_BENCH_DDP1"))
    json_out = sys.stdout
    if (world > 1 or ddp1) and not rehearse:
        # RCCL prints its version banner on stdout when the first communicator is created; the contract is ONE JSON line
        # on stdout: everything a rank (or a library inside it) writes to fd 1 goes to stderr, the JSON line to the real stdout
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if world > 1 or ddp1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
    native.load()

    torch.backends.cudnn.benchmark = True
    amp_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    cfg, model = build_models(device, args.seed, args.dropout, CONFIGS[args.backbone])
    model = model.to(memory_format=torch.channels_last)
    model.teacher_model.to(memory_format=torch.channels_last)
    model.lazy_log = True                      # log scalars stay on the device inside the timed loop
    data, synth = make_batch(args.batch, cfg.num_prev, args.seed + rank, device)
    data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
    inject = {"pred_bboxes": synth["t_b"], "pred_labels": synth["t_l"], "pred_keepid": synth["keep"]}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cdev = torch.device("cpu") if rehearse else device      # small collectives: host tensors over gloo (dist.all_reduce_sum)

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # Default execution: eager launches (the step has no host<->device synchronisation apart
    # from the teacher decode, so the host runs ahead of the GPU) with DDP's bucketed all-reduce
    # overlapped with backward.  --graph: hipGraph replay of the step, gradients exchanged as
    # ONE flat all-reduce over RCCL; falls back to eager if capture fails on any rank.
    mode = "hipgraph" if args.graph else "eager+ddp"
    gsync = None
    stepper = None
    ahead = None
    extra_warmup = 0
    if mode == "hipgraph":
        optimizer = build_optimizer(model, cfg.optimizer[0], capturable=True)
        stepper = GraphedDistillStep(model, optimizer, amp_dtype=amp_dtype, max_norm=0.1, use_graphs=True, warmup=3)
        ok = True
        try:
            n_warm = max(args.warmup, stepper.warmup + 2)       # capture happens inside the untimed warm-up
            extra_warmup = n_warm - args.warmup
            for _ in range(n_warm):
                loss = stepper.step(data, inject)
            torch.cuda.synchronize()
            ok = len(stepper._graphs) == 1 and bool(torch.isfinite(loss).item())
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager",
                  file=sys.stderr, flush=True)
            ok = False
        if not all_ok(ok):
            mode = "eager+ddp(fallback)"
            stepper = None
            model.bbox_head.avg_pos_static = None
            for p in model.parameters():
                p.grad = None
    if stepper is None:
        wrapped = model
        gsync = None
        if world > 1 or ddp1:
            # gradient exchange: flat buckets + hooks (dist.GradSync), not the DistributedDataParallel wrapper -- whose
            # per-parameter bucket copies cost 3 ms per step at one rank (DSKD_BENCH_WRAP_DDP=1 brings it back for A/B)
            if os.environ.get("DSKD_BENCH_WRAP_DDP"):
                from dskd_amd.dist import wrap_ddp
                wrapped = wrap_ddp(model, device_ids=[local_rank])
                mode = "eager+DistributedDataParallel"
            else:
                from dskd_amd.dist import GradSync
                gsync = GradSync(model, force=ddp1)
            if ddp1:
                mode += "(1-rank rccl)"
        optimizer = build_optimizer(model, cfg.optimizer[0])
        ahead = None if args.no_teacher_ahead else model.teacher_ahead()
        if ahead is not None:
            ahead.use_graphs = not args.no_teacher_graph
            mode += "+teacher_ahead" + ("(hipgraph)" if ahead.use_graphs else "")
        for _ in range(args.warmup):     # the first step runs its teacher inline, then the pipeline is primed
            loss, lv = train_step(model, wrapped, optimizer, data, synth, amp_dtype, ahead=ahead, gsync=gsync)
        if ahead is not None and ahead.use_graphs and not any(ahead._graphs.values()):
            mode = mode.replace("(hipgraph)", "(hipgraph pending)" if not ahead._graphs else "(hipgraph rejected)")

    def one_step():
        if stepper is not None:
            return stepper.step(data, inject)
        return train_step(model, wrapped, optimizer, data, synth, amp_dtype, ahead=ahead, gsync=gsync)[0]

    step_events = [] if os.environ.get("DSKD_BENCH_STEPTIMES") else None    # diagnostic: per-step GPU/host times
    host_marks = []
    ms0 = torch.cuda.memory_stats(device) if step_events is not None else None
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
        if step_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append(ev)
            host_marks.append(time.perf_counter() - t0)
    sync()
    dt = time.perf_counter() - t0
    if gsync is not None and rank == 0:
        print(f"[bench] GradSync: {len(gsync.buckets)} buckets, {gsync.stats}", file=sys.stderr)
    if step_events:
        ms1 = torch.cuda.memory_stats(device)
        print("[bench] allocator over the timed region: " + ", ".join(
            f"{k}={ms1.get(k, 0) - ms0.get(k, 0)}" for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
                                                             "num_ooms")) +
              f", reserved={ms1['reserved_bytes.all.current'] / 2**30:.1f} GiB, "
              f"active={ms1['active_bytes.all.current'] / 2**30:.1f} GiB", file=sys.stderr)
        gpu = [step_events[i - 1].elapsed_time(step_events[i]) for i in range(1, len(step_events))]
        host = [1e3 * (host_marks[i] - host_marks[i - 1]) for i in range(1, len(host_marks))]
        print("[bench] per-step ms, main stream: " + " ".join(f"{g:.1f}" for g in gpu), file=sys.stderr)
        print("[bench] per-step ms, host enqueue: " + " ".join(f"{h:.1f}" for h in host), file=sys.stderr, flush=True)

    # Roofline probe: the same step, same inputs, launched eagerly so that every MSDeformAttn
    # launch can be bracketed by HIP events on the launch stream (events cannot be recorded
    # inside a replayed hipGraph on ROCm).  Not part of the timed region.
    native.timing_enable(True)
    if hasattr(model.bbox_head, "graph_head"):
        model.bbox_head.graph_head = False      # launches inside a replayed hipGraph cannot be bracketed by events
    if ahead is not None:
        ahead.use_graphs = False                # the teacher's 12 fused MSDA forwards too: probe them eagerly
    for _ in range(args.probe_steps):
        if stepper is not None:
            stepper.eager_step(data, inject)
        else:
            train_step(model, wrapped, optimizer, data, synth, amp_dtype, ahead=ahead)
    torch.cuda.synchronize()
    kt = native.timing_collect()
    native.timing_enable(False)
    mfma = None
    if world == 1 and not args.no_mfma_probe and stepper is None:      # N=1 only: extra steps on one rank would hang DDP
        try:
            mfma = mfma_utilisation(lambda: train_step(model, wrapped, optimizer, data, synth, amp_dtype, ahead=ahead),
                                    args.dtype)
        except Exception as e:  # noqa: BLE001  (a diagnostic, never a reason to lose the bench line)
            print(f"[bench] mfma probe failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)

    tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
    nranks = torch.ones(1, device=cdev)            # counted over the same backend the gradients travel on
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nranks)
    dt = float(tmax.item())
    final_loss = float(loss.detach().float().item())
    if getattr(model.bbox_head, "last_lsap_status", None) is not None:
        native.raise_for_lsap_status(model.bbox_head.last_lsap_status)

    if rank == 0:
        esz = 2 if args.dtype == "bf16" else 4
        kernels = {}
        for tag, (n, ms) in kt.items():
            kind = "fwd_fused" if tag.endswith("_fused") else ("fwd" if "fwd" in tag else "bwd")
            nq = NV if "enc" in tag else 300
            byts = msda_algorithmic_bytes(kind, args.batch, nq, esz)
            avg_ms = ms / max(n, 1)
            kernels[tag] = {"launches": n, "avg_us": round(avg_ms * 1e3, 2), "algorithmic_MB": round(byts / 1e6, 2),
                            "achieved_GBs": round(byts / (avg_ms * 1e-3) / 1e9, 1), "total_ms": round(ms, 2)}
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
        roofline = None
        if dom:
            a = kernels[dom]["achieved_GBs"]
            # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this
            # process; the number comes from the committed rocprofv3 --pmc passes over the same
            # launch shape (profiles/r01_msda_pmc_hbm_B4_bf16.json: FETCH_SIZE doubled per the
            # gfx950 correction + WRITE_SIZE), valid for the default B=4 bf16 workload only.
            traffic = traffic_detail = None
            pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_msda_pmc_hbm_B4_bf16.json") for r in (4, 3, 1))
                        if os.path.exists(q)), "")          # the newest committed PMC pass (tools/prof/msda_pmc.sh)
            if args.batch == 4 and args.dtype == "bf16" and os.path.exists(pmc):
                with open(pmc) as f:
                    t_mb = json.load(f).get("traffic_corrected_MB", {}).get(dom)
                if t_mb:      # same unit as `achieved`: PMC bytes of one launch / measured launch time
                    traffic = round(t_mb / 1e3 / (kernels[dom]["avg_us"] * 1e-6), 1)
                    traffic_detail = {"MB_per_launch": t_mb, "algorithmic_MB_per_launch": kernels[dom]["algorithmic_MB"],
                                      "source": f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc, separate passes)"}
            roofline = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
                        "timing": f"HIP events around each launch, {args.probe_steps} eager steps of the same "
                                  "workload right after the timed region", "kernels": kernels}
        ips = args.batch * world * args.steps / dt
        out = {"metric": "images/sec (teacher+student distill step), %s COCO 800x1333" %
                         {"r50": "DefDETR-R50", "swin_t": "DefDETR-SwinT", "gfl_r50": "GFL-R50"}[args.backbone],
               "value": round(ips, 3),
               "unit": "images/sec", "n_gpus": world,
               ("gloo_ranks_shared_gpu" if rehearse else "rccl_ranks"): int(nranks.item()), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic (N(0,1) images 800x1333, 7 GT + 10 injected teacher detections per image, "
                       "random-init weights, teacher = perturbed copy)",
               "config": {"workload": {"r50": "Deformable-DETR R50 70+10 incremental DSKD distillation step "
                                              "(BASELINE.json configs[1])",
                                       "swin_t": "Deformable-DETR Swin-T 40+40 incremental DSKD distillation step "
                                                 "(BASELINE.json configs[3])",
                                       "gfl_r50": "GFL R50-FPN 40+40 incremental step with the DSKD feature-map term "
                                                  "(BASELINE.json configs[4])"}[args.backbone], "global_batch": args.batch * world,
                          "per_gpu_batch": args.batch, "image": [IMG_H, IMG_W], "queries": 300, "prev_classes": cfg.num_prev,
                          "parallelism": f"dp{world}", "execution": mode, "extra_untimed_warmup": extra_warmup,
                          "final_loss": round(final_loss, 4)},
               "roofline": roofline, "mfma": mfma}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.seed, cfg.num_prev, batch=args.cpu_baseline_batch)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1 or ddp1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
