#!/usr/bin/env python
"""Incremental-task training driver: the counterpart of
/root/reference/tools/train_increment.py (main :103-375) on top of ``dskd_amd``.

Same flow: load the config (reference config files load unchanged), optional process
group, then per task: build / reuse the student, teacher := frozen deep copy of the previous
student (:250-251), dataset + ``set_datainfo`` (:268-272), DDP wrap of the student only
(:301-303), optimizer / lr / grad-clip / runner from the per-task config lists, run.
Data is the synthetic IL dataset (the reference's dataset class is missing).

  python tools/train_increment.py --config=CONFIG --work-dir=DIR [--resume-from=CKPT] [--auto-resume] \
      [--launcher=pytorch] [--cfg-options k=v ...] [--device cuda|cpu] [--amp bf16] [--max-iters N]

The command line is the reference's (:33-101), so /root/reference/tools/dist_train_increment.sh:22-28 drives
it unchanged; CONFIG may also be given positionally.  ``--resume-from`` / ``cfg.task.resume_by_epoch``
restore student + optimizer + epoch counters into the runner of the first task that runs (:355-361);
``cfg.task.resume_by_task`` skips the earlier tasks the way the reference does (:211-237).
"""
import argparse
import ast
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("MIOPEN_FIND_MODE", "1")   # measured conv algorithm choice in the first process too (see bench.py)
import torch  # noqa: E402

import dskd_amd  # noqa: E402,F401
from dskd_amd.builder import build_detector  # noqa: E402
from dskd_amd.config import Config  # noqa: E402
from dskd_amd.datasets import build_dataloader, build_dataset  # noqa: E402
from dskd_amd.dist import GradSync, get_dist_info, init_dist, wrap_ddp  # noqa: E402
from dskd_amd.runner import TaskEpochBasedRunner, build_optimizer  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train a detector incrementally (DSKD)")
    p.add_argument("config_pos", nargs="?", default=None, metavar="CONFIG", help="alias of --config")
    p.add_argument("--config", default=None, help="train config file path")
    p.add_argument("--work-dir")
    p.add_argument("--resume-from", default="", help="the checkpoint file to resume from")
    p.add_argument("--auto-resume", action="store_true", help="resume from the latest checkpoint in the work dir")
    p.add_argument("--print-model", action="store_true")
    p.add_argument("--gpu-id", type=int, default=0, help="accepted for compatibility (one process drives one GPU)")
    p.add_argument("--find_unused_param", action="store_true",
                   help="accepted for compatibility: the one parameter unused by construction is excluded from DDP")
    p.add_argument("--seed", type=int, default=111)
    p.add_argument("--diff-seed", action="store_true")
    p.add_argument("--deterministic", action="store_true")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="none")
    p.add_argument("--local_rank", "--local-rank", type=int, default=0)
    p.add_argument("--options", nargs="+", default=[], help="deprecated alias of --cfg-options")
    p.add_argument("--cfg-options", nargs="+", default=[])
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--amp", choices=["none", "bf16"], default="none")
    p.add_argument("--max-iters", type=int, default=None, help="iterations per epoch (smoke runs)")
    p.add_argument("--max-epochs", type=int, default=None)
    args = p.parse_args(argv)
    if args.config is not None and args.config_pos is not None:
        p.error("give the config once: positionally or with --config")
    args.config = args.config or args.config_pos
    if args.config is None:
        p.error("a config file is required (--config=FILE)")
    if args.options and args.cfg_options:
        raise ValueError("--options and --cfg-options cannot be both specified, --options is deprecated in favor "
                         "of --cfg-options")
    if args.options:
        args.cfg_options = args.options
    if args.launcher in ("slurm", "mpi"):
        p.error(f"--launcher={args.launcher}: only 'none' and 'pytorch' (one process per GPU) are implemented")
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(args.local_rank)
    return args


def find_latest_checkpoint(work_dir):
    """Newest ``task_{t}_epoch_{e}.pth`` of a work dir by (task, epoch), or None."""
    import re
    best = None
    if work_dir and os.path.isdir(work_dir):
        for f in os.listdir(work_dir):
            m = re.fullmatch(r"task_(\d+)_epoch_(\d+)\.pth", f)
            if m:
                key = (int(m.group(1)), int(m.group(2)))
                if best is None or key > best[0]:
                    best = (key, os.path.join(work_dir, f))
    return best[1] if best else None


def _parse_opts(pairs):
    out = {}
    for kv in pairs:
        k, v = kv.split("=", 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


def per_task(value, tid):
    return value[tid - 1] if isinstance(value, list) else value


def main(argv=None, cpu_checker=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    cfg.merge_from_dict(_parse_opts(args.cfg_options))
    distributed = args.launcher != "none"
    if distributed:
        init_dist(args.launcher, backend="nccl" if args.device == "cuda" else "gloo")
    rank, world = get_dist_info()
    seed = args.seed + (rank if args.diff_seed else 0)
    torch.manual_seed(seed)
    device = torch.device(args.device, int(os.environ.get("LOCAL_RANK", 0)) if args.device == "cuda" else None) \
        if args.device == "cuda" else torch.device("cpu")
    work_dir = args.work_dir or cfg.get("work_dir") or os.path.join("work_dirs", os.path.splitext(os.path.basename(args.config))[0])
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)

    task_nums = len(cfg.data.train.catsplit)
    assert cfg.data.get("cat_split_load", "auto") == "auto", "only continuous task training is implemented"
    # resume controls of the reference driver (:140-142, :211-237, :355-361)
    task_cfg = cfg.get("task") or {}
    resume_by_task = int(task_cfg.get("resume_by_task") or 0)
    resume_from = args.resume_from or task_cfg.get("resume_by_epoch") or cfg.get("resume_from") or ""
    if not resume_from and args.auto_resume:
        resume_from = find_latest_checkpoint(work_dir) or ""
    if resume_from and not os.path.isfile(resume_from):
        raise FileNotFoundError(f"--resume-from: {resume_from} does not exist")

    def student_ckpt_of(tid):
        tcfg = task_cfg.get(f"Task{tid}", {}) or {}
        ck = tcfg.get("student_ckpt") if tcfg.get("load_student") else None
        return ck if ck and os.path.isfile(str(ck)) else None

    # "resume by task" means: the student of task `resume_by_task` comes from its checkpoint and training continues
    # with the next task.  Without that checkpoint on disk there is nothing to resume from: train every task.
    if resume_by_task and student_ckpt_of(resume_by_task) is None:
        log(f"task.resume_by_task={resume_by_task}: no student checkpoint on disk, training from task 1")
        resume_by_task = 0
    model, runners = None, []
    for tid in range(1, task_nums + 1):
        if tid < resume_by_task:
            log(f"======== Task-{tid} skipped (resume_by_task={resume_by_task}) ========")
            continue
        log(f"======== Task-{tid} start ========")
        if tid == resume_by_task:
            model = build_detector(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
            model.init_weights()
            model.set_student(ckptfile=student_ckpt_of(tid))
            model.set_teacher(config=None, ckptfile=None, model=None, trainval="val")
            log(f"======== Task-{tid} skipped: student resumed from {student_ckpt_of(tid)} ========")
            continue
        if tid == 1:
            cfg.model.backbone.init_cfg = cfg.model.backbone.get("init_cfg") if cfg.model.backbone.get("init_cfg") and \
                os.path.isfile(str(cfg.model.backbone.init_cfg.get("checkpoint", ""))) else None
            model = build_detector(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
            model.init_weights()
            tcfg = cfg.get("task", {}).get(f"Task{tid}", {}) if cfg.get("task") else {}
            ck = tcfg.get("student_ckpt") if tcfg.get("load_student") else None
            if ck and os.path.isfile(ck):
                model.set_student(ckptfile=ck)
            model.set_teacher(config=None, ckptfile=None, model=None, trainval="val")
        else:
            model = model.module if hasattr(model, "module") else model
            model.set_teacher(model=copy.deepcopy(model), trainval="val")        # teacher := previous student
        catload = [1 if i == tid - 1 else 0 for i in range(task_nums)]
        ds_cfg = dict(cfg.data.train)
        ds_cfg.update(catload=catload)
        train_dataset = build_dataset(ds_cfg, dict(test_mode=False, seed=seed))
        loader = build_dataloader(train_dataset, cfg.data.samples_per_gpu, cfg.data.get("workers_per_gpu", 0),
                                  dist=distributed, seed=seed)
        model.set_datainfo(cat2id=train_dataset.ALL_CLASSES_IDS, cat2label=train_dataset.cat2label,
                           pred_cat=train_dataset.PRED_CLASSES, load_cat=train_dataset.LOAD_CLASSES,
                           task_cat=train_dataset.TASK_CLASSES)
        model.to(device)
        # data parallel (reference: MMDistributedDataParallel, tools/train_increment.py:301-303): gradient buckets + hooks
        # (dist.GradSync; same averaged gradients, none of DDP's per-parameter bucket copies); DSKD_WRAP_DDP=1: the wrapper
        wrapped, grad_sync = model, None
        if distributed and os.environ.get("DSKD_WRAP_DDP"):
            wrapped = wrap_ddp(model, device_ids=[device.index] if device.type == "cuda" else None)
        elif distributed:
            grad_sync = GradSync(model)
        optimizer = build_optimizer(wrapped, per_task(cfg.optimizer, tid))
        rcfg = dict(per_task(cfg.runner, tid))
        rcfg.pop("type", None)
        if args.max_epochs is not None:
            rcfg["max_epochs"] = args.max_epochs
        runner = TaskEpochBasedRunner(wrapped, optimizer, work_dir=work_dir, logger=log,
                                      grad_clip=(cfg.get("optimizer_config") or {}).get("grad_clip"),
                                      lr_config=per_task(cfg.lr_config, tid),
                                      log_interval=cfg.get("log_config", {}).get("interval", 50),
                                      checkpoint_interval=cfg.get("checkpoint_config", {}).get("interval", 1),
                                      amp_dtype=torch.bfloat16 if args.amp == "bf16" else None,
                                      max_iters_per_epoch=args.max_iters, grad_sync=grad_sync, **rcfg)
        if args.print_model and rank == 0 and not runners:
            log(model)
        if resume_from:                # student + optimizer + epoch / iteration counters of an interrupted task
            meta = runner.resume(resume_from, map_location="cpu")
            log(f"resumed from {resume_from}: {meta}")
            resume_from = ""
        tic = time.time()
        runner.run([loader], cfg.get("workflow", [("train", 1)]), cur_task=tid)
        if grad_sync is not None:
            grad_sync.remove()         # the next task builds its own (its hooks would otherwise pile up on shared parameters)
        log(f"======== Task-{tid} done in {time.time() - tic:.1f}s ========")
        runners.append(runner)
    return runners


if __name__ == "__main__":
    main()
