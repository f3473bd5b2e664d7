#!/usr/bin/env python
"""Incremental-task training driver: the counterpart of
/root/reference/tools/train_increment.py (main :103-375) on top of ``dskd_amd``.

Same flow: load the config (reference config files load unchanged), optional process
group, then per task: build / reuse the student, teacher := frozen deep copy of the previous
student (:250-251), dataset + ``set_datainfo`` (:268-272), DDP wrap of the student only
(:301-303), optimizer / lr / grad-clip / runner from the per-task config lists, run.
Data is the synthetic IL dataset (the reference's dataset class is missing).

  python tools/train_increment.py CONFIG --work-dir DIR [--launcher pytorch] \
      [--cfg-options k=v ...] [--device cuda|cpu] [--amp bf16] [--max-iters N]
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
from dskd_amd.dist import get_dist_info, init_dist, wrap_ddp  # noqa: E402
from dskd_amd.runner import TaskEpochBasedRunner, build_optimizer  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train a detector incrementally (DSKD)")
    p.add_argument("config")
    p.add_argument("--work-dir")
    p.add_argument("--seed", type=int, default=111)
    p.add_argument("--diff-seed", action="store_true")
    p.add_argument("--deterministic", action="store_true")
    p.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    p.add_argument("--local_rank", "--local-rank", type=int, default=0)
    p.add_argument("--cfg-options", nargs="+", default=[])
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--amp", choices=["none", "bf16"], default="none")
    p.add_argument("--max-iters", type=int, default=None, help="iterations per epoch (smoke runs)")
    p.add_argument("--max-epochs", type=int, default=None)
    return p.parse_args(argv)


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
    model, runners = None, []
    for tid in range(1, task_nums + 1):
        log(f"======== Task-{tid} start ========")
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
        wrapped = wrap_ddp(model, device_ids=[device.index] if device.type == "cuda" else None) if distributed else model
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
                                      max_iters_per_epoch=args.max_iters, **rcfg)
        tic = time.time()
        runner.run([loader], cfg.get("workflow", [("train", 1)]), cur_task=tid)
        log(f"======== Task-{tid} done in {time.time() - tic:.1f}s ========")
        runners.append(runner)
    return runners


if __name__ == "__main__":
    main()
