"""Task/epoch runner with the hooks the reference's driver installs
(/root/reference/tools/train_increment.py:314-364): optimizer construction with
``paramwise_cfg.custom_keys`` lr multipliers (config :213-224), step LR with linear warm-up
(config :228-235), the optimizer hook (zero_grad, backward, clip_grad_norm_, step; config
:224), text logging every N iterations, the finite-loss check of
/root/reference/mmdet/core/hook/checkloss_hook.py:7-24, and per-epoch checkpoints named
``task_{t}_epoch_{e}.pth`` holding the STUDENT only (``save_teacher=False``, config :237).
``TaskEpochBasedRunner`` itself lives in the missing ``mmcvil`` package; this is its
counterpart, not a copy."""
import math
import os
import time

import torch

from .datasets import to_device
from .dist import get_dist_info


def build_optimizer(model, cfg, capturable=False, fused_clip=True):
    """ext-mmcv DefaultOptimizerConstructor semantics for the keys used by the DSKD configs:
    a parameter whose name contains a custom key gets lr * lr_mult (longest key wins)."""
    cfg = dict(cfg)
    typ = cfg.pop("type")
    paramwise = dict(cfg.pop("paramwise_cfg", None) or {})
    custom = paramwise.get("custom_keys", {})
    base_lr, base_wd = cfg["lr"], cfg.get("weight_decay", 0.0)
    keys = sorted(custom.keys(), key=len, reverse=True)
    module = model.module if hasattr(model, "module") else model
    groups = []
    for name, p in module.named_parameters():
        if not p.requires_grad:
            continue
        g = {"params": [p], "lr": base_lr, "weight_decay": base_wd}
        for k in keys:
            if k in name:
                g["lr"] = base_lr * custom[k].get("lr_mult", 1.0)
                g["weight_decay"] = base_wd * custom[k].get("decay_mult", 1.0)
                break
        groups.append(g)
    # merge groups with identical hyper-parameters (fewer, larger fused updates)
    merged = {}
    for g in groups:
        merged.setdefault((g["lr"], g["weight_decay"]), []).extend(g["params"])
    groups = [{"params": ps, "lr": lr, "weight_decay": wd} for (lr, wd), ps in merged.items()]
    kwargs = {k: v for k, v in cfg.items() if k not in ("lr", "weight_decay")}
    if typ == "AdamW" and not capturable and fused_clip and len(groups) <= 8 and not kwargs.get("amsgrad") and \
            all(p.is_cuda and p.dtype == torch.float32 for g in groups for p in g["params"]):
        # clip + update of all tensors in two launches (csrc/optim.hip); the runner calls clip_and_step(max_norm)
        from .optim import FusedClipAdamW
        return FusedClipAdamW(groups, lr=base_lr, weight_decay=base_wd,
                              **{k: v for k, v in kwargs.items() if k in ("betas", "eps")})
    opt_cls = getattr(torch.optim, typ)
    if typ in ("AdamW", "Adam") and all(p.is_cuda for g in groups for p in g["params"]):
        kwargs.setdefault("fused", True)     # one multi-tensor kernel per group on the GPU
        if capturable:
            kwargs.setdefault("capturable", True)   # step counter on the device: hipGraph-safe
    return opt_cls(groups, lr=base_lr, weight_decay=base_wd, **kwargs)


class StepLrWarmup:
    """policy='step' with warmup='linear' (by iteration), as ext-mmcv's StepLrUpdaterHook."""

    def __init__(self, optimizer, step, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1, **kwargs):
        self.opt, self.step, self.gamma = optimizer, list(step) if isinstance(step, (list, tuple)) else [step], gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.base = [g["lr"] for g in optimizer.param_groups]

    def regular(self, epoch):
        k = sum(1 for s in self.step if epoch >= s)
        return [b * self.gamma ** k for b in self.base]

    def set(self, epoch, it):
        lrs = self.regular(epoch)
        if self.warmup == "linear" and it < self.warmup_iters:
            k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
            lrs = [lr * (1 - k) for lr in lrs]
        for g, lr in zip(self.opt.param_groups, lrs):
            g["lr"] = lr


class TaskEpochBasedRunner:
    def __init__(self, model, optimizer, work_dir=None, logger=print, max_epochs=12, max_tasks=1, save_teacher=False,
                 grad_clip=None, lr_config=None, log_interval=50, checkpoint_interval=1, amp_dtype=None,
                 max_iters_per_epoch=None, grad_sync=None, **kwargs):
        self.model, self.optimizer, self.work_dir, self.log = model, optimizer, work_dir, logger
        self.grad_sync = grad_sync       # dist.GradSync of a data-parallel run (None: one process, or a DDP-wrapped model)
        self.max_epochs, self.max_tasks, self.save_teacher = max_epochs, max_tasks, save_teacher
        self.grad_clip = dict(grad_clip) if grad_clip else None
        lr_config = dict(lr_config or dict(policy="step", step=[max_epochs + 1]))
        assert lr_config.pop("policy", "step") == "step"
        self.lr = StepLrWarmup(optimizer, **lr_config)
        self.log_interval, self.checkpoint_interval = log_interval, checkpoint_interval
        self.amp_dtype, self.max_iters_per_epoch = amp_dtype, max_iters_per_epoch
        self.epoch = self.iter = 0
        self.history = []

    @property
    def module(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def train_iter(self, data, next_data=None):
        """One optimisation step = what ext-mmcv's runner + OptimizerHook do per batch.
        ``next_data`` (already on the device): its frozen-teacher forward is enqueued on a second
        stream behind this batch's student backward (``TeacherAhead``), so the teacher's decode
        never drains the main stream."""
        dev = next(self.module.parameters()).device
        data = to_device(data, dev)
        self.lr.set(self.epoch, self.iter)
        self.optimizer.zero_grad(set_to_none=True)
        ahead = self.module.teacher_ahead() if (dev.type == "cuda" and getattr(self.module, "has_teacher", False)
                                                and hasattr(self.module, "teacher_ahead")) else None
        with torch.autocast(device_type=dev.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            if ahead is not None:
                data = dict(data, teacher_info=ahead.finish(data["img"], data["img_metas"]))
            if hasattr(self.model, "module"):       # DDP: go through forward so gradient hooks fire
                losses = self.model(**data)
                loss, log_vars = self.module._parse_losses(losses)
                out = dict(loss=loss, log_vars=log_vars, num_samples=len(data["img_metas"]))
            else:
                out = self.model.train_step(data, self.optimizer)
            if ahead is not None and next_data is not None:
                ahead.launch(next_data["img"], next_data["img_metas"], amp_dtype=self.amp_dtype)
        out["loss"].backward()
        if self.grad_sync is not None:          # data parallel without the DDP wrapper (dist.GradSync)
            self.grad_sync.finish()
        if self.grad_clip and hasattr(self.optimizer, "clip_and_step") and self.grad_clip.get("norm_type", 2) == 2:
            out["grad_norm"] = self.optimizer.clip_and_step(self.grad_clip["max_norm"])      # two launches for both
        else:
            if self.grad_clip:
                params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
                out["grad_norm"] = torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)
            self.optimizer.step()
        self.iter += 1
        return out

    def run(self, data_loaders, workflow=(("train", 1),), cur_task=1, **kwargs):
        loader = data_loaders[0]
        rank, _ = get_dist_info()
        self.module.train()
        for epoch in range(self.epoch, self.max_epochs):
            self.epoch = epoch
            if hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(epoch)
            tic = time.time()
            dev = next(self.module.parameters()).device
            batches = iter(loader)
            nxt = next(batches, None)
            i = -1
            while nxt is not None:
                i += 1
                if self.max_iters_per_epoch is not None and i >= self.max_iters_per_epoch:
                    break
                data, nxt = to_device(nxt, dev), next(batches, None)
                if nxt is not None:
                    nxt = to_device(nxt, dev)
                out = self.train_iter(data, nxt)
                if (i + 1) % self.log_interval == 0 or i == 0:
                    lv = out["log_vars"]
                    if not math.isfinite(lv["loss"]):                   # CheckInvalidLossHook
                        raise FloatingPointError(f"loss become infinite or NaN at task {cur_task} iter {self.iter}")
                    self.history.append(dict(task=cur_task, epoch=epoch + 1, iter=i + 1, **lv))
                    if rank == 0:
                        self.log(f"Task [{cur_task}] Epoch [{epoch + 1}][{i + 1}/{len(loader)}] "
                                 f"lr: {self.optimizer.param_groups[0]['lr']:.3e}, "
                                 + ", ".join(f"{k}: {v:.4f}" for k, v in lv.items()))
            ta = self.module.__dict__.get("_teacher_ahead")
            if ta is not None:
                ta.discard()          # a batch launched before the loop ended belongs to no later iteration
            self.epoch = epoch + 1
            if rank == 0 and self.work_dir and (epoch + 1) % self.checkpoint_interval == 0:
                self.save_checkpoint(cur_task, epoch + 1)
            if rank == 0:
                self.log(f"Task [{cur_task}] epoch {epoch + 1} done in {time.time() - tic:.1f}s")

    def save_checkpoint(self, task, epoch):
        os.makedirs(self.work_dir, exist_ok=True)
        path = os.path.join(self.work_dir, f"task_{task}_epoch_{epoch}.pth")
        torch.save(dict(meta=dict(task=task, epoch=epoch, iter=self.iter), state_dict=self.module.state_dict(),
                        optimizer=self.optimizer.state_dict()), path)    # teacher is a plain attribute: not included
        return path

    def resume(self, path, map_location="cpu"):
        ck = torch.load(path, map_location=map_location)
        self.module.load_state_dict(ck["state_dict"])
        if "optimizer" in ck:
            self.optimizer.load_state_dict(ck["optimizer"])
        self.epoch, self.iter = ck["meta"]["epoch"], ck["meta"]["iter"]
        return ck["meta"]
