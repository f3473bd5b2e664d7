"""AdamW with the global-norm gradient clip folded in: two launches per step for the whole detector
(``csrc/optim.hip``, ``dskd_clip_adamw``) instead of PyTorch's ~40 multi-tensor launches.

Stands in for what the reference's config asks of ext-mmcv (``optimizer = dict(type='AdamW', ...)`` with
``paramwise_cfg.custom_keys`` lr multipliers and ``optimizer_config = dict(grad_clip=dict(max_norm=0.1, norm_type=2))``,
/root/reference/configs/deformable_detr/chaosuan_gfl_deformable_detr_70_r50_8x4_1x_qoqo_il.py:213-224): same update rule and
state layout as ``torch.optim.AdamW`` (``state[p] = {step, exp_avg, exp_avg_sq}``), so its checkpoints load either way."""
import ctypes as C
import math

import torch

from . import native


class FusedClipAdamW(torch.optim.Optimizer):
    """``clip_and_step(max_norm)`` = ``clip_grad_norm_(params, max_norm)`` + ``AdamW.step()`` in two launches; ``step()``
    alone is the plain update.  CUDA f32 parameters only (``runner.build_optimizer`` falls back to torch otherwise).
    The gradients are left unscaled (the clipped gradient exists only inside the update kernel)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 8:
            raise ValueError("FusedClipAdamW: at most 8 parameter groups")
        if len({(g["betas"], g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("FusedClipAdamW: betas / eps must be the same in every group")
        self._tables = None
        self._grad_ptrs = None
        self.last_norm = None          # device f32 [2]: total gradient norm, clip coefficient of the last step

    def _params(self):
        return [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"]]

    def _build_tables(self, live):
        dev = live[0][1].device
        chunk = int(native.load().dskd_clip_adamw_chunk())
        ptrs, meta, chunks = [[], [], [], []], [], []
        for ti, (gi, p) in enumerate(live):
            st = self.state[p]
            ptrs[0].append(p.data_ptr())
            ptrs[2].append(st["exp_avg"].data_ptr())
            ptrs[3].append(st["exp_avg_sq"].data_ptr())
            ptrs[1].append(0)
            meta += [p.numel(), gi]
            for s in range(0, p.numel(), chunk):
                chunks += [ti, s]
        n_t, n_c = len(live), len(chunks) // 2
        self._tables = dict(
            key=tuple((id(p), p.data_ptr()) for _, p in live), n_t=n_t, n_c=n_c,
            ptrs=torch.tensor(ptrs, dtype=torch.int64).to(dev), meta=torch.tensor(meta, dtype=torch.int32, device=dev),
            chunks=torch.tensor(chunks, dtype=torch.int32, device=dev),
            partials=torch.empty(max(n_c, 1), dtype=torch.float32, device=dev),
            norm=torch.zeros(2, dtype=torch.float32, device=dev))
        self._grad_ptrs = None

    @staticmethod
    def _dense(t):
        """Element order must not matter and every element must be addressable linearly: any dense layout will do
        (channels_last conv weights are dense but not 'contiguous') as long as all four tensors share it."""
        return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)

    @torch.no_grad()
    def clip_and_step(self, max_norm=None):
        live = [(gi, p) for gi, p in self._params() if p.grad is not None]
        if not live:
            return None
        for gi, p in live:
            st = self.state[p]
            if not st:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and self._dense(p)
                    and g.stride() == p.stride() and st["exp_avg"].stride() == p.stride()
                    and st["exp_avg_sq"].stride() == p.stride() and not g.is_sparse):
                raise native.NativeError("FusedClipAdamW: dense f32 CUDA parameters / gradients with equal strides expected "
                                         f"(got {tuple(p.shape)} {p.dtype} strides {p.stride()} vs grad {g.stride()})")
        key = tuple((id(p), p.data_ptr()) for _, p in live)
        if self._tables is None or self._tables["key"] != key:
            self._build_tables(live)
        tb = self._tables
        gp = [p.grad.data_ptr() for _, p in live]
        if gp != self._grad_ptrs:          # fresh gradient tensors usually come back at the same addresses: copy on change
            # a NEW pinned source per copy: the host runs steps ahead of the GPU, so a reused staging buffer would be
            # overwritten before an earlier step's copy has executed (the caching host allocator keeps this one alive
            # until its copy is done)
            src = torch.tensor(gp, dtype=torch.int64).pin_memory()
            tb["ptrs"][1].copy_(src, non_blocking=True)
            self._grad_ptrs = gp
        # validate BEFORE touching any state: a raise must leave the optimizer as it was
        steps = {int(self.state[p]["step"]) + 1 for _, p in live}
        if len(steps) != 1:
            raise native.NativeError("FusedClipAdamW: parameters with different step counts %s (a parameter that got its first "
                                     "gradient later than the others, or partially loaded state): the kernel applies ONE "
                                     "bias correction per launch -- build the optimizer with "
                                     "runner.build_optimizer(..., fused_clip=False) (torch.optim.AdamW, per-parameter "
                                     "steps) for such a schedule" % sorted(steps))
        for _, p in live:
            st = self.state[p]
            st["step"] = st["step"] + 1            # a host scalar (tensor or int), as in torch.optim.AdamW's state
        g0 = self.param_groups[0]
        ng = len(self.param_groups)
        lr = (C.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (C.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        dev = live[0][1].device
        rc = native.load().dskd_clip_adamw(
            tb["ptrs"].data_ptr(), tb["meta"].data_ptr(), tb["chunks"].data_ptr(), tb["partials"].data_ptr(),
            tb["norm"].data_ptr(), tb["n_t"], tb["n_c"], lr, wd, ng, float(g0["betas"][0]), float(g0["betas"][1]),
            float(g0["eps"]), steps.pop(), float(max_norm) if max_norm else 0.0, torch.cuda.current_stream(dev).cuda_stream)
        native._check(rc, "dskd_clip_adamw")
        self.last_norm = tb["norm"]
        return tb["norm"][0].clone()       # the buffer is rewritten by the next step: a logger may keep what it gets

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(None)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = None            # moments were replaced
        for _, p in self._params():    # a checkpoint written from a differently laid-out model (contiguous vs channels_last)
            st = self.state.get(p)
            for k in ("exp_avg", "exp_avg_sq"):
                if st and k in st and st[k].stride() != p.stride():
                    m = torch.empty_like(p, memory_format=torch.preserve_format)
                    m.copy_(st[k])
                    st[k] = m

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)


def math_reference_step(p, g, m, v, lr, wd, beta1, beta2, eps, step, coef=1.0):
    """The update rule of ``clip_adamw_kernel`` in float64 numpy-free Python (documentation / tests)."""
    g = g * coef
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - beta1)
    v = beta2 * v + (1 - beta2) * g * g
    denom = math.sqrt(v) / math.sqrt(1 - beta2 ** step) + eps
    return p - lr / (1 - beta1 ** step) * m / denom, m, v
