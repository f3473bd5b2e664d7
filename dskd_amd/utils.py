"""Small host-side helpers."""
from collections import OrderedDict

import torch

_CONST_CACHE = OrderedDict()
_CONST_CACHE_MAX = 512


def device_const(values, dtype, device):
    """A small constant tensor built from python numbers, cached per (values, dtype, device).
    Repeated steps reuse the device copy instead of issuing a host->device transfer (which is
    also what makes a step capturable into a hipGraph: no H2D inside the captured region).
    Never modify the returned tensor in place."""
    def freeze(v):
        return tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else v
    key = (freeze(values), dtype, str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.tensor(values, dtype=dtype, device=device)
        _CONST_CACHE[key] = t
        if len(_CONST_CACHE) > _CONST_CACHE_MAX:
            _CONST_CACHE.popitem(last=False)
    else:
        _CONST_CACHE.move_to_end(key)
    return t


def const_cache_snapshot():
    """References to every cached constant: a captured hipGraph holds raw pointers to them, so
    the graph's owner keeps this list alive (the LRU may otherwise evict and free them)."""
    return list(_CONST_CACHE.values())


class GraphedFunction:
    """A pure tensor function ``fn(*tensors) -> tuple of tensors`` replayed as two hipGraphs
    (forward, and backward = ``autograd.grad`` of its outputs), wired into autograd.

    For regions made of hundreds of tiny launches with fixed shapes (the dense detection losses:
    378 kernels, 2.5 ms of GPU time, ~5 ms of host time per step).  Only valid for regions WITHOUT
    memset nodes on this ROCm runtime (graph_step.py explains why; check with the profiler, as
    scratch/loss_memset_check.py does).  ``capture_error_mode='thread_local'``: other threads of
    the process (RCCL watchdog) keep making HIP calls during capture.

    Inputs are copied into static buffers by one multi-tensor launch; outputs and input gradients
    alias static buffers that the next call overwrites (consume them within the step)."""

    def __init__(self, fn, sample_args):
        self.args_meta = [(tuple(a.shape), a.dtype, a.requires_grad) for a in sample_args]
        self.static_in = [a.detach().clone().requires_grad_(a.requires_grad) for a in sample_args]
        self.grad_idx = [i for i, a in enumerate(self.static_in) if a.requires_grad]
        dev = sample_args[0].device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                       # warm-up outside capture (lazy inits, caches)
            for _ in range(2):
                outs = fn(*self.static_in)
                torch.autograd.grad(outs, [self.static_in[i] for i in self.grad_idx],
                                    [torch.ones_like(o) for o in outs], allow_unused=True)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, stream=side, capture_error_mode="thread_local"):
            self.static_out = tuple(fn(*self.static_in))
        self.static_go = [torch.zeros_like(o) for o in self.static_out]
        self.bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd, pool=self.fwd.pool(), stream=side, capture_error_mode="thread_local"):
            self.static_gi = torch.autograd.grad(self.static_out, [self.static_in[i] for i in self.grad_idx],
                                                 self.static_go, allow_unused=True)
        torch.cuda.synchronize(dev)
        outer = self

        class _Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *args):
                torch._foreach_copy_([t.detach() for t in outer.static_in], [a.detach() for a in args])
                outer.fwd.replay()
                return tuple(o.detach() for o in outer.static_out)

            @staticmethod
            def backward(ctx, *gos):
                torch._foreach_copy_(outer.static_go, [g if g is not None else torch.zeros_like(s)
                                                       for g, s in zip(gos, outer.static_go)])
                outer.bwd.replay()
                grads = [None] * len(outer.static_in)
                for i, g in zip(outer.grad_idx, outer.static_gi):
                    grads[i] = g
                return tuple(grads)

        self._fn = _Fn

    def matches(self, args):
        return len(args) == len(self.args_meta) and all(
            (tuple(a.shape), a.dtype, a.requires_grad) == m for a, m in zip(args, self.args_meta))

    def __call__(self, *args):
        return self._fn.apply(*args)


def deepcopy_without(obj, memo, skip):
    """``copy.deepcopy`` of a module minus run-time accelerator state (hipGraphs, streams): the
    incremental driver deep-copies the trained student to make the next task's teacher."""
    import copy
    new = obj.__class__.__new__(obj.__class__)
    memo[id(obj)] = new
    for k, v in obj.__dict__.items():
        if k not in skip:
            new.__dict__[k] = copy.deepcopy(v, memo)
    return new
