"""Small host-side helpers."""
from collections import OrderedDict

import contextlib
import gc

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



@contextlib.contextmanager
def no_gc_during_capture():
    """Python's cyclic garbage collector must not run inside a hipGraph capture: collecting an older, unreachable
    CUDAGraph (or anything else whose destructor makes a HIP call that is illegal while a stream is capturing) aborts
    the process -- seen once the test suite had grown enough garbage for a collection to land inside the capture.
    ``torch.cuda.graph`` collects BEFORE the capture begins; this keeps the collector off until it ends."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

class GraphedFunction:
    """A pure tensor function ``fn(*tensors) -> tuple of tensors`` replayed as two hipGraphs
    (forward, and backward = ``autograd.grad`` of its outputs), wired into autograd.

    For regions made of hundreds of launches with fixed shapes (the dense detection losses: 378 kernels,
    2.5 ms of GPU time, ~5 ms of host time per step; the student's transformer + heads: ~1 000 launches
    forward and backward).  Only valid for regions WITHOUT memset nodes on this ROCm runtime (graph_step.py
    explains why); ``verify=True`` replays forward and backward three times on the same inputs and rejects
    the capture unless every replay returns the same finite numbers (a memset node replays with a garbage
    fill value from the second replay on).  ``capture_error_mode='thread_local'``: other threads of the
    process (RCCL watchdog) keep making HIP calls during capture.

    ``fn`` is called as ``fn(*sample_args, *static_args)`` and must be a function of exactly these tensors (it
    may install them into modules for the duration of the call).  ``sample_args`` are copied into static buffers by
    one multi-tensor launch per call.  ``static_args`` are tensors whose STORAGE is the same on every call
    (parameters, persistent low-precision copies of them): they are used where they lie, and their gradients come
    back like those of the other inputs.  During warm-up and capture ``fn`` receives FRESH leaf aliases of them: the
    backward capture asks autograd for gradients w.r.t. its inputs, and the engine synchronises the capture stream
    with the stream of whatever node consumes such a gradient -- a parameter's AccumulateGrad node kept alive by an
    older graph (the previous step's loss still referenced by the caller), or the node that produced a non-leaf
    input on the main stream.  That pulls the default stream into the capture and crashes ``hipStreamEndCapture``;
    fresh leaves first used on the capture stream have no such consumer.  Outputs and input gradients alias static
    buffers that the next call overwrites (consume them within the step)."""

    def __init__(self, fn, sample_args, static_args=(), verify=False, autocast_dtype=None, against_eager=False,
                 arg_names=None):
        self.arg_names = arg_names
        self.keepalive = []      # set by the owner: cache-owned tensors the captured kernels read (see head._forward_graphed)
        self.args_meta = [(tuple(a.shape), a.dtype, a.requires_grad) for a in sample_args]
        self.static_meta = [(a.data_ptr(), tuple(a.shape), a.dtype) for a in static_args]
        self.static_in = [a.detach().clone().requires_grad_(a.requires_grad) for a in sample_args]
        statics = [a.detach().requires_grad_(a.requires_grad) for a in static_args]      # fresh leaves, same storage
        every = self.static_in + statics
        self.grad_idx = [i for i, a in enumerate(every) if a.requires_grad]
        dev = (sample_args[0] if sample_args else statics[0]).device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)

        def run():
            with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
                return tuple(fn(*every))
        with torch.cuda.stream(side):                       # warm-up outside capture (lazy inits, caches, workspaces)
            for _ in range(2):
                outs = run()
                torch.autograd.grad([o for o in outs if o.requires_grad], [every[i] for i in self.grad_idx],
                                    [torch.ones_like(o) for o in outs if o.requires_grad], allow_unused=True)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        mode = "thread_local"
        import os as _os
        trace = (lambda m: print("[graph]", m, flush=True)) if _os.environ.get("DSKD_GRAPH_TRACE") else (lambda m: None)
        trace("warm-up done")
        self.fwd = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(self.fwd, stream=side, capture_error_mode=mode):
            self.static_out = run()
        trace("forward captured")
        self.out_grad = [o.requires_grad for o in self.static_out]
        self.static_go = [torch.zeros_like(o) for o in self.static_out]
        self.bwd = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(self.bwd, pool=self.fwd.pool(), stream=side, capture_error_mode=mode):
            self.static_gi = torch.autograd.grad([o for o, r in zip(self.static_out, self.out_grad) if r],
                                                 [every[i] for i in self.grad_idx],
                                                 [g for g, r in zip(self.static_go, self.out_grad) if r], allow_unused=True)
        trace("backward captured")
        torch.cuda.synchronize(dev)
        self.n_in, self.n_all = len(self.static_in), len(every)
        if verify:
            self._verify(dev, run if against_eager else None, every)
            trace("verified")
        outer = self

        class _Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *args):
                if outer.n_in:
                    torch._foreach_copy_([t.detach() for t in outer.static_in], [a.detach() for a in args[:outer.n_in]])
                outer.fwd.replay()
                return tuple(o.detach() for o in outer.static_out)

            @staticmethod
            def backward(ctx, *gos):
                torch._foreach_copy_(outer.static_go, [g if g is not None else torch.zeros_like(s)
                                                       for g, s in zip(gos, outer.static_go)])
                outer.bwd.replay()
                grads = [None] * outer.n_all
                for i, g in zip(outer.grad_idx, outer.static_gi):
                    grads[i] = g
                return tuple(grads)

        self._fn = _Fn

    def _verify(self, dev, eager_run=None, every=None):
        """Three forward + backward replays on the same inputs must agree (float atomics: to rounding) and be finite;
        with ``eager_run`` they must also agree with the eager evaluation.  Only meaningful for a deterministic region
        (no dropout: PyTorch's graph-safe generator draws new numbers on every replay)."""
        for g in self.static_go:
            g.fill_(1.0)
        ref = None
        if eager_run is not None:
            outs = eager_run()
            gr = torch.autograd.grad([o for o in outs if o.requires_grad], [every[i] for i in self.grad_idx],
                                     [torch.ones_like(o) for o in outs if o.requires_grad], allow_unused=True)
            ref = [t.detach().float().clone() for t in list(outs) + [g for g in gr if g is not None]]
        snaps = []
        for _ in range(3):
            self.fwd.replay()
            self.bwd.replay()
            torch.cuda.synchronize(dev)
            snaps.append([t.detach().float().clone() for t in list(self.static_out) + [g for g in self.static_gi if g is not None]])
        n_out = len(self.static_out)
        an = self.arg_names
        names = [f"output {k}" for k in range(n_out)] + [f"d({an[i] if an and i < len(an) else 'input %d' % i})"
                                                         for i, g in zip(self.grad_idx, self.static_gi) if g is not None]
        bad = []
        for k, (a, b) in enumerate(zip(snaps[0], snaps[2])):
            scale = float(a.abs().max()) + 1e-20
            if not bool(torch.isfinite(b).all()):
                bad.append(f"{names[k]} {tuple(b.shape)}: non-finite")
            elif float((a - b).abs().max()) > 2e-2 * scale:      # float atomics reorder sums; a garbage fill is O(1) or NaN
                bad.append(f"{names[k]} {tuple(b.shape)}: replays differ by {float((a - b).abs().max()):.2e} of {scale:.2e}")
        if ref is not None:
            for k, (a, b) in enumerate(zip(ref, snaps[2])):
                scale = float(a.abs().max()) + 1e-20
                if a.shape != b.shape or float((a - b).abs().max()) > 3e-2 * scale:
                    bad.append(f"{names[k]} {tuple(b.shape)}: replay differs from eager by "
                               f"{float((a - b).abs().max()):.2e} of {scale:.2e}")
        if bad:
            raise RuntimeError("hipGraph replay check failed (a memset node in the captured region?): " + "; ".join(bad[:12])
                               + (f" ... and {len(bad) - 12} more" if len(bad) > 12 else ""))
        for g in self.static_go:
            g.zero_()

    def matches(self, args, static_args=()):
        return len(args) == len(self.args_meta) and all(
            (tuple(a.shape), a.dtype, a.requires_grad) == m for a, m in zip(args, self.args_meta)) and \
            len(static_args) == len(self.static_meta) and all(
            (a.data_ptr(), tuple(a.shape), a.dtype) == m for a, m in zip(static_args, self.static_meta))

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
