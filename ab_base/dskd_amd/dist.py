"""Distributed helpers: one process per GPU, ``torch.distributed`` over RCCL/xGMI
(backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

Counterparts in the reference: ``init_dist`` (/root/reference/tools/train_increment.py:146-153,
ext-mmcv), ``reduce_mean`` (/root/reference/mmdet/core/utils/dist_utils.py:68-74), the
``MMDistributedDataParallel`` wrap (/root/reference/tools/train_increment.py:301-303) and
the per-key logging all-reduces of ``_parse_losses``
(/root/reference/mmdet/models/detectors/deformable_detr_il.py:236-251), which are replaced
by ONE coalesced all-reduce (``allreduce_scalars``)."""
import os

import torch
import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def ranks_share_a_device():
    """True when more than one rank of this job drives the same GPU (the one-GPU rehearsal of the multi-process
    path: ``DSKD_BENCH_REHEARSE``, or more local ranks than devices).  The compute queues of different PROCESSES on one
    GPU are time-sliced by the hardware scheduler: a 0.75 ms kernel of one rank was bracketed at 120-134 ms while the
    other rank's queue held the device, and replaying hipGraphs (hundreds of nodes per submission) stretched a step to
    5-10 s (gpurun_out/rehearse.json, rh.out of round 1).  That is a property of sharing the card, not of the graphs or
    of the teacher side stream; with one rank per GPU -- the only production layout -- no queue of another process
    exists.  Graph replays are therefore switched off exactly in this situation."""
    if os.environ.get("DSKD_BENCH_REHEARSE"):
        return True
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return False
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", dist.get_world_size()))
    return torch.cuda.is_available() and local_world > torch.cuda.device_count()


def hipgraphs_allowed():
    """hipGraph replays of training regions (dense losses, student head): single process, or one rank per GPU over
    RCCL.  ``DSKD_FORCE_GRAPHS=1`` / ``DSKD_NO_GRAPHS=1`` override."""
    if os.environ.get("DSKD_NO_GRAPHS"):
        return False
    if os.environ.get("DSKD_FORCE_GRAPHS"):
        return True
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    return dist.get_backend() == "nccl" and not ranks_share_a_device()


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    if dist.is_initialized():
        return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)


def all_reduce_sum(tensor, async_op=False):
    """``dist.all_reduce(tensor, SUM)``.  GPU tensors over the gloo backend (the shared-GPU rehearsal of the multi-rank path,
    never a production layout) are staged through the host: gloo's own device path faulted intermittently on this image when
    two ranks drove one GPU (r4: 'Memory access fault ... write access to a read-only page' inside its reduction, with
    torch DDP as well as with GradSync); its host path is what the CPU tests exercise.  Returns the work handle or None."""
    if tensor.is_cuda and dist.get_backend() == "gloo":
        host = tensor.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        tensor.copy_(host)
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=async_op)


def reduce_mean(tensor):
    """dist_utils.py:68-74."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    tensor = tensor.clone()
    all_reduce_sum(tensor.div_(dist.get_world_size()))
    return tensor


def allreduce_scalars(values):
    """Mean over ranks of a list of 0-dim tensors with ONE collective; returns a 1-D tensor."""
    flat = torch.stack([v.detach().float().reshape(()) for v in values])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        all_reduce_sum(flat.div_(dist.get_world_size()))
    return flat


def wrap_ddp(model, device_ids=None, bucket_cap_mb=50, find_unused_parameters=None, **kwargs):
    """DDP over RCCL with bucketed gradient all-reduce overlapped with backward.
    The reference wraps with ``find_unused_parameters=True`` (train_increment.py:301-303) because
    the head's ``prototype`` embedding never receives a gradient; that makes DDP walk the autograd
    graph and all-reduce a used-parameter bitmap every iteration.  Here the parameters that are
    unused BY CONSTRUCTION (``*.prototype.weight``) are excluded from DDP instead
    (``_ddp_params_and_buffers_to_ignore``) and the search is off; pass
    ``find_unused_parameters=True`` to get the reference behaviour.  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): ~160 MB of fp32 gradients in 50 MB buckets keeps several
    ring steps in flight per link while the backward still runs."""
    from torch.nn.parallel import DistributedDataParallel
    if find_unused_parameters is None:
        ignore = [n for n, _ in model.named_parameters() if n.endswith("prototype.weight")]
        if ignore:
            DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, ignore)
        find_unused_parameters = False
    return DistributedDataParallel(model, device_ids=device_ids, broadcast_buffers=False,
                                   find_unused_parameters=find_unused_parameters, bucket_cap_mb=bucket_cap_mb,
                                   gradient_as_bucket_view=True, **kwargs)


_GRAD_SLOTS = {}      # id(parameter) -> its view in the flat buffer of the live GradSync


def grad_slot(pid, shape, dtype):
    """A fresh alias of the GradSync slot of the parameter with ``id`` ``pid`` (None without a live GradSync, or when shape
    / dtype differ): the multi-tensor casts that produce most parameter gradients (transformer._CastParams, backbones.
    _FoldTrainable) write straight into it, autograd adopts the alias as ``.grad`` (AccumulateGrad keeps a gradient that
    nobody else references and that has the parameter's strides), and GradSync's bucket copy has nothing left to move."""
    v = _GRAD_SLOTS.get(pid)
    if v is None or v.dtype != dtype or tuple(v.shape) != tuple(shape):
        return None
    return v.detach()


class GradSync:
    """Data-parallel gradient exchange WITHOUT the DDP wrapper (r4): one persistent flat f32 buffer holds every trainable
    parameter's gradient, cut into buckets in the order the backward produces them (reverse registration: head and
    transformer first, backbone last).  A post-accumulate hook per parameter counts its bucket down; the last one packs
    the bucket's fresh gradients into its slice with ONE multi-tensor copy, re-points ``.grad`` at the slice views and
    starts the bucket's all-reduce (RCCL: on the process group's stream, overlapped with the rest of the backward).
    :meth:`finish` (after ``backward()``, before the optimizer) flushes what is left, waits, and averages.

    Why not ``DistributedDataParallel`` (what the reference wraps with, tools/train_increment.py:301-303): its reducer copies
    every parameter's gradient into the bucket with a launch of its own (~200 per step: the hand-written weight-gradient
    kernels hand autograd freshly allocated tensors, ``gradient_as_bucket_view`` cannot adopt them) and costs 3 ms per step
    at ONE rank, before a byte crosses xGMI (profiles/r03_ddp_one_rank_rccl.json).  Same numbers as DDP: mean over ranks of
    the local gradients; a parameter that received no gradient on this rank contributes zeros; ``*.prototype.weight``
    (never used, transformer.py / head) is left out as in :func:`wrap_ddp`."""

    def __init__(self, model, bucket_mb=48.0, overlap=True, force=False):
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if inited else 1
        self.active = self.world > 1 or force
        self.comm = inited and self.active       # force: the collectives run even in a one-rank group (diagnostic)
        if os.environ.get("DSKD_GRADSYNC_NOCOMM") and self.world == 1:
            self.comm = False                    # diagnostic: hooks + packing only
        bucket_mb = float(os.environ.get("DSKD_GRADSYNC_BUCKET_MB", bucket_mb))
        self.overlap = overlap
        if self.world > 1:                       # what DDP's constructor does: rank 0's parameters and buffers everywhere
            with torch.no_grad():
                stage = dist.get_backend() == "gloo"          # see all_reduce_sum
                for t in list(model.parameters()) + list(model.buffers()):
                    if stage and t.is_cuda:
                        host = t.data.cpu()
                        dist.broadcast(host, 0)
                        t.data.copy_(host)
                    else:
                        dist.broadcast(t.data, 0)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not n.endswith("prototype.weight")]
        self.params = [p for _, p in reversed(named)]
        self.buckets, self.handles = [], []
        self.stats = dict(copied=0, adopted=0, zeroed=0, flushed_in_backward=0, flushed_in_finish=0)
        if not self.active or not self.params:
            return
        dev, dt = self.params[0].device, torch.float32
        if any(p.device != dev or p.dtype != dt or not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last))
               for p in self.params):
            raise ValueError("GradSync: dense f32 parameters on one device expected")
        total = sum((p.numel() + 7) // 8 * 8 for p in self.params)      # every slot 32-byte aligned (vector casts into it)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        cap = int(bucket_mb * (1 << 20) / 4)
        lo = off = 0
        cur = []
        self.views = {}
        for p in self.params:
            v = self.flat[off:off + p.numel()]
            # a dense parameter in any memory format (channels_last convolution weights): same strides as the parameter
            # (as_strided also for 'contiguous' ones: a [N, C, 1, 1] channels_last weight reports both formats)
            self.views[p] = v.as_strided(p.shape, p.stride())
            cur.append(p)
            off += (p.numel() + 7) // 8 * 8
            if off - lo >= cap:
                self.buckets.append(dict(params=cur, lo=lo, hi=off, pending=len(cur), work=None, done=False))
                cur, lo = [], off
        if cur:
            self.buckets.append(dict(params=cur, lo=lo, hi=off, pending=len(cur), work=None, done=False))
        self._avg = self.comm and dist.get_backend() == "nccl"
        owner = {}
        for b in self.buckets:
            for p in b["params"]:
                owner[p] = b
        for p in self.params:
            self.handles.append(p.register_post_accumulate_grad_hook(self._make_hook(owner[p])))
            _GRAD_SLOTS[id(p)] = self.views[p]

    def _make_hook(self, bucket):
        def hook(param):
            bucket["pending"] -= 1
            if bucket["pending"] == 0 and self.overlap and not bucket["done"]:
                self._flush(bucket)
        return hook

    @torch.no_grad()
    def _flush(self, b):
        have = [p for p in b["params"] if p.grad is not None and p.grad.data_ptr() != self.views[p].data_ptr()]
        for p in b["params"]:
            if p.grad is None:
                self.views[p].zero_()
        st = self.stats
        st["copied"] += len(have)
        st["zeroed"] += sum(1 for p in b["params"] if p.grad is None)
        st["adopted"] += sum(1 for p in b["params"] if p.grad is not None) - len(have)
        st["flushed_in_backward" if b["pending"] == 0 else "flushed_in_finish"] += 1
        if have:
            torch._foreach_copy_([self.views[p] for p in have], [p.grad for p in have])
        for p in b["params"]:
            p.grad = self.views[p]
        if self.comm:
            if self._avg:
                b["work"] = dist.all_reduce(self.flat[b["lo"]:b["hi"]], op=dist.ReduceOp.AVG, async_op=True)
            else:
                b["work"] = all_reduce_sum(self.flat[b["lo"]:b["hi"]], async_op=True)
        b["done"] = True

    @torch.no_grad()
    def finish(self):
        """Call between ``backward()`` and the optimizer step."""
        if not self.active:
            return
        for b in self.buckets:
            if not b["done"]:
                self._flush(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            b["pending"], b["done"] = len(b["params"]), False
        if self.comm and not self._avg and self.world > 1:
            self.flat.div_(self.world)

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        if self.active:
            for p in self.params:
                if _GRAD_SLOTS.get(id(p)) is self.views.get(p):
                    del _GRAD_SLOTS[id(p)]
