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


def reduce_mean(tensor):
    """dist_utils.py:68-74."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


def allreduce_scalars(values):
    """Mean over ranks of a list of 0-dim tensors with ONE collective; returns a 1-D tensor."""
    flat = torch.stack([v.detach().float().reshape(()) for v in values])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat.div_(dist.get_world_size()))
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
