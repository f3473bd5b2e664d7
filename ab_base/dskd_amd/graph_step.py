"""The distillation training step as replayable hipGraphs.

An eager step issues ~3 700 kernel launches and is host-bound on MI355X (68 ms wall for 60 ms
of GPU work at B=4).  When the shapes of a step repeat (same image size, same number of GT
and teacher boxes per image) the launch sequence is identical, so it is captured once and
replayed:

    graph T   teacher backbone + neck + transformer + heads          (no_grad, eval)
    eager     teacher box decode (score threshold / top-k: data-dependent sizes, one sync)
    graph F   student backbone + neck forward (autograd graph kept: its buffers are static)
    graph S   transformer + heads + all losses + backward down to the neck outputs
    eager     backward of neck + backbone through the autograd graph recorded under F
    eager     [world > 1] ONE all-reduce of the flattened gradients over RCCL + log scalars
    graph U   global-norm gradient clip + fused AdamW update

STATUS: experimental, opt-in (``bench.py --graph``), NOT trusted at full scale on this image.
Root cause, pinned on the GPU (scratch/graph_memset_repro.py, scratch/graph_debug.py): a
``hipMemsetAsync`` captured into a hipGraph replays with a garbage fill value from the second
replay on (ROCm runtime bundled with torch 2.10+rocm7.0).  Three users of memset nodes sit
inside a training step:
  * MIOpen's split-K weight-gradient kernels zero their output with it -> whole tiles of conv
    weight gradients came back NaN.  That is why the neck/backbone backward runs EAGERLY here
    (its ~500 launches queue behind graph S on the stream, so the GPU never waits for them);
  * our own kernels used it for two workspaces -> they now zero with a fill kernel
    (csrc/common.h ``zero_fill``);
  * ATen's multi-block reductions (``Reduce.cuh``: bias-gradient column sums, loss sums) reset
    their semaphores with it -> single elements of bias gradients come back NaN, other
    reductions silently return the previous replay's value.  There is no clean way around
    this one from outside ATen, so the eager step stays the default execution mode.
Small shapes (single-block reductions, no split-K) replay correctly: tests/test_gpu_model.py.

For data parallelism the gradients are flattened into one buffer and exchanged as a single
large collective -- the shape xGMI likes -- and nothing inside a captured
region talks to the host or to another rank.  Requirements
met elsewhere in the package: no host->device copies inside the step (``utils.device_const``),
no ``.item()``/``nonzero`` in the loss path, custom kernels launched on the capturing stream
with their arguments passed by value.

Shapes that were not seen before run eagerly (and are captured after ``warmup`` repeats), so
variable-size data stays correct; the graphs are an accelerator, not a requirement.
"""
import torch
import torch.distributed as dist

from .dist import get_dist_info
from .utils import no_gc_during_capture


class GraphedDistillStep:
    def __init__(self, model, optimizer, amp_dtype=None, max_norm=0.1, norm_type=2, use_graphs=True, warmup=3):
        self.model, self.opt = model, optimizer
        self.amp_dtype, self.max_norm, self.norm_type = amp_dtype, max_norm, norm_type
        self.use_graphs = use_graphs and torch.cuda.is_available()
        self.warmup = warmup
        self.rank, self.world = get_dist_info()
        self.dev = next(model.parameters()).device
        self._graphs = {}
        self._seen = {}
        self.last_logs = None
        self._avg_pos = None
        # this class captures the whole step itself: the head's own forward / backward graphs (utils.GraphedFunction)
        # would be replayed inside its eager warm-up steps and then sit inside its captures
        module = model.module if hasattr(model, "module") else model
        if hasattr(module, "bbox_head"):
            module.bbox_head.graph_head = False

    # ------------------------------------------------------------------ pieces of a step
    def _autocast(self):
        return torch.autocast(device_type=self.dev.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None)

    def _teacher(self, data):
        with torch.no_grad(), self._autocast():
            feats = self.model.teacher_model.extract_feat(data["img"])
            outs = self.model.teacher_model.bbox_head.forward(feats, data["img_metas"])
        return feats, outs

    def _decode(self, outs, data):
        m = self.model
        with torch.no_grad():
            cfg = m.teacher_test_cfg if m.teacher_test_cfg is not None else m.test_cfg
            pred = m.teacher_model.bbox_head.get_bboxes(*outs, img_metas=data["img_metas"], rescale=False, cfg=cfg,
                                                        need_logits=True)
            keep = torch.cat([r[3] + i * outs[0].shape[2] for i, r in enumerate(pred)])
            return dict(pred_bboxes=[r[0][:, 0:4] for r in pred], pred_scores=[r[0][:, 4] for r in pred],
                        pred_labels=[r[1] for r in pred], pred_logits=[r[2] for r in pred], pred_keepid=keep)

    def _fwd_bwd(self, data, feats, outs, det):
        self.opt.zero_grad(set_to_none=True)
        with self._autocast():
            ti = {"neck_feats": feats if self.model.bbox_head.feats_distill else None, "head_outs": outs,
                  "pred_keepid": det["pred_keepid"], "pred_logits": det.get("pred_logits"),
                  "pred_scores": det.get("pred_scores"), "pred_labels": det["pred_labels"],
                  "pred_bboxes": det["pred_bboxes"]}
            losses = self.model(img=data["img"], img_metas=data["img_metas"], gt_bboxes=data["gt_bboxes"],
                                gt_labels=data["gt_labels"], teacher_info=ti)
            loss, keys, flat = self.model.parse_losses_local(losses)
        loss.backward()
        return loss.detach(), keys, flat

    def _student_feats(self, data):
        with self._autocast():
            return self.model.extract_feat(data["img"])

    def _head_fwd_bwd(self, data, xs, feats, outs, det):
        """Student head on the (detached) neck outputs ``xs``: losses and backward; the
        gradients of ``xs`` are what the eager neck/backbone backward continues from."""
        m = self.model
        for meta in data["img_metas"]:
            meta.setdefault("batch_input_shape", tuple(data["img"].shape[-2:]))
        with self._autocast():
            ti = {"neck_feats": feats if m.bbox_head.feats_distill else None, "head_outs": outs,
                  "pred_keepid": det["pred_keepid"], "pred_logits": det.get("pred_logits"),
                  "pred_scores": det.get("pred_scores"), "pred_labels": det["pred_labels"],
                  "pred_bboxes": det["pred_bboxes"]}
            losses = m.bbox_head.forward_train(xs, data["img_metas"], data["gt_bboxes"], data["gt_labels"], None,
                                               proposal_cfg=None, teacher_info=ti, task_labels=m.LableInPCNTask)
            loss, keys, flat = m.parse_losses_local(losses)
        loss.backward()
        return loss.detach(), keys, flat

    def _feat_params(self):
        m = self.model
        mods = [m.backbone] + ([m.neck] if m.with_neck else [])
        return [p for mod in mods for p in mod.parameters() if p.requires_grad]

    def _update(self):
        params = [p for g in self.opt.param_groups for p in g["params"] if p.grad is not None]
        torch.nn.utils.clip_grad_norm_(params, max_norm=self.max_norm, norm_type=self.norm_type, foreach=True)
        self.opt.step()

    def _set_avg_pos(self, data, det):
        """clamp(mean over ranks of num_total_pos, 1) (gfl_deformable_detr_head_il.py:1491-1492),
        computed from host-known box counts and written into a static device scalar read by the
        captured loss."""
        head = self.model.bbox_head
        Q = head.num_query
        n = 0
        for i, g in enumerate(data["gt_bboxes"]):
            extra = det["pred_bboxes"][i].shape[0] if (self.model.has_teacher and "hard" in head.cates_distill) else 0
            n += min(Q, g.shape[0] + extra)
        if self._avg_pos is None:
            self._avg_pos = torch.zeros((), dtype=torch.float32, device=self.dev)
        t = self._avg_pos                      # ONE static scalar: the captured loss reads this address
        head.avg_pos_static = t
        t.fill_(float(n))
        if self.world > 1:
            dist.all_reduce(t)
            t.div_(self.world)
        t.clamp_(min=1)

    def _exchange(self, flat_logs):
        """Everything that crosses ranks, outside the graphs: one gradient all-reduce (mean) and
        one small all-reduce of the log scalars."""
        if self.world > 1:
            grads = [p.grad for g in self.opt.param_groups for p in g["params"] if p.grad is not None]
            flat = torch._utils._flatten_dense_tensors(grads)
            dist.all_reduce(flat)
            flat.div_(self.world)
            torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))
            dist.all_reduce(flat_logs)
            flat_logs.div_(self.world)
        return flat_logs

    # ------------------------------------------------------------------ public
    @staticmethod
    def _signature(data, det):
        return (id(data["img"]), tuple(data["img"].shape), data["img"].dtype, tuple(tuple(b.shape) for b in data["gt_bboxes"]),
                None if det is None else tuple(tuple(b.shape) for b in det["pred_bboxes"]),
                tuple(tuple(m["img_shape"]) for m in data["img_metas"]))

    def eager_step(self, data, inject=None):
        self.model.bbox_head.avg_pos_static = None          # eager: the head reduces it itself
        feats, outs = self._teacher(data)
        det = self._decode(outs, data)
        if inject is not None:
            det = dict(det, **inject)
        loss, keys, flat = self._fwd_bwd(data, feats, outs, det)
        flat = self._exchange(flat)
        self._update()
        self.last_logs = (keys, flat)
        return loss

    def step(self, data, inject=None):
        """One optimisation step.  ``inject``: optional dict overriding the decoded teacher
        detections (``pred_bboxes / pred_labels / pred_keepid``), as the benchmark does for an
        untrained teacher.  ``data`` tensors must be the SAME objects (static buffers) across
        calls for a signature that has been captured."""
        if not self.use_graphs or inject is None:
            # decoded detections have data-dependent sizes: only the injected (static) form is graphed
            return self.eager_step(data, inject)
        sig = self._signature(data, inject)
        g = self._graphs.get(sig)
        if g is None:
            n = self._seen.get(sig, 0)
            self._seen[sig] = n + 1
            if n < self.warmup:
                return self.eager_step(data, inject)
            g = self._capture(data, inject)
            self._graphs[sig] = g
        from . import native
        native.advance_dropout_epoch(self.dev)  # the captured dropout launches read it: new masks on every replay
        g["T"].replay()
        self._decode(g["outs"], data)           # executed (and synchronising) as in the eager step
        self._set_avg_pos(data, inject)
        g["F"].replay()
        g["S"].replay()
        # neck + backbone backward, eager, accumulating in place into the static gradients F zeroed
        torch.autograd.backward(g["xs_raw"], [x.grad for x in g["xs"]], retain_graph=True)
        flat = self._exchange(g["flat_logs"])
        g["U"].replay()
        self.last_logs = (g["keys"], flat)
        return g["loss"]

    def _capture(self, data, inject):
        torch.cuda.synchronize()
        gT = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(gT):
            feats, outs = self._teacher(data)
        gT.replay()                             # capture does not execute: produce real outputs
        det = dict(self._decode(outs, data), **inject)
        self._set_avg_pos(data, det)
        # static gradients of the feature extractor (eager backward accumulates into them)
        fparams = self._feat_params()
        fgrads = [torch.zeros_like(p) for p in fparams]
        for p, gr in zip(fparams, fgrads):
            p.grad = gr
        fset = {id(p) for p in fparams}
        for grp in self.opt.param_groups:
            for p in grp["params"]:
                if id(p) not in fset:
                    p.grad = None               # head gradients are allocated inside graph S
        gF = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(gF):
            torch._foreach_zero_(fgrads)
            xs_raw = self._student_feats(data)
        xs = [f.detach().requires_grad_(True) for f in xs_raw]
        gS = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(gS):
            loss, keys, flat_logs = self._head_fwd_bwd(data, xs, feats, outs, det)
        gF.replay()
        gS.replay()
        torch.autograd.backward(xs_raw, [x.grad for x in xs], retain_graph=True)
        gU = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(gU):
            self._update()                      # the first real update happens on the next replay
        torch.cuda.synchronize()
        from .utils import const_cache_snapshot
        return dict(T=gT, F=gF, S=gS, U=gU, outs=outs, feats=feats, xs_raw=xs_raw, xs=xs, loss=loss, keys=keys,
                    flat_logs=flat_logs,
                    keepalive=(const_cache_snapshot(), det, data, inject, fgrads))   # everything the graphs point at

    def logs(self):
        """Host copy of the last step's log vars (one device->host copy)."""
        if self.last_logs is None:
            return {}
        keys, flat = self.last_logs
        return dict(zip(keys, flat.detach().cpu().tolist()))
