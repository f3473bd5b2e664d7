"""``DeformableDETR_il``: student detector holding a frozen, un-registered teacher copy.

Restated from /root/reference/mmdet/models/detectors/deformable_detr_il.py
(ctor :36-77, ``set_teacher`` :79-114, ``out_teacher`` :116-152, ``set_student`` :154-160,
``set_datainfo`` :172-181, ``extract_feat`` :183-188, ``_parse_losses`` :210-253,
``forward_train`` :255-318, ``train_step`` :419-450, ``cuda`` / ``train`` /
``__setattr__`` :467-496).  ``_parse_losses`` keeps the reference's key set and values but
reduces all log scalars with one collective and one device->host copy."""
import copy
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .builder import DETECTORS, build_backbone, build_head, build_neck
from .dist import allreduce_scalars
from .utils import no_gc_during_capture


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)


def _scalar_mean(v):
    """``v.mean()`` -- which for the 0-dim loss terms of this head is the value itself: no launch forward, no MeanBackward
    division backward (26 terms per step)."""
    return v if v.dim() == 0 else v.mean()


def _sum_scalars(vals):
    """Sum of 0-dim tensors.  On the GPU: one stack + one sum (and views backward) instead of a chain of n - 1 add launches and
    their backward nodes -- a step's ~26 loss terms sat as ~130 launches of 2 us on the main stream.  f32 summation order
    differs from the sequential chain in the last bit.  CPU: the reference's sequential sum (mmdet's _parse_losses)."""
    if len(vals) > 2 and all(v.is_cuda and v.dim() == 0 and v.dtype == vals[0].dtype for v in vals):
        return torch.stack(vals).sum()
    return sum(vals)


class TeacherAhead:
    """Runs the frozen teacher of the NEXT batch on a second HIP stream while the student's
    backward of the current batch is executing.

    The teacher's box decode (``get_bboxes``: score threshold / top-k, data-dependent sizes) ends
    in a device->host copy.  Inline (``out_teacher`` inside ``forward_train``) that copy drains
    the stream in the middle of every step: the host then has to enqueue the whole student
    forward/backward (~2 800 launches) with the GPU idling behind it -- measured 15-20 % of the
    step at B=4.  Here the copy only waits for the side stream, and by the time the host asks
    for it the main stream still holds the queued backward + optimizer of the previous batch:

        launch(img, metas)   enqueue teacher backbone/neck/transformer/heads on the side stream
                             (a hipGraph replay once the batch signature repeats)
        finish()             decode on the side stream (host waits for THAT stream only), make
                             the main stream wait for it, hand the tensors over

    The teacher is frozen, so computing it one batch ahead changes no number.  Use:
    ``finish()`` at the top of a step (falls back to an inline teacher when nothing is
    pending), ``launch(next batch)`` right after the student forward has been enqueued."""

    def __init__(self, detector, use_graphs=True, graph_warmup=2, max_graphs=4):
        self.det = detector
        self.stream = None
        self.pending = None
        self.use_graphs = use_graphs
        self.graph_warmup = graph_warmup
        self.max_graphs = max_graphs          # batch signatures kept as graphs (each holds its activations twice)
        self._graphs, self._seen, self._flip = {}, {}, 0

    # The teacher FORWARD is also replayed as a hipGraph once a batch signature has been seen
    # ``graph_warmup`` times: ~550 launches (8 ms of host time) become one.  This is safe where the
    # training step as a whole is not (graph_step.py): the no-grad forward contains no memset
    # node (checked with the profiler) and replays with changing inputs track eager results
    # (scratch/teacher_graph_check.py); the capture is verified against the eager result right
    # away and dropped if it disagrees.  Two graphs with separate output buffers alternate, so
    # batch i's teacher tensors stay intact while batch i+1's forward runs beside the student's
    # backward of batch i.
    def _forward(self, img, img_metas, amp_dtype):
        det = self.det
        with torch.no_grad(), torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
            feats = det.teacher_model.extract_feat(img)
            outs = det._teacher_heads(feats, img_metas)
        return feats, outs

    @staticmethod
    def _signature(img, img_metas, amp_dtype):
        return (tuple(img.shape), img.dtype, img.is_contiguous(memory_format=torch.channels_last), amp_dtype,
                tuple(tuple(m["img_shape"]) for m in img_metas),
                tuple(tuple(m.get("batch_input_shape", ())) for m in img_metas))

    def _capture(self, img, img_metas, amp_dtype):
        """Two graphs (double buffer) on the side stream; None when capture fails or the replay does
        not reproduce the eager forward (the eager path then stays in use for this signature)."""
        entries = []
        try:
            with torch.cuda.stream(self.stream):
                ref_feats, ref_outs = self._forward(img, img_metas, amp_dtype)
            for _ in range(2):
                static = img.detach().clone(memory_format=torch.preserve_format)
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads of the process (RCCL watchdog under DDP) keep making HIP
                # calls while this thread captures
                with no_gc_during_capture(), torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    feats, outs = self._forward(static, img_metas, amp_dtype)
                g.replay()
                pins = getattr(self.det.teacher_model.bbox_head, "graph_pins", lambda d: [])(img.device)
                entries.append(dict(graph=g, img=static, feats=feats, outs=outs, keepalive=pins))
            torch.cuda.synchronize(img.device)
        except Exception as e:  # noqa: BLE001  (an accelerator, not a requirement)
            import warnings
            warnings.warn(f"TeacherAhead: hipGraph capture failed ({type(e).__name__}: {e}); staying eager")
            torch.cuda.synchronize(img.device)
            return None
        for e in entries:
            for a, r in zip(_tensors((e["feats"], e["outs"])), _tensors((ref_feats, ref_outs))):
                if a.dtype.is_floating_point:
                    scale = float(r.float().abs().max()) + 1e-6
                    if not bool(torch.isfinite(a.float()).all()) or float((a.float() - r.float()).abs().max()) > 0.05 * scale:
                        return None
        return entries

    def launch(self, img, img_metas, amp_dtype=None):
        """Enqueue the teacher forward of ``img`` on the side stream (eagerly, or as a graph replay).
        ``amp_dtype``: autocast dtype to run under."""
        if not img.is_cuda:
            self.pending = ("inline", img, img_metas)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=img.device)
        main = torch.cuda.current_stream(img.device)
        entries = None
        if self.use_graphs:
            sig = self._signature(img, img_metas, amp_dtype)
            entries = self._graphs.get(sig)
            if entries is None and sig not in self._graphs:
                n = self._seen.get(sig, 0)
                self._seen[sig] = n + 1
                if n >= self.graph_warmup and sum(1 for v in self._graphs.values() if v) < self.max_graphs:
                    self.stream.wait_stream(main)
                    entries = self._graphs[sig] = self._capture(img, img_metas, amp_dtype)   # None = keep eager
        self.stream.wait_stream(main)                      # the batch (and anything it depends on) is ready
        with torch.cuda.stream(self.stream):
            if entries:
                e = entries[self._flip]
                self._flip ^= 1
                e["img"].copy_(img, non_blocking=True)
                e["graph"].replay()
                feats, outs = e["feats"], e["outs"]
            else:
                feats, outs = self._forward(img, img_metas, amp_dtype)
        self.pending = ("ahead", feats, outs, img_metas, img)

    def invalidate(self):
        """Drop the captured graphs, the seen-signature counters and any pending batch (the teacher they
        were made with is gone, or the detector moved)."""
        if self.stream is not None:
            self.stream.synchronize()        # nothing of the old teacher is still being replayed
        self.pending = None
        self._graphs, self._seen, self._flip = {}, {}, 0

    def discard(self):
        """Forget a launched batch that will not be consumed (end of an epoch cut short, end of a task)."""
        self.pending = None

    @staticmethod
    def _same_batch(pend_img, pend_metas, img, img_metas):
        """Does the pending entry belong to the batch the caller is about to train on?  The same tensor
        (identity or storage + shape) and the same per-image shapes."""
        if img is None:
            return True                      # caller did not say: it consumes what it launched
        if pend_img is not img and (pend_img.data_ptr() != img.data_ptr() or pend_img.shape != img.shape
                                    or pend_img.dtype != img.dtype):
            return False
        if img_metas is not None and pend_metas is not img_metas:
            if len(pend_metas) != len(img_metas):
                return False
            for a, b in zip(pend_metas, img_metas):
                if tuple(a.get("img_shape", ())) != tuple(b.get("img_shape", ())):
                    return False
        return True

    def finish(self, img=None, img_metas=None):
        """teacher_info of the launched batch (same dict as ``forward_train`` builds).  With ``img`` given, a
        pending entry launched for ANOTHER batch (a loop that broke after launching, a new epoch / task) is
        discarded and the teacher runs inline on ``img``."""
        det = self.det
        pend, self.pending = self.pending, None
        if pend is not None and not self._same_batch(pend[-1] if pend[0] == "ahead" else pend[1],
                                                     pend[3] if pend[0] == "ahead" else pend[2], img, img_metas):
            pend = None
        if pend is None or pend[0] == "inline":
            if pend is not None:
                img, img_metas = pend[1], pend[2]
            feats, outs, keepid, logits, labels, scores, bboxes = det.out_teacher(img, img_metas, cat_keepid=True)
        else:
            _, feats, outs, img_metas, _ = pend
            main = torch.cuda.current_stream(feats[0].device)
            with torch.cuda.stream(self.stream), torch.no_grad():
                cfg = det.teacher_test_cfg if det.teacher_test_cfg is not None else det.test_cfg
                pred = det.teacher_model.bbox_head.get_bboxes(*outs, img_metas=img_metas, rescale=False, cfg=cfg,
                                                              need_logits=True)
                bboxes = [r[0][:, 0:4] for r in pred]
                scores = [r[0][:, 4:5].flatten() for r in pred]
                labels, logits = [r[1] for r in pred], [r[2] for r in pred]
                stride = det._keepid_stride(outs)
                keepid = torch.cat([r[3] + i * stride for i, r in enumerate(pred)])
            main.wait_stream(self.stream)
            # allocated on the side stream, consumed on the main one: keep the allocator from
            # recycling them before the main stream is done
            for t in _tensors((feats, outs, bboxes, scores, labels, logits, keepid)):
                t.record_stream(main)
        return {"neck_feats": feats if det.bbox_head.feats_distill else None, "head_outs": outs,
                "pred_keepid": keepid, "pred_logits": logits or None, "pred_scores": scores,
                "pred_labels": labels, "pred_bboxes": bboxes}


@DETECTORS.register_module()
class DeformableDETR_il(nn.Module):
    def __init__(self, backbone, neck, bbox_head, teacher_config=None, teacher_ckpt=None, eval_teacher=True,
                 teacher_test_cfg=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        object.__setattr__(self, "has_teacher", bool(teacher_config and teacher_ckpt))
        backbone = dict(backbone)
        if pretrained:
            backbone["pretrained"] = pretrained
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(dict(neck)) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg, has_teacher=self.has_teacher)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg, self.teacher_test_cfg = train_cfg, test_cfg, teacher_test_cfg
        self.Label2CatNameId = dict()
        self.LableInPCNTask = {"prev": [], "curr": [], "next": []}
        self.eval_teacher = eval_teacher
        self.teacher_model = None
        self.lazy_log = False   # True: train_step returns device log vars (no host sync)
        if self.has_teacher:    # :70-74 -- the teacher is built from its config and checkpoint right here
            self.set_teacher(config=teacher_config, ckptfile=teacher_ckpt, trainval="val")

    @property
    def with_neck(self):
        return self.neck is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()

    # ------------------------------------------------------------------ teacher / task state
    def _drop_teacher_ahead(self):
        """Forget the ahead-of-time teacher pipeline: its hipGraphs hold raw pointers into the weights and
        activations of the teacher they were captured with, and a pending batch belongs to that teacher.
        Called wherever the teacher (or the device / layout of the detector) changes."""
        ta = self.__dict__.pop("_teacher_ahead", None)
        if ta is not None:
            ta.invalidate()

    def set_teacher(self, config=None, ckptfile=None, model=None, trainval="val"):
        """:79-114."""
        self._drop_teacher_ahead()
        if (config is None or ckptfile is None) and model is None:
            self.has_teacher = False
            self.bbox_head.has_teacher = False
            return None
        if model is None:
            from .builder import build_detector
            from .config import Config
            if isinstance(config, str):
                config = Config.fromfile(config)
            model = build_detector(config["model"])
            sd = torch.load(ckptfile, map_location="cpu")
            model.load_state_dict(sd.get("state_dict", sd), strict=False)
        self.has_teacher = True          # before the assignment: keeps it out of nn.Module registration
        self.teacher_model = model
        if trainval == "val":
            self.eval_teacher = True
            self.teacher_model.train(False)
            for _, p in self.teacher_model.named_parameters():
                p.requires_grad = False
        else:
            self.eval_teacher = False
            self.teacher_model.train(True)
        if getattr(self.teacher_model, "teacher_model", None) is not None:
            object.__setattr__(self.teacher_model, "teacher_model", None)
        if getattr(self.teacher_model, "has_teacher", False):
            self.teacher_model.has_teacher = False
            self.teacher_model.bbox_head.has_teacher = False
        self.bbox_head.has_teacher = True
        return self.teacher_model

    def set_student(self, ckptfile=None):
        if ckptfile is not None:
            sd = torch.load(ckptfile, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)
        return self

    def load_student(self, ckptfile):
        self.set_student(ckptfile)
        self._drop_teacher_ahead()
        if self.teacher_model is not None:
            self.teacher_model = None
            self.has_teacher = False
        return None

    def set_datainfo(self, cat2id, cat2label, pred_cat=[], load_cat=[], task_cat=[]):
        """:172-181."""
        catid2catname = {v: k for k, v in cat2id.items()}
        self.Label2CatNameId = {v: [catid2catname[k], k] for k, v in cat2label.items()}
        all_cat = []
        for cat in task_cat:
            all_cat.extend(cat)
        prev_label = [cat2label[cat2id[cat]] for cat in list(set(pred_cat) - set(load_cat))]
        curr_label = [cat2label[cat2id[cat]] for cat in load_cat]
        next_label = [cat2label[cat2id[cat]] for cat in list(set(all_cat) - set(pred_cat))]
        self.LableInPCNTask = {"prev": prev_label, "curr": curr_label, "next": next_label}

    def __setattr__(self, name, value):
        """:485-496 -- the teacher is a plain attribute: not in parameters(), state_dict(), DDP."""
        if name in ("teacher_model", "has_teacher") and (name == "has_teacher" or self.__dict__.get("has_teacher")):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def cuda(self, device=None):
        self._drop_teacher_ahead()
        if self.has_teacher and self.teacher_model is not None:
            self.teacher_model.cuda(device=device)
        return super().cuda(device=device)

    def to(self, *args, **kwargs):
        self._drop_teacher_ahead()
        if self.has_teacher and self.teacher_model is not None:
            self.teacher_model.to(*args, **kwargs)
        return super().to(*args, **kwargs)

    def train(self, mode=True):
        if self.has_teacher and self.teacher_model is not None:
            self.teacher_model.train(False if self.eval_teacher else mode)
        return super().train(mode)

    # ------------------------------------------------------------------ forward
    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def out_teacher(self, img, img_metas, cat_keepid=True):
        """:116-152."""
        assert self.has_teacher, "no teacher model is set"
        with torch.no_grad():
            neck_feat = self.teacher_model.extract_feat(img)
            head_outs = self._teacher_heads(neck_feat, img_metas)
            cfg = self.teacher_test_cfg if self.teacher_test_cfg is not None else self.test_cfg
            pred_outs = self.teacher_model.bbox_head.get_bboxes(*head_outs, img_metas=img_metas, rescale=False,
                                                                cfg=cfg, need_logits=True)
            pred_bboxes = [r[0][:, 0:4].detach() for r in pred_outs]
            pred_scores = [r[0][:, 4:5].flatten().detach() for r in pred_outs]
            pred_labels = [r[1].detach() for r in pred_outs]
            pred_logits = [r[2].detach() for r in pred_outs]
            pred_keepid = [r[3].detach() for r in pred_outs]
            if cat_keepid:
                stride = self._keepid_stride(head_outs)
                pred_keepid = torch.cat([pk + i * stride for i, pk in enumerate(pred_keepid)])
        return neck_feat, head_outs, pred_keepid, pred_logits, pred_labels, pred_scores, pred_bboxes

    # the two places where the teacher pipeline (inline here, one batch ahead in TeacherAhead) depends on the head's kind
    def _teacher_heads(self, feats, img_metas):
        return self.teacher_model.bbox_head.forward(feats, img_metas)

    @staticmethod
    def _keepid_stride(head_outs):
        """Predictions per image: image i's kept indices are offset by i times this in the concatenated ``pred_keepid``."""
        return head_outs[0].shape[2]

    def __deepcopy__(self, memo):
        from .utils import deepcopy_without
        return deepcopy_without(self, memo, ("_teacher_ahead",))

    def teacher_ahead(self):
        """The :class:`TeacherAhead` pipeline of this detector (created on first use)."""
        ta = self.__dict__.get("_teacher_ahead")
        if ta is None:
            ta = self.__dict__["_teacher_ahead"] = TeacherAhead(self)
        return ta

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """:190-208.  ``return_loss=False``: img / img_metas are double-nested (outer list = test-time
        augmentations), as the reference's test pipeline hands them over."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        """``BaseDetector.forward_test`` (/root/reference/mmdet/models/detectors/base.py:112-154): one augmentation
        only (``aug_test`` of the reference raises NotImplementedError in its head too)."""
        for var, name in [(imgs, "imgs"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError(f"{name} must be a list, but got {type(var)}")
        if len(imgs) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})")
        for img, img_meta in zip(imgs, img_metas):
            for m in img_meta:
                m["batch_input_shape"] = tuple(img.size()[-2:])
        if len(imgs) != 1:
            raise NotImplementedError("test-time augmentation is not implemented")
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, teacher_info=None):
        """:255-318.  ``teacher_info`` may be injected (bench / tests: synthetic teacher
        detections through the same dict); otherwise it is produced by
        ``out_teacher``."""
        for m in img_metas:
            m.setdefault("batch_input_shape", tuple(img.size()[-2:]))
        if teacher_info is None:
            teacher_info = {k: None for k in ("neck_feats", "head_outs", "pred_keepid", "pred_logits", "pred_scores",
                                              "pred_labels", "pred_bboxes")}
            if self.has_teacher:
                feats, outs, keepid, logits, labels, scores, bboxes = self.out_teacher(img, img_metas, cat_keepid=True)
                teacher_info = {"neck_feats": feats if self.bbox_head.feats_distill else None, "head_outs": outs,
                                "pred_keepid": keepid, "pred_logits": logits or None, "pred_scores": scores,
                                "pred_labels": labels, "pred_bboxes": bboxes}
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, proposal_cfg=None,
                                            teacher_info=teacher_info, task_labels=self.LableInPCNTask)

    def simple_test(self, img, img_metas, rescale=False):
        """:365-387 -- per image a list with one [n_c, 5] numpy array per class (``bbox2result``)."""
        from .bbox import bbox2result
        feat = self.extract_feat(img)
        results_list = self.bbox_head.simple_test(feat, img_metas, rescale=rescale)
        return [bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes) for det_bboxes, det_labels in results_list]

    # ------------------------------------------------------------------ step
    def _parse_losses(self, losses):
        """:210-253.  Same keys / values; all log scalars are averaged over ranks by ONE
        all-reduce and fetched with ONE device->host copy (the reference: one blocking
        all-reduce + .item() per key)."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = _scalar_mean(value)
            elif isinstance(value, list):
                log_vars[name] = sum(_scalar_mean(v) for v in value)
            else:
                raise TypeError(f"{name} is not a tensor or list of tensors")
        loss = _sum_scalars([v for k, v in log_vars.items() if "loss" in k])
        log_vars["loss"] = loss
        keys = list(log_vars.keys())
        flat = allreduce_scalars([torch.full((), float(len(keys)), device=loss.device)] + [log_vars[k] for k in keys])
        if self.lazy_log:
            return loss, OrderedDict(_keys=keys, _flat=flat)
        host = flat.cpu().tolist()
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        assert abs(host[0] - len(keys)) < 1e-6, "loss log variables are different across GPUs!\n" + ",".join(keys)
        del world
        return loss, OrderedDict((k, v) for k, v in zip(keys, host[1:]))

    def parse_losses_local(self, losses):
        """(loss, keys, flat) with flat = [n_keys, values...] on the device and NO collective and
        no host copy: the graphed step reduces ``flat`` across ranks outside the captured region."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            log_vars[name] = _scalar_mean(value) if isinstance(value, torch.Tensor) else sum(_scalar_mean(v) for v in value)
        loss = _sum_scalars([v for k, v in log_vars.items() if "loss" in k])
        log_vars["loss"] = loss
        keys = list(log_vars.keys())
        flat = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        return loss, keys, flat

    def train_step(self, data, optimizer=None):
        """:419-450."""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["img_metas"]))

    val_step = train_step
