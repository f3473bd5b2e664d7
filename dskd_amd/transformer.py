"""Deformable-DETR transformer with the reference's config surface.

Structure follows /root/reference/mmdet/models/utils/transformer.py
(``DeformableDetrTransformer`` :712-1055, ``DeformableDetrTransformerDecoder`` :624-709,
``inverse_sigmoid`` :388-404) and the ext-mmcv building blocks it configures
(``BaseTransformerLayer``, ``FFN``, ``MultiheadAttention``, ``MultiScaleDeformableAttention``;
mmcv-full 1.3.17..1.6.2, not in the reference tree -- behaviour restated from SURVEY.md
section 3.3).  Parameter names match mmcv's so reference checkpoints load.

MI355X notes: activations stay batch-first ``[B, N, C]`` inside (the reference's
``(N, B, C)`` layout and its permutes exist only at the module boundary as views); the
sampling+aggregation core is the HIP kernel behind ``native.ms_deform_attn``; all dense
projections are plain ``nn.Linear`` -> hipBLASLt/MFMA (bf16 under autocast).
"""
import copy
import math
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .dist import grad_slot
from .utils import device_const
from .builder import (ATTENTION, FEEDFORWARD_NETWORK, POSITIONAL_ENCODING, TRANSFORMER, TRANSFORMER_LAYER,
                      TRANSFORMER_LAYER_SEQUENCE, build_attention, build_feedforward_network,
                      build_transformer_layer, build_transformer_layer_sequence)


_ATTN_KERNEL = not os.environ.get("DSKD_SDPA_ATTN")      # A/B switch: PyTorch's scaled_dot_product_attention for the decoder's queries
_CHUNK_CACHE = {}


def _token_chunk(T, out_elems=65536, lo=256, hi=4096):
    """Largest divisor of T in [lo, hi] that still leaves enough chunks to fill the chip
    (measured on MI355X: 256x256 outputs want >= 128 batches, 1024x256 ones >= 64)."""
    min_batches = 128 if out_elems <= 131072 else 64
    key = (T, min_batches)
    if key not in _CHUNK_CACHE:
        best = None
        for c in range(lo, min(hi, T) + 1):
            if T % c == 0 and T // c >= min_batches:
                best = c
        if best is None:                      # fewer, larger chunks are still better than one GEMM
            for c in range(lo, min(hi, T) + 1):
                if T % c == 0 and T // c >= 16:
                    best = c
        _CHUNK_CACHE[key] = best
    return _CHUNK_CACHE[key]


_ONES = {}


def rowsum(t2d):
    """Column sums of a [rows, C] CUDA tensor as a GEMM with a row of ones (result in t2d's dtype, f32 accumulation).
    ATen's ``sum(0)`` splits a tall reduction over several workgroups and resets their semaphore with
    hipMemsetAsync; captured into a hipGraph that memset replays with a garbage value on this ROCm runtime
    (csrc/common.h), so bias gradients inside a replayed region must not come from it."""
    key = (t2d.shape[0], t2d.dtype, t2d.device)
    ones = _ONES.get(key)
    if ones is None:
        if len(_ONES) > 64:                  # captured graphs pin the rows they read (head._forward_graphed's keepalive)
            _ONES.pop(next(iter(_ONES)))
        ones = _ONES[key] = torch.ones((1, t2d.shape[0]), dtype=t2d.dtype, device=t2d.device)
    return torch.mm(ones, t2d).view(-1)


def bias_grad(g2):
    """d(bias) of a Linear from the [rows, C] output gradient: the library's column-sum kernel where it applies, else
    the ones-GEMM; ATen's reduction only on the CPU."""
    if not g2.is_cuda:
        return g2.sum(0, dtype=torch.float32).to(g2.dtype)
    if g2.dtype == torch.bfloat16 and g2.shape[-1] in native.COLSUM_WIDTHS and g2.is_contiguous() and g2.shape[0] >= 4096:
        return native.colsum(g2, out_dtype=g2.dtype)
    if native.colsum_short_ok(g2):
        return native.colsum_short(g2)          # the decoder's / head branches' short inputs: one launch, no library GEMM
    return rowsum(g2.contiguous())


def _sum_partials(part, dtype):
    """Sum of the split-K partial products over their leading dimension in ``dtype``.  For bf16 partials and a bf16 result
    ATen's reduction already accumulates in f32 and rounds once, so ``sum(0)`` equals ``sum(0, dtype=f32).to(bf16)`` without
    the extra cast launch (36 of them per step)."""
    if part.dtype == dtype and dtype in (torch.bfloat16, torch.float16):
        return part.sum(0)
    return part.sum(0, dtype=torch.float32).to(dtype)


def _weight_grad(g2, x2, chunk, dtype):
    """dW = g2^T x2 of a tall Linear ([tokens, N]^T [tokens, K]): the hand-written split-K MFMA kernel with transposing
    LDS reads where it applies (native.gemm_tn: one launch, f32 result), else the library (token chunks as the batch of
    one bmm + an f32 sum of the partial products, or one GEMM)."""
    if native.gemm_tn_ok(g2, x2):
        return native.gemm_tn_bf16(g2, x2) if dtype == torch.bfloat16 else native.gemm_tn(g2, x2).to(dtype)
    if chunk is None and g2.is_cuda and g2.shape[0] >= 2048 and g2.is_contiguous() and x2.is_contiguous():
        # mid-size inputs with an output the split-K kernel does not take (the head's class / box branches: [7 200, 80 | 68 | 4]^T
        # [7 200, 256]): as ONE GEMM the library runs them on a handful of workgroups -- 46-73 us of kernel time against 16-22
        # for token chunks as bmm batches + the sum of the partial products (tools/prof/cls_dw_bench.py)
        chunk = _token_chunk(g2.shape[0], g2.shape[1] * x2.shape[1])
    if chunk is None:
        return (g2.t() @ x2).to(dtype)
    nb = x2.shape[0] // chunk
    part = torch.bmm(g2.view(nb, chunk, -1).transpose(1, 2), x2.view(nb, chunk, -1))
    return _sum_partials(part, dtype)


class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for a very tall x (tens of thousands of tokens, 256..1024 features).
    Forward and dX are ordinary GEMMs.  dW = dY^T X has a tiny output (<= 1024 x 256) and a
    huge reduction dimension (88 892 tokens at B=4): as one GEMM it runs on a handful of
    workgroups (hipBLASLt: ~55 TFLOP/s measured).  Here the token axis is cut into chunks that
    become the batch dimension of one bmm (hundreds of workgroups), followed by an fp32 sum of
    the partial products -- a split-K GEMM expressed through the library."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunk, relu):
        ctx.chunk, ctx.relu = chunk, relu
        ctx.has_bias = bias is not None
        assert x.dim() == 2           # the caller flattens: the output must not be a view
        if native.lin256_ok(x, weight.shape[0], weight.shape[1]) and weight.dtype == x.dtype and weight.is_contiguous():
            # 256 inputs, tall: the hand-written MFMA kernel (memory-bound; hipBLASLt takes ~1.7x as long)
            y = native.lin256(x, native.lin256_pack(weight), weight.shape[0], bias, relu)
        elif native.gemm_nt_2d_ok(x, weight, bias):
            y = native.gemm_nt_2d(x, weight, bias, relu)        # short inputs (decoder, head branches): own MFMA GEMM
        elif relu and bias is not None and x.is_cuda:
            y = torch._addmm_activation(bias, x, weight.t())     # bias + ReLU in the GEMM epilogue
        else:
            y = F.linear(x, weight, bias)
            if relu:
                y = torch.relu_(y)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if weight.shape[0] == 256 and native.lin256_ok(g2, weight.shape[1], weight.shape[0]) and \
                    weight.dtype == g2.dtype and weight.is_contiguous():
                gx = native.lin256(g2, native.lin256_pack(weight, transposed=True), weight.shape[1]).view(x.shape)
            else:
                gx = (g2 @ weight).view(x.shape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_b and weight.dtype == torch.bfloat16 and native.gemm_tn_ok(g2, x2):
            gw, gb = native.gemm_tn_bf16(g2, x2, want_bias=True)      # the bias gradient rides in the dW launch pair
            return gx, gw, gb, None, None
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(g2, x2, ctx.chunk, weight.dtype)
        if want_b:
            gb = bias_grad(g2)
        return gx, gw, gb, None, None


class _FFNInnerFn(torch.autograd.Function):
    """``dropout_p(relu(x W^T + b))`` of the FFN for a very tall bf16 ``x`` on the GPU: bias + ReLU
    in the GEMM epilogue, dropout in place without a stored mask, and ONE backward pass that
    applies dropout + ReLU backward and yields the bias gradient (native.relu_dropout_bwd);
    dX is a GEMM, dW the split-K bmm of :class:`_TallLinearFn`."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunk, p):
        ctx.chunk, ctx.p = chunk, p
        y = torch._addmm_activation(bias, x, weight.t())
        native.dropout_(y, p)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        cdt = weight.dtype if weight.dtype in (torch.float32, torch.bfloat16) else torch.float32
        g1, colsum = native.relu_dropout_bwd(g, y, ctx.p, want_colsum=ctx.needs_input_grad[2], colsum_dtype=cdt)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g1 @ weight
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(g1, x, ctx.chunk, weight.dtype)
        if colsum is not None:
            gb = colsum.to(g.dtype)
        return gx, gw, gb, None, None


class _FusedFFNFn(torch.autograd.Function):
    """``dropout_p(relu(x W1^T + b1)) W2^T + b2`` of a very tall bf16 ``x`` as ONE hand-written MFMA kernel per
    direction (csrc/ffn_mfma.hip): the 1024-wide hidden activation stays on chip between the two GEMMs; it leaves
    once (H, for the backward) and its gradient once (for dW1 / db1).  Weight gradients stay split-K library GEMMs
    over token chunks (see :class:`_TallLinearFn`), bias gradients streaming column sums."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, chunk):
        need_bwd = any(ctx.needs_input_grad[:5])
        pf, pb = native.ffn_pack(w1, w2, want_bwd=need_bwd)
        y, h = native.ffn_fwd_raw(x, pf, b1, b2, p, store_h=need_bwd, hidden=w1.shape[0])
        if need_bwd:
            ctx.save_for_backward(x, h, pb)
            ctx.p, ctx.chunk, ctx.dt = p, chunk, w1.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, h, pb = ctx.saved_tensors
        gy = gy.contiguous()
        cdt = ctx.dt if ctx.dt in (torch.float32, torch.bfloat16) else torch.float32
        gh, gx, cs = native.ffn_bwd_raw(gy, h, pb, ctx.p, want_colsum=True, colsum_dtype=cdt)
        gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[1]:
            gw1 = _weight_grad(gh, x, ctx.chunk, ctx.dt)
        if ctx.needs_input_grad[2]:
            gb1 = cs.to(ctx.dt)
        if ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and ctx.dt == torch.bfloat16 and native.gemm_tn_ok(gy, h):
            gw2, gb2 = native.gemm_tn_bf16(gy, h, want_bias=True)       # d(b2) rides in the dW2 launch pair
        else:
            if ctx.needs_input_grad[3]:
                gw2 = _weight_grad(gy, h, ctx.chunk, ctx.dt)
            if ctx.needs_input_grad[4]:
                gb2 = native.colsum(gy, out_dtype=ctx.dt) if ctx.dt in (torch.float32, torch.bfloat16) else \
                    native.colsum(gy).to(ctx.dt)
        return (gx if ctx.needs_input_grad[0] else None), gw1, gb1, gw2, gb2, None, None


def _frozen_packed(owner, w1, w2):
    """Fragment-order image of FROZEN bf16 FFN weights (the teacher's), packed once per weight version and kept ON the
    owning module (a global keyed by data_ptr could hand out another model's image when the allocator reuses an address)."""
    ver = (w1.data_ptr(), w2.data_ptr(), w1._version, w2._version)
    hit = owner.__dict__.get("_ffn_packed")
    if hit is None or hit[0] != ver:
        hit = owner.__dict__["_ffn_packed"] = (ver, native.ffn_pack(w1, w2, want_bwd=False)[0], w1, w2)
    return hit[1]


FFN_FUSED = True      # tests set this to False to get the library GEMM chain (the control of the fused kernel's parity tests)


def ffn_fused_ok(x, w1, w2, b1, b2):
    """Can csrc/ffn_mfma.hip take this FFN (tall bf16 CUDA tokens, d_model 256 / hidden 1024, both biases)?"""
    dev = x.device.type
    if not x.is_cuda or not FFN_FUSED or b1 is None or b2 is None:
        return False
    dtype = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x.dtype
    tokens = x.numel() // max(x.shape[-1], 1)
    return dtype == torch.bfloat16 and tokens >= 16384 and (w1.shape[1], w1.shape[0]) == native.FFN_FUSED_DIMS \
        and tuple(w2.shape) == (w1.shape[1], w1.shape[0]) and x.is_contiguous()


def ffn_fused(x, w1, b1, w2, b2, p, owner=None):
    """The two Linears of the FFN with ReLU + Dropout(p) between them, fused (see :class:`_FusedFFNFn`).  ``owner``: the
    module that keeps the packed image of frozen weights."""
    dev = x.device.type
    tokens = x.numel() // x.shape[-1]
    bf = torch.bfloat16
    x2 = x.reshape(tokens, x.shape[-1]).to(bf)
    if torch.is_grad_enabled() and (w1.requires_grad or x.requires_grad):
        chunk = _token_chunk(tokens, w1.numel())
        if chunk is None:
            return None
        with torch.autocast(dev, enabled=False):
            y = _FusedFFNFn.apply(x2, w1.to(bf).contiguous(), b1.to(bf), w2.to(bf).contiguous(), b2.to(bf), float(p), chunk)
    else:
        w1b, w2b = w1.detach().to(bf).contiguous(), w2.detach().to(bf).contiguous()
        pf = _frozen_packed(owner, w1b, w2b) if (owner is not None and not w1.requires_grad) \
            else native.ffn_pack(w1b, w2b, want_bwd=False)[0]
        y, _ = native.ffn_fwd_raw(x2, pf, b1.detach().to(bf), b2.detach().to(bf), 0.0 if not p else float(p), store_h=bool(p))
    return y.view(*x.shape[:-1], y.shape[-1])


def ffn_inner(x, weight, bias, p):
    """Linear + ReLU + Dropout(p) of the FFN.  Tall bf16 GPU inputs that need gradients take
    :class:`_FFNInnerFn`; everything else is ``tall_linear(relu=True)`` + ``F.dropout``."""
    dev = x.device.type
    tokens = x.numel() // max(x.shape[-1], 1)
    dtype = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x.dtype
    if x.is_cuda and dtype == torch.bfloat16 and tokens >= 16384 and weight.requires_grad and bias is not None \
            and torch.is_grad_enabled() and x.is_contiguous() and weight.shape[0] in (256, 512, 1024, 2048):
        chunk = _token_chunk(tokens, weight.numel())
        if chunk is not None:
            x2 = x.reshape(tokens, x.shape[-1]).to(dtype)
            with torch.autocast(dev, enabled=False):
                y = _FFNInnerFn.apply(x2, weight.to(dtype), bias.to(dtype), chunk, float(p))
            return y.view(*x.shape[:-1], y.shape[-1])
    y = tall_linear(x, weight, bias, relu=True)
    return F.dropout(y, p, training=True) if p > 0 else y


class _CastParams(torch.autograd.Function):
    """fp32 master parameters -> compute dtype in ONE multi-tensor launch, and their gradients
    back in one.  torch.autocast does the same per parameter: ~200 cast launches forward and as
    many ``ToCopyBackward`` launches backward for the student head, each a few microseconds of
    GPU time and ~10 us of host time."""

    @staticmethod
    def forward(ctx, dtype, static, *params):
        """``static``: persistent ``dtype`` buffers to cast into (their storage is then the same on every step, which
        is what lets a captured hipGraph read the step's parameters), or None for fresh tensors."""
        ctx.set_materialize_grads(False)
        ctx.pdtypes = [p.dtype for p in params]
        ctx.pids = [id(p) for p in params]
        outs = [torch.empty_like(p, dtype=dtype) for p in params] if static is None else static
        srcs = [p.detach() for p in params]
        # r4: ONE launch of dskd_cast_scale_many (its table lives with the persistent buffers); ATen's multi-tensor copy
        # otherwise
        cache = _CAST_TABLES.setdefault(id(static), (native.MultiCast(0), native.MultiCast(1))) if static is not None else None
        ctx.cache = cache
        if cache is not None and dtype == torch.bfloat16 and cache[0].ready(srcs, outs, [None] * len(srcs)):
            cache[0].run(srcs, outs, [None] * len(srcs))
        else:
            torch._foreach_copy_(outs, srcs)
        return tuple(outs) if static is None else tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None]
        # data parallel: straight into the parameter's slot of the flat gradient buffer (dist.GradSync), else a fresh tensor
        slots = [grad_slot(ctx.pids[i], grads[i].shape, ctx.pdtypes[i]) for i in idx]
        ups = [s if s is not None else torch.empty_like(grads[i], dtype=ctx.pdtypes[i]) for s, i in zip(slots, idx)]
        if idx:
            gs = [grads[i] for i in idx]
            if ctx.cache is not None and ctx.cache[1].ready(gs, ups, [None] * len(gs)):
                ctx.cache[1].run(gs, ups, [None] * len(gs))
            else:
                torch._foreach_copy_(ups, gs)
        out = [None] * len(grads)
        for j, i in enumerate(idx):
            out[i] = ups[j]
        return (None, None, *out)


_CAST_TABLES = {}      # id(list of persistent low-precision buffers) -> (forward, backward) MultiCast tables


class _AddLevelEmbed(torch.autograd.Function):
    """``pos + level_embed`` for one level ([B, HW, C] + [C]) whose d(level_embed) is a ones-GEMM instead of ATen's
    multi-workgroup reduction over B * HW rows (see rowsum)."""

    @staticmethod
    def forward(ctx, pos, emb):
        dev = pos.device.type
        # under bf16 autocast the table reaches the encoder in bf16 (DetrTransformerEncoder._forward_fused), so the gradient
        # that comes back here in f32 holds bf16 values: the column sums may read them as bf16 (half the bytes, own kernel)
        ctx.lowp = pos.is_cuda and torch.is_autocast_enabled(dev) and torch.get_autocast_dtype(dev) == torch.bfloat16
        return pos + emb.view(1, 1, -1)

    @staticmethod
    def backward(ctx, g):
        ge = None
        if ctx.needs_input_grad[1]:
            if g.is_cuda and ctx.lowp and g.dtype == torch.float32 and g.shape[-1] in native.COLSUM_WIDTHS:
                # one strided read into a contiguous bf16 copy + the streaming column sum: ~40 us for the finest level where
                # contiguous() + an f32 ones-row GEMM through the library took ~170 (132 us for the GEMM alone)
                gb = g.to(torch.bfloat16).reshape(-1, g.shape[-1])
                ge = native.colsum(gb if gb.is_contiguous() else gb.contiguous(), out_dtype=torch.float32)
            else:
                g2 = g.reshape(-1, g.shape[-1])
                ge = rowsum(g2.contiguous()) if g2.is_cuda else g2.sum(0)
        return (g if ctx.needs_input_grad[0] else None), ge


class _JoinRows(torch.autograd.Function):
    """``torch.cat([a, b], 0)`` for two tensors that already lie behind each other in ONE buffer (lowp_params allocates the
    low-precision copies of sampling_offsets / attention_weights that way): the result is a view, the backward two views."""

    @staticmethod
    def adjacent(a, b):
        return (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:]
                and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and a.storage_offset() + a.numel() == b.storage_offset())

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.shape[0]
        return a.as_strided((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:]


class _SplitRows(torch.autograd.Function):
    """``(w[:n], w[n:])`` whose backward is ONE concatenation.  Autograd's own slice backward makes a zero tensor of the
    whole shape per slice, copies the slice's gradient in and adds the two: five launches per parameter where one does (the
    in_proj weight and bias of the decoder's MultiheadAttention: 60 launches per step)."""

    @staticmethod
    def forward(ctx, w, n):
        ctx.n, ctx.shape = n, w.shape
        return w[:n], w[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros((ctx.n,) + tuple(ctx.shape[1:]))
        if gb is None:
            gb = ref.new_zeros((ctx.shape[0] - ctx.n,) + tuple(ctx.shape[1:]))
        return torch.cat([ga, gb], 0), None


def join_rows(a, b):
    """cat([a, b], 0) -- as a view when the two already are neighbours in memory (see :class:`_JoinRows`)."""
    if _JoinRows.adjacent(a, b):
        return _JoinRows.apply(a, b)
    return torch.cat([a, b], 0)


class lowp_params:
    """``with lowp_params(root, dtype):`` -- every trainable :class:`Linear` below ``root`` uses
    a ``dtype`` copy of its parameters made by ONE :class:`_CastParams` call (autograd routes
    the gradients back to the fp32 masters).  Frozen modules keep their cached copies."""

    def __init__(self, root, dtype):
        self.root, self.dtype = root, dtype

    def __enter__(self):
        mods = self.root.__dict__.get("_lowp_mods")
        if mods is None:
            mods = [m for m in self.root.modules() if isinstance(m, Linear)]
            self.root.__dict__["_lowp_mods"] = mods
        self.live = [m for m in mods if m.weight.requires_grad and (m.bias is None or m.bias.requires_grad)]
        mhas = self.root.__dict__.get("_lowp_mhas")
        if mhas is None:
            mhas = [m for m in self.root.modules() if isinstance(m, MultiheadAttention)]
            self.root.__dict__["_lowp_mhas"] = mhas
        self.live_mha = [m for m in mhas if m.attn.in_proj_weight is not None and m.attn.in_proj_weight.requires_grad
                         and m.attn.in_proj_bias is not None and m.attn.out_proj.bias is not None]
        params = []
        for m in self.live:
            params.append(m.weight)
            if m.bias is not None:
                params.append(m.bias)
        for m in self.live_mha:
            params += [m.attn.in_proj_weight, m.attn.in_proj_bias, m.attn.out_proj.weight, m.attn.out_proj.bias]
        self.params, self.outs = params, []
        if params:
            key = tuple((p.data_ptr(), tuple(p.shape)) for p in params) + (self.dtype,)
            static = self.root.__dict__.get("_lp_static")
            if static is None or static[0] != key:
                bufs = [None] * len(params)
                # sampling_offsets | attention_weights of one MultiScaleDeformableAttention are used as ONE GEMM on the
                # concatenated weights: their copies lie behind each other, so that the concatenation is a view (join_rows)
                index = {id(p): i for i, p in enumerate(params)}
                for m in self.root.modules():
                    if isinstance(m, MultiScaleDeformableAttention):
                        for pa, pb in ((m.sampling_offsets.weight, m.attention_weights.weight),
                                       (m.sampling_offsets.bias, m.attention_weights.bias)):
                            ia, ib = index.get(id(pa)), index.get(id(pb))
                            if ia is None or ib is None or bufs[ia] is not None or bufs[ib] is not None \
                                    or not (pa.is_contiguous() and pb.is_contiguous()) or pa.shape[1:] != pb.shape[1:]:
                                continue
                            joint = torch.empty((pa.shape[0] + pb.shape[0],) + tuple(pa.shape[1:]), dtype=self.dtype,
                                                device=pa.device)
                            bufs[ia], bufs[ib] = joint[:pa.shape[0]], joint[pa.shape[0]:]
                static = (key, [b if b is not None else torch.empty_like(p, dtype=self.dtype) for b, p in zip(bufs, params)])
                self.root.__dict__["_lp_static"] = static
                # fragment-order images of the tall 256-input weights (and of the joint so | aw buffers), refreshed by one
                # launch per step right after the cast below (native.Lin256Prepack)
                old = self.root.__dict__.pop("_lp_prepack", None)
                if old is not None:
                    old.drop()
                if self.dtype == torch.bfloat16 and params[0].is_cuda and not torch.cuda.is_current_stream_capturing():
                    joints, seen = [], set()
                    for b in bufs:
                        if b is not None and b.dim() == 2 and b.untyped_storage().data_ptr() not in seen:
                            seen.add(b.untyped_storage().data_ptr())
                            base = b._base if b._base is not None else b
                            joints.append(base)
                    epoch = self.root.__dict__.setdefault("_lp_epoch", [0])
                    self.root.__dict__["_lp_prepack"] = native.Lin256Prepack(
                        [t for t in static[1] if t.dim() == 2 and t._base is None] + joints, epoch)
            epoch = self.root.__dict__.setdefault("_lp_epoch", [0])
            epoch[0] += 1                       # the copies are about to be rewritten: images made before are stale
            self.outs = list(_CastParams.apply(self.dtype, static[1], *params))
            pre = self.root.__dict__.get("_lp_prepack")
            if pre is not None:
                pre.refresh()
            outs = iter(self.outs)
            for m in self.live:
                m.__dict__["_live_lp"] = (next(outs), next(outs) if m.bias is not None else None)
            for m in self.live_mha:
                m.__dict__["_live_lp"] = (next(outs), next(outs), next(outs), next(outs))
        return self

    def __exit__(self, *exc):
        for m in self.live + self.live_mha:
            m.__dict__.pop("_live_lp", None)
        return False

    def install(self, tensors):
        """Make the modules read ``tensors`` (same order and shapes as ``self.outs``) as their low-precision
        parameters; returns the previous assignment (a list to pass back here)."""
        prev = [m.__dict__.get("_live_lp") for m in self.live + self.live_mha]
        it = iter(tensors)
        for m in self.live:
            m.__dict__["_live_lp"] = (next(it), next(it) if m.bias is not None else None)
        for m in self.live_mha:
            m.__dict__["_live_lp"] = (next(it), next(it), next(it), next(it))
        return [t for pr in prev for t in (pr if pr is not None else ()) if t is not None]


class Linear(nn.Linear):
    """nn.Linear that keeps a low-precision copy of FROZEN parameters under autocast.
    torch.autocast caches weight casts only for leaf tensors that require grad, so the frozen
    teacher (and any frozen student layer) would re-cast every weight on every call; trainable
    parameters use the step's :class:`lowp_params` copies when a caller provides them, else
    the stock path (autocast's own per-step cache)."""

    def lp(self):
        """(weight, bias) to compute with: the live low-precision copies inside ``lowp_params``,
        else the parameters themselves."""
        live = self.__dict__.get("_live_lp")
        return live if live is not None else (self.weight, self.bias)

    def frozen_lp(self, dev):
        """(weight, bias) in the autocast dtype when both are frozen and autocast is on, else None."""
        if torch.is_autocast_enabled(dev) and not self.weight.requires_grad and \
                (self.bias is None or not self.bias.requires_grad):
            dtype = torch.get_autocast_dtype(dev)
            key = (self.weight._version, -1 if self.bias is None else self.bias._version, dtype, self.weight.device)
            cache = self.__dict__.get("_lp")
            if cache is None or cache[0] != key:
                cache = (key, self.weight.detach().to(dtype), None if self.bias is None else self.bias.detach().to(dtype))
                self.__dict__["_lp"] = cache
            return cache[1], cache[2]
        return None

    def forward(self, x):
        lp = self.frozen_lp(x.device.type)
        if lp is not None:
            return tall_linear(x, lp[0], lp[1])       # frozen: F.linear, or the packed MFMA kernel for tall 256-wide inputs
        w, b = self.lp()
        return tall_linear(x, w, b)


def tall_linear(x, weight, bias, relu=False):
    """F.linear (optionally + ReLU), switching to the split-K weight gradient -- and the fused
    bias+ReLU GEMM epilogue -- for very tall inputs."""
    dev = x.device.type
    tokens = x.numel() // max(x.shape[-1], 1)
    if x.is_cuda and weight.requires_grad and torch.is_grad_enabled() and tokens > 0:
        # own backward on the GPU for every size: split-K dW for very tall inputs, and bias gradients that never go
        # through ATen's multi-workgroup reduction (see rowsum)
        chunk = _token_chunk(tokens, weight.numel()) if (tokens >= 16384 and x.is_contiguous()) else None
        lead = x.shape[:-1]
        x2 = x.reshape(tokens, x.shape[-1])
        if torch.is_autocast_enabled(dev):       # what autocast would do for F.linear
            dtype = torch.get_autocast_dtype(dev)
            x2, weight = x2.to(dtype), weight.to(dtype)
            bias = None if bias is None else bias.to(dtype)
            with torch.autocast(dev, enabled=False):
                y = _TallLinearFn.apply(x2, weight, bias, chunk, relu)
        else:
            if x2.dtype != weight.dtype:
                x2 = x2.to(weight.dtype)
            y = _TallLinearFn.apply(x2, weight, bias, chunk, relu)
        return y.view(*lead, y.shape[-1])
    if x.is_cuda and not weight.requires_grad and not (torch.is_grad_enabled() and x.requires_grad) and \
            weight.dtype == torch.bfloat16 and weight.dim() == 2 and weight.is_contiguous():
        x2 = x.reshape(tokens, x.shape[-1]) if tokens > 0 else None
        if x2 is not None and x2.dtype != weight.dtype and torch.is_autocast_enabled(dev):
            x2 = x2.to(weight.dtype)
        if x2 is not None and native.lin256_ok(x2, weight.shape[0], weight.shape[1]):
            # frozen weights (the teacher): packed once per weight version, kept on the weight tensor itself
            hit = weight.__dict__.get("_dskd_lin256") if hasattr(weight, "__dict__") else None
            if hit is None or hit[0] != weight._version:
                hit = (weight._version, native.lin256_pack(weight))
                try:
                    weight._dskd_lin256 = hit
                except Exception:
                    pass
            return native.lin256(x2, hit[1], weight.shape[0], bias, relu).view(*x.shape[:-1], weight.shape[0])
        if x2 is not None and x2.dtype == weight.dtype and x2.is_contiguous() and native.gemm_nt_2d_ok(x2, weight, bias):
            return native.gemm_nt_2d(x2, weight, bias, relu).view(*x.shape[:-1], weight.shape[0])
    y = F.linear(x, weight, bias)
    return torch.relu_(y) if relu else y


def inverse_sigmoid(x, eps=1e-5):
    """transformer.py:388-404."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding(nn.Module):
    """/root/reference/mmdet/models/utils/positional_encoding.py:11-100."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0.,
                 init_cfg=None):
        super().__init__()
        if normalize:
            assert isinstance(scale, (float, int))
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


@ATTENTION.register_module()
class MultiScaleDeformableAttention(nn.Module):
    """ext-mmcv ``MultiScaleDeformableAttention``.  ``forward`` takes
    and returns ``(num_query, bs, embed_dims)`` tensors unless ``batch_first``."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.output_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid_init.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, tokens_batch_first=None,
                value_batch_first=None, fuse_tail=False, **kwargs):
        """``tokens_batch_first`` / ``value_batch_first`` (set by DeformableDetrTransformer)
        override the module's ``batch_first`` for query/output and for ``value``: the encoder
        runs batch-first end to end so that no [B, 22k, 256] tensor is ever permuted+copied.
        ``fuse_tail``: return the projected output WITHOUT ``dropout(.) + identity`` (the caller
        folds them into the following LayerNorm launch)."""
        q_bf = self.batch_first if tokens_batch_first is None else tokens_batch_first
        v_bf = q_bf if value_batch_first is None else value_batch_first
        if value is None:
            value = query
            v_bf = q_bf
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not q_bf:
            query = query.permute(1, 0, 2)
        if not v_bf:
            value = value.permute(1, 0, 2)
        output = self.core(query, value, reference_points, spatial_shapes, key_padding_mask)
        if not q_bf:
            output = output.permute(1, 0, 2)
        if fuse_tail:
            return output
        return self.dropout(output) + identity

    def tail_dropout_p(self):
        return self.dropout.p if self.training else 0.0

    def core(self, query, value, reference_points, spatial_shapes, key_padding_mask=None):
        """Batch-first body of ``forward``: ``query`` already carries its positional encoding;
        returns ``output_proj(sampling(...))`` WITHOUT the module's dropout + identity (the
        fused encoder path folds those into the following LayerNorm launch)."""
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        shapes = spatial_shapes.tolist() if isinstance(spatial_shapes, torch.Tensor) else list(spatial_shapes)
        assert sum(h * w for h, w in shapes) == num_value

        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        # sampling offsets and attention logits: one GEMM on the concatenated weights (the query
        # is read and cast once instead of twice); parameters keep their own names.
        n_off = self.sampling_offsets.out_features
        so, aw = self.sampling_offsets, self.attention_weights
        frozen = not (so.weight.requires_grad or aw.weight.requires_grad or so.bias.requires_grad or aw.bias.requires_grad)
        dev = query.device.type
        if frozen:      # teacher: concatenate (and cast, under autocast) once
            dtype = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else so.weight.dtype
            key = (so.weight._version, aw.weight._version, so.bias._version, aw.bias._version, dtype, so.weight.device)
            if getattr(self, "_cat_key", None) != key:
                self._cat_key = key
                self._cat = (torch.cat([so.weight, aw.weight], 0).detach().to(dtype),
                             torch.cat([so.bias, aw.bias], 0).detach().to(dtype))
            w_cat, b_cat = self._cat
        else:
            (sw, sb), (ww, wb) = so.lp(), aw.lp()
            w_cat = join_rows(sw, ww)
            b_cat = join_rows(sb, wb)
        both = tall_linear(query, w_cat, b_cat)
        if value.is_cuda and reference_points.shape[-1] == 2 and self.num_levels * self.num_points == 16 \
                and self.num_levels <= 4 and both.dtype == value.dtype and not (
                    torch.is_grad_enabled() and (both.requires_grad or value.requires_grad or reference_points.requires_grad)):
            # no gradients (frozen teacher, inference): prologue folded into the sampling kernel
            output = native.ms_deform_attn_fused(value, shapes, both, reference_points, self.num_levels, self.num_points)
            return self.output_proj(output)
        if reference_points.shape[-1] == 2 and self.num_levels * self.num_points == 16 and self.num_levels <= 4:
            # softmax + location arithmetic in one HIP pass each way (encoder and decoder)
            sampling_locations, attention_weights = native.msda_prepare(
                both, reference_points, shapes, self.num_heads, self.num_levels, self.num_points)
        else:
            sampling_offsets = both[..., :n_off].float().view(
                bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
            attention_weights = both[..., n_off:].float().view(
                bs, num_query, self.num_heads, self.num_levels * self.num_points)
            attention_weights = attention_weights.softmax(-1).view(
                bs, num_query, self.num_heads, self.num_levels, self.num_points)
            if reference_points.shape[-1] == 2:
                key = (tuple(map(tuple, shapes)), sampling_offsets.device)
                if getattr(self, "_norm_key", None) != key:      # constant per geometry: build once
                    self._norm_key, self._norm = key, sampling_offsets.new_tensor([[w, h] for h, w in shapes])
                normalizer = self._norm
                sampling_locations = reference_points[:, :, None, :, None, :].float() \
                    + sampling_offsets / normalizer[None, None, None, :, None, :]
            elif reference_points.shape[-1] == 4:
                sampling_locations = reference_points[:, :, None, :, None, :2] \
                    + sampling_offsets / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5
            else:
                raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]}")
        output = native.ms_deform_attn(value, shapes, sampling_locations, attention_weights)
        return self.output_proj(output)


@ATTENTION.register_module()
class MultiheadAttention(nn.Module):
    """ext-mmcv wrapper around ``nn.MultiheadAttention`` with identity + dropout; the
    deprecated ``dropout`` kwarg sets both attn_drop and the dropout layer (as mmcv does)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=dict(type="Dropout", drop_prob=0.),
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        dropout_layer = dict(dropout_layer) if dropout_layer else None
        if "dropout" in kwargs:
            warnings.warn("The arguments `dropout` in MultiheadAttention has been deprecated", DeprecationWarning)
            attn_drop = kwargs["dropout"]
            dropout_layer["drop_prob"] = kwargs.pop("dropout")
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer["drop_prob"]) if dropout_layer else nn.Identity()

    def _attend(self, query, key, value, batch_first=False):
        """``nn.MultiheadAttention`` without masks on the GPU, same arithmetic with fewer launches:
        q and k projected by ONE GEMM when they are the same tensor (DETR self-attention: q = k =
        query + query_pos, v = query), the attention core in csrc/attn.hip (``scaled_dot_product_attention`` for other shapes), the step's
        low-precision parameter copies (``lowp_params``) when present.  [L, B, E] in and out."""
        E, H = self.embed_dims, self.num_heads
        live = self.__dict__.get("_live_lp")
        w, b, wo, bo = live if live is not None else (self.attn.in_proj_weight, self.attn.in_proj_bias,
                                                      self.attn.out_proj.weight, self.attn.out_proj.bias)
        p_drop = self.attn.dropout if self.training else 0.0
        (w_qk, w_v), (b_qk, b_v) = _SplitRows.apply(w, 2 * E), _SplitRows.apply(b, 2 * E)
        v = tall_linear(value, w_v, b_v)
        if query is key:
            qk = tall_linear(query, w_qk, b_qk)
            if _ATTN_KERNEL and E == H * native.ATTN_HEAD_DIM and query.shape[1 if batch_first else 0] <= native.ATTN_MAX_TOKENS \
                    and native.self_attention_ok(qk, v, H):
                # own kernels (csrc/attn.hip): q | k, v and the result stay where the projections wrote / read them
                return tall_linear(native.self_attention(qk, v, H, p_drop, batch_first=batch_first), wo, bo)
            q, k = qk.split(E, dim=-1)
        else:
            q, k = tall_linear(query, w_qk[:E], b_qk[:E]), tall_linear(key, w_qk[E:], b_qk[E:])
        if batch_first:         # [B, L, E] tokens: the head split is a view either way, nothing is permuted + copied on the way in
            B, L, _ = query.shape
            S = key.shape[1]
            q = q.reshape(B, L, H, E // H).permute(0, 2, 1, 3)
            k = k.reshape(B, S, H, E // H).permute(0, 2, 1, 3)
            v = v.reshape(B, S, H, E // H).permute(0, 2, 1, 3)
            out = F.scaled_dot_product_attention(q, k, v, dropout_p=p_drop)
            return tall_linear(out.permute(0, 2, 1, 3).reshape(B, L, E), wo, bo)
        L, B, _ = query.shape
        S = key.shape[0]
        q = q.reshape(L, B, H, E // H).permute(1, 2, 0, 3)
        k = k.reshape(S, B, H, E // H).permute(1, 2, 0, 3)
        v = v.reshape(S, B, H, E // H).permute(1, 2, 0, 3)
        out = F.scaled_dot_product_attention(q, k, v, dropout_p=p_drop)
        return tall_linear(out.permute(2, 0, 1, 3).reshape(L, B, E), wo, bo)

    def tail_dropout_p(self):
        """p of ``dropout_layer`` when the tail can be fused (no extra ``proj_drop``), else None."""
        if self.proj_drop.p != 0 and self.training:
            return None
        if isinstance(self.dropout_layer, nn.Identity):
            return 0.0
        return self.dropout_layer.p if self.training else 0.0

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, fuse_tail=False, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        shared_qk = key is query and key_pos is query_pos       # self-attention: one tensor for q and k
        if query_pos is not None:
            query = query + query_pos
        if shared_qk:
            key = query
        elif key_pos is not None:
            key = key + key_pos
        tbf = kwargs.get("tokens_batch_first")
        bf = self.batch_first if tbf is None else bool(tbf)     # set by the decoder's batch-first GPU path: [B, L, E] tokens
        if attn_mask is None and key_padding_mask is None and query.is_cuda and self.attn.in_proj_weight is not None \
                and self.attn.in_proj_bias is not None and self.attn._qkv_same_embed_dim:
            out = self._attend(query, key, value, batch_first=bf)
        else:
            if bf:
                query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
            out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask,
                            need_weights=False)[0]
            if bf:
                out = out.transpose(0, 1)
        if fuse_tail:
            return out
        return identity + self.dropout_layer(self.proj_drop(out))


@FEEDFORWARD_NETWORK.register_module()
class FFN(nn.Module):
    """ext-mmcv FFN: Sequential(Linear, act, Dropout) x (num_fcs-1), Linear, Dropout + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        act = {"ReLU": lambda: nn.ReLU(inplace=True), "GELU": nn.GELU}[act_cfg.get("type", "ReLU")]
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(Linear(cin, feedforward_channels), act(), nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers += [Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Dropout(dropout_layer["drop_prob"]) if dropout_layer else nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.core(x, final_dropout=True)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)

    def tail_dropout_p(self):
        """p of the trailing ``Dropout(ffn_drop)`` when the tail can be fused, else None."""
        if not self.add_identity or not isinstance(self.dropout_layer, nn.Identity) or \
                not isinstance(self.layers[-1], nn.Dropout):
            return None
        return self.layers[-1].p if self.training else 0.0

    def core(self, x, final_dropout):
        """The Linear/act/Dropout stack; ``final_dropout=False`` leaves out the trailing
        ``Dropout(ffn_drop)`` (the fused encoder path applies it inside the LayerNorm launch)."""
        first = self.layers[0]
        rest = list(self.layers)[1:] if final_dropout else list(self.layers)[1:-1]
        if self.num_fcs == 2 and isinstance(first[1], nn.ReLU) and isinstance(self.layers[1], nn.Linear):
            # both Linears + ReLU + Dropout in one MFMA kernel per direction when the shape allows
            lin2 = self.layers[1]
            frozen = not first[0].weight.requires_grad
            lp1 = first[0].frozen_lp(x.device.type) if frozen else first[0].lp()
            lp2 = lin2.frozen_lp(x.device.type) if frozen else lin2.lp()
            if lp1 is not None and lp2 is not None and ffn_fused_ok(x, lp1[0], lp2[0], lp1[1], lp2[1]):
                out = ffn_fused(x, lp1[0], lp1[1], lp2[0], lp2[1], first[2].p if first[2].training else 0.0, owner=self)
                if out is not None:
                    for m in rest[1:]:
                        out = m(out)
                    return out
        if self.num_fcs == 2 and isinstance(first[1], nn.ReLU) and first[0].weight.requires_grad:
            # Linear + ReLU as one GEMM with a fused epilogue, then the rest of the stack
            w1, b1 = first[0].lp()
            out = ffn_inner(x, w1, b1, first[2].p if first[2].training else 0.0)
            for m in rest:
                out = m(out)
        elif self.num_fcs == 2 and isinstance(first[1], nn.ReLU) and x.is_cuda and not torch.is_grad_enabled() \
                and first[0].bias is not None and first[0].frozen_lp(x.device.type) is not None:
            # frozen teacher under autocast: cached low-precision weights + fused bias/ReLU epilogue
            w, b = first[0].frozen_lp(x.device.type)
            x2 = x.reshape(-1, x.shape[-1]).to(w.dtype)
            if x2.is_contiguous() and native.gemm_nt_2d_ok(x2, w, b):
                out = native.gemm_nt_2d(x2, w, b, True).view(*x.shape[:-1], -1)
            else:
                out = torch._addmm_activation(b, x2, w.t()).view(*x.shape[:-1], -1)
            out = first[2](out)
            for m in rest:
                out = m(out)
        else:
            out = first(x)
            for m in rest:
                out = m(out)
        return out


@TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(nn.Module):
    """ext-mmcv ``BaseTransformerLayer``: runs ``operation_order`` over attentions/ffns/norms."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2,
                                                     ffn_drop=0., act_cfg=dict(type="ReLU", inplace=True)),
                 operation_order=None, norm_cfg=dict(type="LN"), init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        ffn_cfgs = copy.deepcopy(dict(ffn_cfgs))
        for old, new in dict(feedforward_channels="feedforward_channels", ffn_dropout="ffn_drop",
                             ffn_num_fcs="num_fcs").items():
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        self.batch_first = batch_first
        assert set(operation_order) <= {"self_attn", "norm", "ffn", "cross_attn"}
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        idx = 0
        for op in operation_order:
            if op in ("self_attn", "cross_attn"):
                cfg = dict(attn_cfgs[idx])
                cfg.setdefault("batch_first", batch_first)
                att = build_attention(cfg)
                att.operation_name = op
                self.attentions.append(att)
                idx += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        num_ffns = operation_order.count("ffn")
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        for i in range(num_ffns):
            c = dict(ffn_cfgs[i])
            c.setdefault("embed_dims", self.embed_dims)
            c.setdefault("type", "FFN")
            self.ffns.append(build_feedforward_network(c))
        self.norms = nn.ModuleList(nn.LayerNorm(self.embed_dims) for _ in range(operation_order.count("norm")))

    def _fused_plan(self):
        """[(op, module, norm)] when the layer is post-norm with every sub-layer followed by its
        LayerNorm and 256 wide, else None.  Cached."""
        plan = self.__dict__.get("_fused_plan_cache", False)
        if plan is False:
            order = tuple(self.operation_order)
            plan = None
            if not self.pre_norm and len(order) % 2 == 0 and self.embed_dims == 256 and \
                    all(o != "norm" for o in order[0::2]) and all(o == "norm" for o in order[1::2]):
                plan, ai, fi = [], 0, 0
                for k, op in enumerate(order[0::2]):
                    if op == "ffn":
                        mod, fi = self.ffns[fi], fi + 1
                    else:
                        mod, ai = self.attentions[ai], ai + 1
                    if not hasattr(mod, "tail_dropout_p") or not self.norms[k].elementwise_affine:
                        plan = None
                        break
                    plan.append((op, mod, self.norms[k]))
            self.__dict__["_fused_plan_cache"] = plan
        return plan

    def _forward_fused(self, plan, query, key, value, query_pos, key_pos, attn_masks, query_key_padding_mask,
                       key_padding_mask, kwargs):
        """MI355X path of a post-norm layer on the GPU: every ``identity + dropout(out)`` -> LayerNorm
        is one HIP launch each way (native.add_layer_norm) and the residual stream stays in the
        compute dtype -- PyTorch's mixed-dtype adds alone cost 60 us each on the 300-query decoder."""
        dev = query.device.type
        dtype = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else query.dtype
        x = query.to(dtype)
        if query_pos is not None and query_pos.dtype != dtype:
            query_pos = query_pos.to(dtype)
        if key_pos is not None and key_pos.dtype != dtype:
            key_pos = key_pos.to(dtype)
        # ``x + query_pos`` for the NEXT attention sub-layer comes out of the LayerNorm launch in front of it (want_q: one more
        # 16-byte store per lane instead of an add launch, and in the backward the two gradients are summed inside
        # add_ln_bwd) -- within the layer, and across layers through a tag on the layer's output (the decoder hands each
        # layer's output straight to the next).  Only when pos has the shape of the tokens (elementwise: any layout).
        can_q = query_pos is not None and query_pos.shape == x.shape and query_pos.dtype == dtype and x.is_cuda
        tag = getattr(query, "_dskd_q", None)
        q_next = tag[0] if (can_q and tag is not None and tag[1] is query_pos and tag[0].shape == x.shape) else None
        ai = 0
        x_ffn = None
        for k, (op, mod, norm) in enumerate(plan):
            p = mod.tail_dropout_p()
            if op == "ffn":
                h = mod.core(x if x_ffn is None else x_ffn, final_dropout=False)
                x_ffn = None
            elif op == "self_attn":
                if q_next is not None:        # q = k = x + pos already formed; v = x
                    h = mod(q_next, q_next, x, None, query_pos=None, key_pos=None, attn_mask=attn_masks[ai],
                            key_padding_mask=query_key_padding_mask, fuse_tail=True, **kwargs)
                else:
                    h = mod(x, x, x, None, query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[ai],
                            key_padding_mask=query_key_padding_mask, fuse_tail=True, **kwargs)
                ai += 1
            else:
                if q_next is not None:
                    h = mod(q_next, key, value, None, query_pos=None, key_pos=key_pos, attn_mask=attn_masks[ai],
                            key_padding_mask=key_padding_mask, fuse_tail=True, **kwargs)
                else:
                    h = mod(x, key, value, None, query_pos=query_pos, key_pos=key_pos, attn_mask=attn_masks[ai],
                            key_padding_mask=key_padding_mask, fuse_tail=True, **kwargs)
                ai += 1
            nxt = plan[(k + 1) % len(plan)][0]          # the sub-layer that reads this LayerNorm's output (next layer: same plan)
            want_q = can_q and nxt != "ffn" and (nxt == "self_attn" or key_pos is None)
            if nxt == "ffn" and k + 1 < len(plan) and x.is_cuda:
                # the FFN and the residual of the LayerNorm behind it both read this output: two autograd outputs, their
                # gradients summed inside add_ln_bwd (see the encoder)
                x_ffn, x, q_next = native.add_layer_norm(h.to(dtype), x, norm, p=p, fork=True)
            else:
                x_ffn = None
                x, q_next = native.add_layer_norm(h.to(dtype), x, norm, p=p, pos=query_pos if want_q else None, want_q=want_q)
        if q_next is not None:
            x._dskd_q = (q_next, query_pos)
        return x

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        dev = query.device.type
        if query.is_cuda and (torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else query.dtype) in (
                torch.float32, torch.bfloat16):
            plan = self._fused_plan()
            if plan is not None and all(m.tail_dropout_p() is not None for _, m, _ in plan):
                return self._forward_fused(plan, query, key, value, query_pos, key_pos, attn_masks,
                                           query_key_padding_mask, key_padding_mask, kwargs)
        norm_index = attn_index = ffn_index = 0
        identity = query
        for layer in self.operation_order:
            if layer == "self_attn":
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_index], key_padding_mask=query_key_padding_mask,
                    **kwargs)
                attn_index += 1
                identity = query
            elif layer == "norm":
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == "cross_attn":
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos, key_pos=key_pos,
                    attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "ffn":
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """transformer.py:407-458."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels, ffn_dropout=ffn_dropout,
                         operation_order=operation_order, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == {"self_attn", "norm", "cross_attn", "ffn"}


class TransformerLayerSequence(nn.Module):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList(build_transformer_layer(dict(c)) for c in transformerlayers)
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos, attn_masks=attn_masks,
                          query_key_padding_mask=query_key_padding_mask, key_padding_mask=key_padding_mask, **kwargs)
        return query


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    """transformer.py:461-497 (post_norm only when pre_norm)."""

    def __init__(self, *args, post_norm_cfg=dict(type="LN"), **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = nn.LayerNorm(self.embed_dims) if (post_norm_cfg is not None and self.pre_norm) else None

    def forward(self, *args, **kwargs):
        x = self._forward_fused(*args, **kwargs)
        if x is not None:
            return x
        x = super().forward(*args, **kwargs)
        if self.post_norm is not None:
            x = self.post_norm(x)
        return x

    def _fusable(self):
        ok = getattr(self, "_fusable_cache", None)
        if ok is None:
            ok = self.post_norm is None and self.embed_dims == 256
            for layer in self.layers:
                ok = ok and tuple(layer.operation_order) == ("self_attn", "norm", "ffn", "norm")
                if not ok:
                    break
                att, ffn = layer.attentions[0], layer.ffns[0]
                ok = ok and isinstance(att, MultiScaleDeformableAttention) and isinstance(ffn, FFN) \
                    and ffn.add_identity and isinstance(ffn.dropout_layer, nn.Identity) \
                    and isinstance(ffn.layers[-1], nn.Dropout) and all(n.elementwise_affine for n in layer.norms)
            self._fusable_cache = ok
        return ok

    def _forward_fused(self, query, key=None, value=None, query_pos=None, query_key_padding_mask=None,
                       reference_points=None, spatial_shapes=None, tokens_batch_first=None, **kwargs):
        """MI355X path of the deformable encoder (post-norm layers, batch-first tokens on the
        GPU): the residual stream stays in the compute dtype and each sub-layer's
        ``identity + dropout(out)`` -> LayerNorm -> (next layer's) ``+ query_pos`` is ONE HIP
        launch each way (native.add_layer_norm) instead of dropout / mixed-dtype add / fp32
        LayerNorm / cast / positional add.  Returns None when the configuration does not
        match, and the generic operation_order interpreter runs instead."""
        if not (tokens_batch_first and query.is_cuda and query_pos is not None and reference_points is not None
                and reference_points.shape[-1] == 2 and self._fusable()):
            return None
        dev = query.device.type
        dtype = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else query.dtype
        if dtype not in (torch.float32, torch.bfloat16):
            return None
        x = query.to(dtype)
        pos = query_pos.float()
        if dtype != torch.float32 and query_pos.requires_grad:
            # The table enters autograd in the compute dtype (its gradient -- one [tokens, 256] tensor per layer, summed
            # over the layers for the level embeddings -- is then handed back and accumulated in bf16: no f32 cast /
            # summing pass per layer); the kernels read the f32 values attached to it.
            pos_in = query_pos.to(dtype)
            pos_in._dskd_f32 = pos.detach()
            pos = pos_in
        q = native.add_pos(x, pos)                  # one pass; ATen: generic mixed-dtype add (120 us) + cast
        last = len(self.layers) - 1
        # Every LayerNorm output feeds TWO consumers (the next sub-layer and the next residual add).  The LayerNorm hands it
        # out as two autograd outputs (fork) and sums their gradients inside add_ln_bwd: autograd's own sum was an add
        # launch over [B, 22 223, 256] per LayerNorm and step (12 of the 21 such adds, profiles/r04_aten_tail.txt).
        xv = xr = x                                  # value_proj input / residual of the attention sub-layer
        for i, layer in enumerate(self.layers):
            att, ffn = layer.attentions[0], layer.ffns[0]
            h = att.core(q, xv, reference_points, spatial_shapes, query_key_padding_mask)
            x1f, x1r, _ = native.add_layer_norm(h, xr, layer.norms[0], p=att.dropout.p if att.training else 0.0, fork=True)
            p_tail = ffn.layers[-1].p if ffn.training else 0.0
            f = ffn.core(x1f, final_dropout=False)
            if i == last:
                x, _ = native.add_layer_norm(f, x1r, layer.norms[1], p=p_tail)
            else:
                xv, xr, q = native.add_layer_norm(f, x1r, layer.norms[1], p=p_tail, pos=pos, want_q=True, fork=True)
        return x


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DeformableDetrTransformerDecoder(TransformerLayerSequence):
    """transformer.py:624-709."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate

    def batch_first_ok(self):
        """Can the layers run on [B, L, E] tokens: every attention module is one of ours that takes ``tokens_batch_first``
        (MultiheadAttention for the self-attention, MultiScaleDeformableAttention for the cross-attention)?"""
        ok = self.__dict__.get("_bf_ok")
        if ok is None:
            ok = all(isinstance(a, (MultiheadAttention, MultiScaleDeformableAttention))
                     for layer in self.layers for a in layer.attentions)
            self.__dict__["_bf_ok"] = ok
        return ok

    def forward(self, query, *args, reference_points=None, valid_ratios=None, reg_branches=None, **kwargs):
        output = query
        # ``tokens_batch_first`` (set by DeformableDetrTransformer on the GPU): query / query_pos come as [B, L, E] and every
        # layer keeps that layout -- the deformable cross-attention and the regression branches are batch-first anyway, and
        # the reference's [L, B, E] costs a permute + copy of the query tensor around each of them (transformer.py:983-995)
        bf = bool(kwargs.get("tokens_batch_first"))
        qp = kwargs.get("query_pos")
        if qp is not None and qp.is_cuda and torch.is_autocast_enabled(qp.device.type):
            # every layer's fused path wants the positional queries in the compute dtype: cast once, not six times
            dt = torch.get_autocast_dtype(qp.device.type)
            if qp.dtype != dt and dt in (torch.bfloat16, torch.float16):
                kwargs["query_pos"] = qp.to(dt)
                if bf:      # the LayerNorm launches that also form x + query_pos read the f32 values (native._pos_f32)
                    kwargs["query_pos"]._dskd_f32 = qp.detach().float().contiguous()
        intermediate, intermediate_reference_points = [], []
        reference_points_input = None
        for lid, layer in enumerate(self.layers):
            # without box refinement (every DSKD config) the reference points never change: one product for all layers
            if reference_points_input is None or reg_branches is not None:
                if reference_points.shape[-1] == 4:
                    reference_points_input = reference_points[:, :, None] * \
                        torch.cat([valid_ratios, valid_ratios], -1)[:, None]
                else:
                    assert reference_points.shape[-1] == 2
                    reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            output = layer(output, *args, reference_points=reference_points_input, **kwargs)
            if not bf:
                output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branches[lid](output)
                if reference_points.shape[-1] == 4:
                    new_reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_reference_points = tmp
                    new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points)
                    new_reference_points = new_reference_points.sigmoid()
                reference_points = new_reference_points.detach()
            if not bf:
                output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            states = torch.stack(intermediate)
            # batch-first run: [layers, B, L, E] in memory, handed out in the reference layout [layers, L, B, E] as a view
            return (states.permute(0, 2, 1, 3) if bf else states), torch.stack(intermediate_reference_points)
        return (output.permute(1, 0, 2) if bf else output), reference_points


@TRANSFORMER.register_module()
class DeformableDetrTransformer(nn.Module):
    """transformer.py:712-1055 (single-stage path; ``as_two_stage`` is off in every DSKD
    config and is not built).  Returns the fork's 6-tuple (:1053-1055)."""

    def __init__(self, encoder=None, decoder=None, as_two_stage=False, num_feature_levels=4,
                 two_stage_num_proposals=300, init_cfg=None, **kwargs):
        super().__init__()
        assert not as_two_stage, "as_two_stage is not used by the DSKD configs and is not implemented"
        self.encoder = build_transformer_layer_sequence(dict(encoder))
        self.decoder = build_transformer_layer_sequence(dict(decoder))
        self.embed_dims = self.encoder.embed_dims
        self.as_two_stage = as_two_stage
        self.num_feature_levels = num_feature_levels
        self.two_stage_num_proposals = two_stage_num_proposals
        self.level_embeds = nn.Parameter(torch.zeros(self.num_feature_levels, self.embed_dims))
        self.reference_points = Linear(self.embed_dims, 2)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()
        nn.init.xavier_uniform_(self.reference_points.weight)
        nn.init.constant_(self.reference_points.bias, 0.)
        nn.init.normal_(self.level_embeds)

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """:830-863."""
        pts = []
        for lvl, (H, W) in enumerate(spatial_shapes):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device),
                                          indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            pts.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(pts, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def get_valid_ratio(self, mask):
        """:865-873."""
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def forward(self, mlvl_feats, mlvl_masks, query_embed, mlvl_pos_embeds, reg_branches=None, cls_branches=None,
                all_valid=False, **kwargs):
        """``all_valid=True`` (no image in the batch is padded, known on the host from
        img_metas): the padding masks are all False, so they are not materialised, the
        valid ratios are exactly 1 and the encoder reference points are a cached constant --
        same numbers, fewer passes over the 22k-token tensors."""
        assert query_embed is not None
        feat_flatten, mask_flatten, lvl_pos_embed_flatten, spatial_shapes = [], [], [], []
        for lvl, (feat, mask, pos_embed) in enumerate(zip(mlvl_feats, mlvl_masks, mlvl_pos_embeds)):
            bs, c, h, w = feat.shape
            spatial_shapes.append((h, w))
            feat_flatten.append(feat.flatten(2).transpose(1, 2))
            if not all_valid:
                mask_flatten.append(mask.flatten(1))
            # (a positional encoding with batch size 1 -- the head's, for an un-padded batch -- makes a [1, sum HW, C] table:
            # the kernels that add it repeat its rows over the batch)
            lvl_pos_embed_flatten.append(_AddLevelEmbed.apply(pos_embed.flatten(2).transpose(1, 2), self.level_embeds[lvl]))
        feat_flatten = torch.cat(feat_flatten, 1)
        mask_flatten = torch.cat(mask_flatten, 1) if not all_valid else None
        lvl_pos_embed_flatten = torch.cat(lvl_pos_embed_flatten, 1)
        device = feat_flatten.device
        level_start_index = [0]
        for h, w in spatial_shapes[:-1]:
            level_start_index.append(level_start_index[-1] + h * w)
        if all_valid:
            bs0 = feat_flatten.shape[0]
            key = (tuple(spatial_shapes), bs0, device)
            ref_cache = self.__dict__.setdefault("_ref_cache", {})      # per shape: captured head graphs read these tensors
            if key not in ref_cache:
                vr = torch.ones((bs0, len(spatial_shapes), 2), dtype=torch.float32, device=device)
                if len(ref_cache) >= 16:              # bounded; the graph captured on an entry pins it (head._forward_graphed)
                    ref_cache.pop(next(iter(ref_cache)))
                ref_cache[key] = (vr, self.get_reference_points(spatial_shapes, vr, device=device))
            valid_ratios, reference_points = ref_cache[key]
            mask_flatten = None
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
            reference_points = self.get_reference_points(spatial_shapes, valid_ratios, device=device)

        # The encoder runs batch-first [bs, sum HW, C] (the reference feeds (sum HW, bs, C) and
        # MSDA permutes inside, transformer.py:983-995): LayerNorm / FFN / MSDA are per-token,
        # so the numbers are identical and no 22k-token tensor is permuted + copied.
        memory = self.encoder(query=feat_flatten, key=None, value=None, query_pos=lvl_pos_embed_flatten,
                              query_key_padding_mask=mask_flatten, spatial_shapes=spatial_shapes,
                              reference_points=reference_points, level_start_index=level_start_index,
                              valid_ratios=valid_ratios, tokens_batch_first=True, **kwargs)
        bs, _, c = memory.shape
        query_pos, query = torch.split(query_embed, c, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        init_reference_out = reference_points

        dec_bf = memory.is_cuda and self.decoder.batch_first_ok()
        if not dec_bf:
            query = query.permute(1, 0, 2)
            query_pos = query_pos.permute(1, 0, 2)
        inter_states, inter_references = self.decoder(
            query=query, key=None, value=memory, query_pos=query_pos, key_padding_mask=mask_flatten,
            reference_points=reference_points, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
            valid_ratios=valid_ratios, reg_branches=reg_branches, value_batch_first=True,
            **(dict(kwargs, tokens_batch_first=True) if dec_bf else kwargs))
        spatial_shapes_t = device_const(spatial_shapes, torch.long, device)
        info_all = (memory.permute(1, 0, 2), spatial_shapes_t)      # reference layout (sum HW, bs, C), a view
        return inter_states, init_reference_out, inter_references, info_all, None, None
