"""ResNet backbone with the reference's constructor surface and state-dict names
(/root/reference/mmdet/models/backbones/resnet.py:306-659: ``ResNet``; forward :631-646,
``_freeze_stages`` :613-629, ``train`` / norm_eval :648-659; ``Bottleneck`` :100-303 with
style='pytorch' = stride on the 3x3 conv).  Dense convolutions run on MIOpen/hipBLASLt (MFMA)
through PyTorch-ROCm; nothing here is hand-written."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .dist import grad_slot
from .builder import BACKBONES


class FrozenAffineBN(nn.BatchNorm2d):
    """BatchNorm2d whose forward, in eval mode, is a fused per-channel scale/shift
    (what ``norm_eval=True`` + ``requires_grad=False`` make of every BN of this backbone)."""

    def forward(self, x):
        if self.training:
            return super().forward(x)
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        shift = self.bias - self.running_mean * scale
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + shift.to(x.dtype).view(1, -1, 1, 1)


_CONV1X1_MFMA = not os.environ.get("DSKD_CONV_LIB")      # A/B switch: library convolutions + the bias_act pass


def _conv_epilogue(conv, x, w, b, relu, identity):
    """Folded convolution WITHOUT bias, then ONE in-place pass for bias (+ identity) (+ ReLU)
    (native.bias_act).  PyTorch's MIOpen path would run the bias add, the residual add and the
    ReLU as separate launches that each stream the whole activation."""
    if _CONV1X1_MFMA and native.conv1x1_ok(x, w, conv) and (b is None or b.shape[0] == w.shape[0]) and \
            (identity is None or identity.is_cuda):
        # 1x1 convolution + folded-BN shift + residual + ReLU as ONE hand-written MFMA launch (csrc/gemm_nt.hip)
        return native.conv1x1(x, w, b, identity, relu, conv.stride[0])
    if _CONV1X1_MFMA and native.conv3x3_ok(x, w, conv) and (b is None or b.shape[0] == w.shape[0]) and \
            (identity is None or identity.is_cuda):
        # 3x3 convolution as an implicit GEMM on the same kernel, epilogue fused (dskd_conv3x3)
        return native.conv3x3(x, w, b, identity, relu, conv.stride[0])
    y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if b is None and identity is None:
        return F.relu(y, inplace=True) if relu else y
    if not y.is_cuda:                       # host tensors: the same arithmetic with PyTorch ops
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        if identity is not None:
            y = y + identity
        return F.relu(y, inplace=True) if relu else y
    if b is None:
        b = y.new_zeros(y.shape[1])
    return native.bias_act(y, b, identity, relu)


def conv_bn(conv, bn, x, relu=False, identity=None):
    """``act(bn(conv(x)) (+ identity))`` with an eval-mode (frozen-statistics) BatchNorm folded
    into the convolution: bn(conv(x, w)) = conv(x, w * scale) + shift with scale = gamma /
    sqrt(var + eps).  Saves two full passes over the activation per conv (the largest maps here
    are 137 MB in bf16), and the shift / residual / ReLU that remain are one fused pass.  When
    neither the conv weight nor the BN affine require grad (the frozen stem/stage 1 and the whole
    teacher) the folded weight is cached in the compute dtype."""
    if bn.training:
        y = bn(conv(x))
        if identity is not None:
            y = y + identity
        return F.relu(y, inplace=True) if relu else y
    live = conv.__dict__.get("_folded_live")
    if live is not None:            # produced for all trainable convs at once by ResNet._fold_trainable
        return _conv_epilogue(conv, x, live[0], live[1], relu, identity)
    dtype = torch.get_autocast_dtype(x.device.type) if torch.is_autocast_enabled(x.device.type) else x.dtype
    frozen = not (conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad)
    key = (conv.weight._version, bn.weight._version, bn.running_var._version, bn.running_mean._version, dtype,
           conv.weight.device)
    cache = conv.__dict__.get("_folded")
    if frozen and cache is not None and cache[0] == key:
        w, b = cache[1], cache[2]
    else:
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        w = (conv.weight * scale.view(-1, 1, 1, 1)).to(dtype)
        b = (bn.bias - bn.running_mean * scale).to(dtype)
        if conv.bias is not None:
            b = b + (conv.bias * scale).to(dtype)
        if frozen:
            w, b = w.detach(), b.detach()
            if x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
                w = w.contiguous(memory_format=torch.channels_last)
            conv.__dict__["_folded"] = (key, w, b)
    return _conv_epilogue(conv, x, w, b, relu, identity)


class _FoldTrainable(torch.autograd.Function):
    """w_i * scale_i (per output channel) for every trainable conv of a backbone stage, cast to the compute dtype, and the
    way back of the gradients.  bf16 on the GPU: ONE launch each way (native.MultiCast / dskd_cast_scale_many: the weight is
    read once, the scale is a [Cout] vector).  Otherwise two multi-tensor launches on scales expanded to the weights' shapes
    (foreach mul + foreach cast) instead of ~5 small launches per conv."""

    @staticmethod
    def forward(ctx, scales, dtype, tables, *weights):
        """``scales``: per conv (vector [Cout] f32, the same expanded to the weight's shape and strides | None);
        ``tables``: (forward, backward) MultiCast of this stage | None."""
        ctx.scales, ctx.wdtype, ctx.tables = scales, weights[0].dtype, tables
        ctx.pstrides = [w.stride() for w in weights]
        ctx.pids = [id(w) for w in weights]
        ctx.wshapes = [w.shape for w in weights]
        vecs = [v for v, _ in scales]
        srcs = [w.detach() for w in weights]
        if tables is not None and dtype == torch.bfloat16 and ctx.wdtype == torch.float32:
            outs = [torch.empty_like(w, dtype=dtype) for w in srcs]
            if tables[0].ready(srcs, outs, vecs):
                tables[0].run(srcs, outs, vecs)
                ctx.fast = True
                return tuple(outs)
        ctx.fast = False
        prod = torch._foreach_mul(srcs, [_full_scale(sc, w) for sc, w in zip(scales, weights)])
        if dtype == ctx.wdtype:
            return tuple(prod)
        outs = [torch.empty_like(p, dtype=dtype) for p in prod]
        torch._foreach_copy_(outs, prod)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        gdt = next((g.dtype for g in grads if g is not None), ctx.wdtype)
        gs = [g if g is not None else torch.zeros(sh, dtype=gdt, device=ctx.scales[0][0].device)
              for g, sh in zip(grads, ctx.wshapes)]
        if ctx.fast and gs[0].dtype == torch.bfloat16:
            # bf16 gradients -> f32 * scale in ONE launch, into the parameters' slots of the flat gradient buffer of a
            # data-parallel run (dist.GradSync) when there is one
            gs = [g if g.stride() == st else g.as_strided(g.shape, st) if g.shape[2:] == (1, 1) else g.contiguous(
                memory_format=torch.channels_last if st[1] == 1 else torch.contiguous_format) for g, st in zip(gs, ctx.pstrides)]
            slots = [grad_slot(pid, g.shape, ctx.wdtype) for pid, g in zip(ctx.pids, gs)]
            ups = [s if s is not None else torch.empty_like(g, dtype=ctx.wdtype) for s, g in zip(slots, gs)]
            vecs = [v for v, _ in ctx.scales]
            if ctx.tables[1].ready(gs, ups, vecs):
                ctx.tables[1].run(gs, ups, vecs)
                return (None, None, None) + tuple(ups)
        full = [_full_scale(sc, g) for sc, g in zip(ctx.scales, gs)]
        if gs[0].dtype != ctx.wdtype:
            slots = [grad_slot(pid, g.shape, ctx.wdtype) for pid, g in zip(ctx.pids, gs)]
            if all(s is not None for s in slots):
                torch._foreach_copy_(slots, gs)
                torch._foreach_mul_(slots, full)
                return (None, None, None) + tuple(slots)
            up = [torch.empty_like(g, dtype=ctx.wdtype) for g in gs]
            torch._foreach_copy_(up, gs)
            gs = up
        outs = torch._foreach_mul(gs, full)
        # 1x1 kernels: [Cin, 1, 1, 1] and the parameter's channels_last [Cin, 1, Cin, Cin] describe the
        # same memory; hand DDP the parameter's strides so it does not copy into its bucket view
        outs = [o.as_strided(o.shape, st) if (o.stride() != st and o.shape[2:] == (1, 1)) else o
                for o, st in zip(outs, ctx.pstrides)]
        return (None, None, None) + tuple(outs)


def _full_scale(sc, like):
    """The per-channel scale expanded to ``like``'s shape and strides (built on first use: only the paths without the
    multi-tensor kernel need it -- fp32 runs, host tensors)."""
    vec, full = sc
    if full is None or full.shape != like.shape or full.stride() != like.stride():
        # same strides as the weight (channels_last models): a stride mismatch sends torch._foreach_mul down its
        # per-tensor slow path -- ~80 single launches per step instead of 6
        full = torch.empty_like(like, dtype=vec.dtype).copy_(vec.view(-1, 1, 1, 1).expand(like.shape))
        sc[1] = full
    return full


def _bn(ch, requires_grad):
    bn = FrozenAffineBN(ch)
    for p in bn.parameters():
        p.requires_grad = requires_grad
    return bn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 bn_requires_grad=True):
        super().__init__()
        assert style in ("pytorch", "caffe")
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = _bn(planes, bn_requires_grad)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _bn(planes, bn_requires_grad)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = _bn(planes * self.expansion, bn_requires_grad)
        self.downsample = downsample

    input_is_relu = False       # set by ResNet: the block's input is the (ReLU) output of another Bottleneck

    def _fused(self, x):
        """The trainable block as ONE autograd node with the elementwise steps of its backward folded into the
        input-gradient GEMMs (native.bottleneck); None when the block is not in that shape (frozen, host tensors, ...)."""
        if not (_CONV1X1_MFMA and x.is_cuda and torch.is_grad_enabled()):
            return None
        convs = [self.conv1, self.conv2, self.conv3] + ([self.downsample[0]] if self.downsample is not None else [])
        live = [c.__dict__.get("_folded_live") for c in convs]
        if any(l is None for l in live):
            return None
        (w1, b1), (w2, b2), (w3, b3) = (l[:2] for l in live[:3])
        wd, bd = live[3][:2] if self.downsample is not None else (None, None)
        down = self.downsample[0] if self.downsample is not None else None
        if not native.bottleneck_ok(x, w1, w2, w3, wd, self.conv1, self.conv2, self.conv3, down):
            return None
        # the operands of the input-gradient launches, made for the whole stage in one launch (ResNet._fold_trainable)
        wts = [l[2] if len(l) > 2 else None for l in live[:3]] + [live[3][2] if (self.downsample is not None and len(live[3]) > 2)
                                                                   else None]
        if self.conv2.stride[0] != 1:
            wts[1] = None             # stride-2 conv2: its input gradient comes from the library
        return native.bottleneck(x, w1, b1, w2, b2, w3, b3, wd, bd, self.conv2.stride[0], self.input_is_relu, wts=wts)

    def forward(self, x):
        out = self._fused(x)
        if out is not None:
            return out
        identity = x
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        if self.downsample is not None:
            identity = conv_bn(self.downsample[0], self.downsample[1], x)
        return conv_bn(self.conv3, self.bn3, out, relu=True, identity=identity)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 bn_requires_grad=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = _bn(planes, bn_requires_grad)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = _bn(planes, bn_requires_grad)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = F.relu(conv_bn(self.conv1, self.bn1, x), inplace=True)
        out = conv_bn(self.conv2, self.bn2, out)
        if self.downsample is not None:
            identity = conv_bn(self.downsample[0], self.downsample[1], x)
        return F.relu(out + identity, inplace=True)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {
        18: (BasicBlock, (2, 2, 2, 2)),
        34: (BasicBlock, (3, 4, 6, 3)),
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3)),
    }

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch",
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f"invalid depth {depth} for resnet")
        assert not deep_stem and not avg_down and dcn is None and plugins is None, \
            "only the plain ResNet of the DSKD configs is implemented"
        assert norm_cfg.get("type", "BN") == "BN"
        self.depth = depth
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        self.init_cfg = init_cfg
        self.zero_init_residual = zero_init_residual
        bn_rg = norm_cfg.get("requires_grad", True)
        stem = stem_channels or base_channels
        block, stage_blocks = self.arch_settings[depth]
        self.conv1 = nn.Conv2d(in_channels, stem, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(stem, bn_rg)
        self.res_layers = []
        inplanes = stem
        for i, nblk in enumerate(stage_blocks[:num_stages]):
            planes = base_channels * 2 ** i
            layers = []
            for j in range(nblk):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                         _bn(planes * block.expansion, bn_rg))
                layers.append(block(inplanes, planes, stride, dilations[i], down, style, bn_rg))
                layers[-1].input_is_relu = not (i == 0 and j == 0)     # every block but the first follows a block's ReLU
                inplanes = planes * block.expansion
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._freeze_stages()

    def init_weights(self):
        """kaiming for convs, constant for norms (resnet.py:371-385 default init_cfg); a
        'Pretrained' init_cfg is honoured when the checkpoint file exists."""
        ck = (self.init_cfg or {}).get("checkpoint") if isinstance(self.init_cfg, dict) else None
        import os
        if ck and os.path.isfile(ck):
            sd = torch.load(ck, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def _conv_bn_pairs(self):
        """[(conv, bn, stage)]: stage 0 = stem, 1.. = res layers."""
        pairs = [(self.conv1, self.bn1, 0)]
        for si, name in enumerate(self.res_layers):
            for blk in getattr(self, name):
                pairs.append((blk.conv1, blk.bn1, si + 1))
                pairs.append((blk.conv2, blk.bn2, si + 1))
                if hasattr(blk, "conv3"):
                    pairs.append((blk.conv3, blk.bn3, si + 1))
                if blk.downsample is not None:
                    pairs.append((blk.downsample[0], blk.downsample[1], si + 1))
        return pairs

    def _fold_trainable(self, x):
        """Folded (weight, bias) of every conv whose weight trains while its BN is frozen
        (stages 2-4 of the student), for this forward, in a handful of launches."""
        pairs = [(c, b, st) for c, b, st in self._conv_bn_pairs()
                 if c.weight.requires_grad and not b.training and not b.weight.requires_grad and c.bias is None]
        if not pairs or not torch.is_grad_enabled():
            return []
        dtype = torch.get_autocast_dtype(x.device.type) if torch.is_autocast_enabled(x.device.type) else x.dtype
        key = (tuple(b.running_var._version for _, b, _ in pairs), dtype, x.device, tuple(c.weight.stride() for c, _, _ in pairs))
        if getattr(self, "_fold_key", None) != key:      # BN statistics are frozen: constants
            scales, biases = [], []
            with torch.no_grad():
                for c, b, _ in pairs:
                    sc = (b.weight * torch.rsqrt(b.running_var + b.eps)).contiguous()
                    scales.append([sc, None])          # [vector, the same expanded to the weight (built on demand)]
                    biases.append((b.bias - b.running_mean * sc).to(dtype))
            tables = {st: (native.MultiCast(0), native.MultiCast(1)) for st in {p[2] for p in pairs}} if x.is_cuda else {}
            self._fold_key, self._fold_const = key, (scales, biases, tables)
        scales, biases, tables = self._fold_const
        # One fold per stage: its backward hands the stage's weight gradients over as soon as that
        # stage's backward is done (layer4 first: 2/3 of the backbone's parameters), so DDP can
        # all-reduce them while the earlier stages still run their backward.
        for st in sorted({p[2] for p in pairs}):
            idx = [i for i, p in enumerate(pairs) if p[2] == st]
            ws = _FoldTrainable.apply([scales[i] for i in idx], dtype, tables.get(st), *[pairs[i][0].weight for i in idx])
            wts = [None] * len(ws)
            if x.is_cuda and torch.is_grad_enabled() and _CONV1X1_MFMA and ws[0].dtype == torch.bfloat16:
                tr = self.__dict__.setdefault("_fold_wt", {}).setdefault(st, native.WeightTransposes())
                with torch.no_grad():
                    wts = tr.run([w.detach() for w in ws])
            for i, w, wt in zip(idx, ws, wts):
                pairs[i][0].__dict__["_folded_live"] = (w, biases[i], wt)
        return [c for c, _, _ in pairs]

    def _stem(self, x):
        """conv1 -> norm1 -> ReLU -> MaxPool (resnet.py:633-640).  Without gradients (teacher; student with a frozen stem) and
        a folded weight in the cache: the library's 7x7 convolution, then bias + ReLU + pooling as ONE pass
        (native.bias_relu_maxpool) instead of an in-place bias + ReLU pass and the pooling."""
        conv, bn = self.conv1, self.bn1
        folded = conv.__dict__.get("_folded")
        if folded is not None and not bn.training and x.is_cuda and conv.__dict__.get("_folded_live") is None \
                and not (conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad or x.requires_grad):
            dtype = torch.get_autocast_dtype(x.device.type) if torch.is_autocast_enabled(x.device.type) else x.dtype
            key = (conv.weight._version, bn.weight._version, bn.running_var._version, bn.running_mean._version, dtype,
                   conv.weight.device)
            if folded[0] == key and folded[2] is not None:
                y = F.conv2d(x, folded[1], None, conv.stride, conv.padding, conv.dilation, conv.groups)
                if native.bias_relu_maxpool_ok(y, folded[2]):
                    return native.bias_relu_maxpool(y, folded[2])
                return F.max_pool2d(native.bias_act(y, folded[2], None, True), kernel_size=3, stride=2, padding=1)
        x = conv_bn(conv, bn, x, relu=True)         # (also fills the cache of a frozen stem for the next call)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        live = self._fold_trainable(x)
        try:
            x = self._stem(x)
            outs = []
            for i, name in enumerate(self.res_layers):
                x = getattr(self, name)(x)
                if i in self.out_indices:
                    outs.append(x)
        finally:
            for c in live:
                c.__dict__.pop("_folded_live", None)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self
