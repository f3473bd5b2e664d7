"""Swin Transformer backbone with the reference's constructor surface and state-dict names
(/root/reference/mmdet/models/backbones/swin.py: ``SwinTransformer`` :467-763, ``SwinBlockSequence``
:381-464, ``SwinBlock`` :288-378, ``ShiftWindowMSA`` :128-285, ``WindowMSA`` :22-125; patch
embedding / merging from /root/reference/mmdet/models/utils/transformer.py: ``AdaptivePadding``
:62-131, ``PatchEmbed`` :134-257, ``PatchMerging`` :260-385).  SURVEY.md section 8f row 2: serves
BASELINE config #4 (Swin-T + ChannelMapper(in_channels=[192, 384, 768]) + the DSKD head).

Dense work runs on hipBLASLt / the fused attention kernels of PyTorch-ROCm (MFMA).  What is done
differently from the reference, with identical numbers:
  * window attention is ONE ``scaled_dot_product_attention`` call per block with the relative
    position bias and the shift mask folded into a single additive mask (the reference
    materialises q@k^T, adds bias and mask, softmaxes and multiplies by v as separate ops);
  * the shift masks depend only on the padded resolution: built once per geometry and cached;
  * the pad -> roll -> window-partition chain is one gather of cached indices per direction.
Parity is pinned against the independent implementation in ``transformers`` (tests/test_swin.py)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .builder import BACKBONES


def convert_official_swin(state_dict):
    """Checkpoint of the original Swin release (``layers.i.blocks.j.attn.qkv``, ``mlp.fc1``,
    ``patch_embed.proj`` ...; position-major patch-merging channels) -> the names and the
    ``nn.Unfold`` channel order of this backbone (what ``convert_weights=True`` asks for; the
    reference does it in mmdet/models/utils/ckpt_convert.py:85-137).  Classification-head
    entries are dropped."""
    def to_unfold_order(t):            # last dim: [(0,0), (1,0), (0,1), (1,1)] x C  ->  c*4 + kh*2 + kw
        c = t.shape[-1] // 4
        t = t.reshape(*t.shape[:-1], 4, c)[..., [0, 2, 1, 3], :]
        return t.transpose(-1, -2).reshape(*t.shape[:-2], 4 * c)

    out = {}
    for k, v in state_dict.items():
        if k.startswith("head"):
            continue
        if k.startswith("layers"):
            if "attn." in k:
                k = k.replace("attn.", "attn.w_msa.")
            elif "mlp.fc1." in k:
                k = k.replace("mlp.fc1.", "ffn.layers.0.0.")
            elif "mlp.fc2." in k:
                k = k.replace("mlp.fc2.", "ffn.layers.1.")
            elif "downsample" in k and ("reduction." in k or "norm." in k):
                v = to_unfold_order(v)
            k = k.replace("layers", "stages", 1)
        elif k.startswith("patch_embed"):
            k = k.replace("proj", "projection") if "projection" not in k else k
        out[k] = v
    return out


class DropPath(nn.Module):
    """Stochastic depth per sample (ext-mmcv ``DropPath``)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * (mask / keep)


def _corner_pad(size, kernel, stride):
    """AdaptivePadding('corner'): zeros appended at the bottom / right (transformer.py:109-131)."""
    out = math.ceil(size / stride)
    return max((out - 1) * stride + kernel - size, 0)


class PatchEmbed(nn.Module):
    """Non-overlapping patch projection + LayerNorm; returns ([B, L, C], (H, W))."""

    def __init__(self, in_channels, embed_dims, patch_size, norm):
        super().__init__()
        self.patch_size = patch_size
        self.projection = nn.Conv2d(in_channels, embed_dims, patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dims) if norm else None

    def forward(self, x):
        ph = _corner_pad(x.shape[-2], self.patch_size, self.patch_size)
        pw = _corner_pad(x.shape[-1], self.patch_size, self.patch_size)
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        x = self.projection(x)
        hw = (x.shape[2], x.shape[3])
        x = x.flatten(2).transpose(1, 2)
        return (self.norm(x) if self.norm is not None else x), hw


class PatchMerging(nn.Module):
    """2x2 patch merging in the reference's ``nn.Unfold`` channel order (channel-major: index
    c*4 + kh*2 + kw), LayerNorm(4C), Linear(4C -> out, no bias)."""

    def __init__(self, in_channels, out_channels, stride=2, norm=True):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.norm = nn.LayerNorm(4 * in_channels) if norm else None
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=False)

    def forward(self, x, hw):
        B, L, C = x.shape
        H, W = hw
        assert L == H * W, "input feature has wrong size"
        x = x.view(B, H, W, C)
        ph, pw = _corner_pad(H, 2, self.stride), _corner_pad(W, 2, self.stride)
        if ph or pw:
            x = F.pad(x, (0, 0, 0, pw, 0, ph))
            H, W = H + ph, W + pw
        assert self.stride == 2, "only the 2x2 / stride-2 merging of the Swin configs is implemented"
        # [B, H/2, 2, W/2, 2, C] -> [B, H/2, W/2, C, kh, kw]  (== Unfold's c*4 + kh*2 + kw)
        x = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, (H // 2) * (W // 2), 4 * C)
        if self.norm is not None:
            x = self.norm(x)
        return self.reduction(x), (H // 2, W // 2)


class WindowMSA(nn.Module):
    """Window attention with relative position bias (:22-125)."""

    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None, attn_drop_rate=0.0,
                 proj_drop_rate=0.0):
        super().__init__()
        self.embed_dims, self.num_heads, self.window_size = embed_dims, num_heads, window_size
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        # (:62-69) index of the bias of token pair (i, j) of a window
        seq1 = torch.arange(0, (2 * Ww - 1) * Wh, 2 * Ww - 1)
        seq2 = torch.arange(0, Ww, 1)
        rel = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
        index = (rel + rel.T).flip(1).contiguous()
        self.register_buffer("relative_position_index", index)
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop_rate)
        self.proj = nn.Linear(embed_dims, embed_dims)
        self.proj_drop = nn.Dropout(proj_drop_rate)

    def init_weights(self):
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias(self):
        N = self.window_size[0] * self.window_size[1]
        return self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)

    def _mask_types(self, mask):
        """The distinct masks of a shifted layer (at most four: interior, last row, last column, corner of the window
        grid) and the type of each window of an image; cached per mask tensor."""
        key = (mask.data_ptr(), tuple(mask.shape))
        if self.__dict__.get("_mt_key") != key:
            nW = mask.shape[0]
            types, inverse = torch.unique(mask.reshape(nW, -1), dim=0, return_inverse=True)
            self.__dict__["_mt_key"] = key
            self.__dict__["_mt"] = (types.view(-1, mask.shape[1], mask.shape[2]).contiguous(), inverse.to(torch.int32).contiguous(), mask)
        return self.__dict__["_mt"][:2]

    def forward(self, x, mask=None):
        """x [nW*B, N, C]; mask [nW, N, N] additive (0 / -100) or None."""
        Bw, N, C = x.shape
        nH, d = self.num_heads, C // self.num_heads
        qkv_flat = self.qkv(x)
        if d == native.WINATTN_HEAD_DIM and native.window_attention_ok(qkv_flat, nH, N, self.attn_drop.p if self.training else 0.0):
            # hand-written MFMA window attention (csrc/winattn.hip): q / k / v read in place, no [Bw, nH, N, N] bias tensor
            types, wtype = (None, None) if mask is None else self._mask_types(mask)
            x = native.window_attention(qkv_flat, self.bias(), types, wtype, nH, self.scale)
            return self.proj_drop(self.proj(x))
        qkv = qkv_flat.view(Bw, N, 3, nH, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        bias = self.bias()                                                  # [nH, N, N]
        if mask is not None:
            # per-window mask, repeated over the images.  (A broadcast 5-D formulation keeps the mask
            # small but drops PyTorch-ROCm to the unfused bmm/softmax path: measured slower.)
            nW = mask.shape[0]
            bias = (bias[None] + mask[:, None]).to(q.dtype)
            bias = bias.unsqueeze(0).expand(Bw // nW, -1, -1, -1, -1).reshape(Bw, nH, N, N)
        else:
            bias = bias.to(q.dtype).unsqueeze(0).expand(Bw, -1, -1, -1)
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=self.attn_drop.p if self.training else 0.0,
                                           scale=self.scale)
        x = x.transpose(1, 2)                                               # [Bw, N, nH, d]
        return self.proj_drop(self.proj(x.reshape(Bw, N, C)))


class _TokenGather(torch.autograd.Function):
    """``x.index_select(1, fwd)`` whose backward is ALSO an index_select (``bwd`` = the inverse map;
    a trailing all-zero row absorbs positions without a source).  The window partition and its
    reverse are permutations of the token axis (plus padding): autograd's generic index backward
    is a sort + scatter that cost 25 % of the Swin-T step on MI355X."""

    @staticmethod
    def forward(ctx, x, fwd, bwd, n_in):
        ctx.save_for_backward(bwd)
        ctx.n_in = n_in
        return x.index_select(1, fwd)

    @staticmethod
    def backward(ctx, g):
        (bwd,) = ctx.saved_tensors
        pad = g.new_zeros(g.shape[0], 1, g.shape[2])
        return torch.cat([g, pad], 1).index_select(1, bwd)[:, :ctx.n_in], None, None, None


class ShiftWindowMSA(nn.Module):
    """(Shifted-)window attention on a [B, H*W, C] token map (:128-285)."""

    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None,
                 attn_drop_rate=0.0, proj_drop_rate=0.0, drop_path=0.0):
        super().__init__()
        assert 0 <= shift_size < window_size
        self.window_size, self.shift_size = window_size, shift_size
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale, attn_drop_rate,
                               proj_drop_rate)
        self.drop = DropPath(drop_path)
        self._geo = {}

    def _geometry(self, H, W, device):
        """Cached per resolution: gather indices window-token -> source token (pad / roll /
        partition in one step; padded positions point at an appended zero row), and the shift mask."""
        key = (H, W, str(device))
        if key not in self._geo:
            ws, ss = self.window_size, self.shift_size
            Hp, Wp = H + (ws - H % ws) % ws, W + (ws - W % ws) % ws
            ys, xs = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing="ij")
            src_y, src_x = (ys + ss) % Hp, (xs + ss) % Wp              # roll by -shift
            src = torch.where((src_y < H) & (src_x < W), src_y * W + src_x, torch.full_like(ys, H * W))
            part = src.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1)   # window-major order
            # inverse: position of every real token inside the window-major list
            inv = torch.empty(H * W + 1, dtype=torch.long)
            inv[part] = torch.arange(part.numel())
            mask = None
            if ss > 0:
                img = torch.zeros(Hp, Wp)
                cnt = 0
                for hs in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    for wsl in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                        img[hs, wsl] = cnt
                        cnt += 1
                mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
                diff = mw[:, None, :] - mw[:, :, None]
                mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff)).to(device)
            self._geo[key] = (part.to(device), inv[:H * W].to(device), mask, Hp * Wp)
        return self._geo[key]

    def forward(self, query, hw_shape):
        B, L, C = query.shape
        H, W = hw_shape
        assert L == H * W, "input feature has wrong size"
        part, inv, mask, Lp = self._geometry(H, W, query.device)
        ws2 = self.window_size ** 2
        padded = torch.cat([query, query.new_zeros(B, 1, C)], 1)           # row L = the zero padding token
        # partition: windows[pos] = padded[part[pos]]; gradient: d(padded)[t] = d(windows)[inv[t]] (row L: dropped)
        windows = _TokenGather.apply(padded, part, torch.cat([inv, inv.new_full((1,), Lp)]), L + 1)
        out = self.w_msa(windows.view(B * (Lp // ws2), ws2, C), mask=mask).view(B, Lp, C)
        # reverse: tokens[t] = out[inv[t]]; gradient: d(out)[pos] = d(tokens)[part[pos]], pad slots (part == L) get 0
        return self.drop(_TokenGather.apply(out, inv, part, Lp))


class _SwinFFN(nn.Module):
    """ext-mmcv FFN with the parameter names of the reference (``layers.0.0`` / ``layers.1``)."""

    def __init__(self, embed_dims, feedforward_channels, ffn_drop, drop_path):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Dropout(ffn_drop)),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))
        self.dropout_layer = DropPath(drop_path)

    def forward(self, x, identity):
        return identity + self.dropout_layer(self.layers(x))


class SwinBlock(nn.Module):
    """Pre-norm block: x + attn(norm1(x)); then + ffn(norm2(.)) (:288-378)."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias, qk_scale,
                                   attn_drop_rate, drop_rate, drop_path_rate)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = _SwinFFN(embed_dims, feedforward_channels, drop_rate, drop_path_rate)

    def forward(self, x, hw_shape):
        x = x + self.attn(self.norm1(x), hw_shape)
        return self.ffn(self.norm2(x), identity=x)


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size=7, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, downsample=None):
        super().__init__()
        rates = list(drop_path_rate) if isinstance(drop_path_rate, (list, tuple)) else [drop_path_rate] * depth
        assert len(rates) == depth
        self.blocks = nn.ModuleList(
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, shift=i % 2 == 1, qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=rates[i])
            for i in range(depth))
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for blk in self.blocks:
            x = blk(x, hw_shape)
        if self.downsample is not None:
            down, down_hw = self.downsample(x, hw_shape)
            return down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


@BACKBONES.register_module()
class SwinTransformer(nn.Module):
    """:467-763.  Outputs one NCHW map per ``out_indices`` stage, each behind its ``norm{i}``."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                 use_abs_pos_embed=False, act_cfg=dict(type="GELU"), norm_cfg=dict(type="LN"), with_cp=False,
                 pretrained=None, convert_weights=False, frozen_stages=-1, init_cfg=None):
        super().__init__()
        assert act_cfg.get("type", "GELU") == "GELU" and norm_cfg.get("type", "LN") == "LN"
        assert not with_cp, "activation checkpointing is not implemented"
        assert strides[0] == patch_size, "Use non-overlapping patch embed."
        if isinstance(pretrain_img_size, int):
            pretrain_img_size = (pretrain_img_size, pretrain_img_size)
        self.frozen_stages, self.out_indices, self.use_abs_pos_embed = frozen_stages, tuple(out_indices), use_abs_pos_embed
        self.convert_weights, self.pretrained = convert_weights, pretrained
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size, patch_norm)
        if use_abs_pos_embed:
            n = (pretrain_img_size[0] // patch_size) * (pretrain_img_size[1] // patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, n, embed_dims))
        self.drop_after_pos = nn.Dropout(drop_rate)
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()          # stochastic depth decay rule
        self.stages = nn.ModuleList()
        ch = embed_dims
        for i, depth in enumerate(depths):
            down = PatchMerging(ch, 2 * ch, strides[i + 1], patch_norm) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(ch, num_heads[i], mlp_ratio * ch, depth, window_size, qkv_bias, qk_scale,
                                                 drop_rate, attn_drop_rate, dpr[sum(depths[:i]):sum(depths[:i + 1])], down))
            if down is not None:
                ch = down.out_channels
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    def init_weights(self):
        """:670-743: a ``pretrained`` checkpoint (converted from the official layout when
        ``convert_weights``), else trunc-normal Linear weights and unit LayerNorm."""
        if self.pretrained:
            sd = torch.load(self.pretrained, map_location="cpu")
            sd = sd.get("state_dict", sd.get("model", sd))
            if self.convert_weights:
                sd = convert_official_swin(sd)
            sd = {(k[9:] if k.startswith("backbone.") else k): v for k, v in sd.items()}
            self.load_state_dict(sd, strict=False)
            return
        if self.use_abs_pos_embed:
            nn.init.trunc_normal_(self.absolute_pos_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, WindowMSA):
                m.init_weights()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
            if self.use_abs_pos_embed:
                self.absolute_pos_embed.requires_grad = False
            self.drop_after_pos.eval()
        for i in range(1, self.frozen_stages + 1):
            if (i - 1) in self.out_indices:
                norm = getattr(self, f"norm{i - 1}")
                norm.eval()
                for p in norm.parameters():
                    p.requires_grad = False
            stage = self.stages[i - 1]
            stage.eval()
            for p in stage.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    def forward(self, x):
        x, hw = self.patch_embed(x)
        if self.use_abs_pos_embed:
            x = x + self.absolute_pos_embed
        x = self.drop_after_pos(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw, out, out_hw = stage(x, hw)
            if i in self.out_indices:
                out = getattr(self, f"norm{i}")(out)
                outs.append(out.view(-1, *out_hw, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return outs
