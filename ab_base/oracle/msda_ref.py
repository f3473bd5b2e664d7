"""CPU restatement of multi-scale deformable attention (TEST INFRASTRUCTURE).

The op lives in ext-mmcv (``mmcv-full>=1.3.17,<=1.6.2``, mmcv/ops/multi_scale_deform_attn.py:
``multi_scale_deformable_attn_pytorch``), which is NOT in /root/reference; the reference only
imports it (mmdet/models/utils/transformer.py:22-29) and calls it from the encoder/decoder
(transformer.py:985-995, :1032-1043).  Published formulation restated here:

    for each level l: value_l -> [B*heads, ch, H_l, W_l];
                      grid = 2 * loc[:, :, :, l] - 1  -> [B*heads, Nq, P, 2]
                      sampled_l = grid_sample(value_l, grid, bilinear, zeros, align_corners=False)
    out = (stack(sampled) * attn).sum(-1)  -> [B, Nq, heads*ch]

``msda_scalar`` is the same thing written as explicit loops from the formula in SURVEY.md
appendix A; ``tests/test_oracle.py`` checks the two against each other and against the
independent implementation shipped in the installed ``transformers`` package.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B,Nv,heads,ch]; sampling_locations [B,Nq,heads,L,P,2]; attention_weights
    [B,Nq,heads,L,P]  ->  [B,Nq,heads*ch].  Differentiable (autograd = backward oracle)."""
    B, _, heads, ch = value.shape
    _, Nq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    value_list = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (H, W) in enumerate(spatial_shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(B * heads, ch, int(H), int(W))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    attn = attention_weights.transpose(1, 2).reshape(B * heads, 1, Nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * attn).sum(-1).view(B, heads * ch, Nq)
    return out.transpose(1, 2).contiguous()


def msda_scalar(value, spatial_shapes, loc, attn):
    """Pure-loop float64 restatement (small cases only)."""
    value = np.asarray(value, dtype=np.float64)
    loc = np.asarray(loc, dtype=np.float64)
    attn = np.asarray(attn, dtype=np.float64)
    B, Nv, heads, ch = value.shape
    _, Nq, _, L, P, _ = loc.shape
    starts = np.cumsum([0] + [int(h) * int(w) for h, w in spatial_shapes])
    out = np.zeros((B, Nq, heads, ch))
    for b in range(B):
        for q in range(Nq):
            for h in range(heads):
                for l, (H, W) in enumerate(spatial_shapes):
                    H, W = int(H), int(W)
                    for p in range(P):
                        x = loc[b, q, h, l, p, 0] * W - 0.5
                        y = loc[b, q, h, l, p, 1] * H - 0.5
                        if not (x > -1 and y > -1 and x < W and y < H):
                            continue
                        x0, y0 = math.floor(x), math.floor(y)
                        for dy in (0, 1):
                            for dx in (0, 1):
                                xx, yy = x0 + dx, y0 + dy
                                if 0 <= xx < W and 0 <= yy < H:
                                    wgt = (1 - abs(x - xx)) * (1 - abs(y - yy))
                                    out[b, q, h] += attn[b, q, h, l, p] * wgt * value[b, starts[l] + yy * W + xx, h]
    return out.reshape(B, Nq, heads * ch)
