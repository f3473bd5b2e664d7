"""Transformer layers around the sampling op against an independent
implementation: ``transformers``' Deformable-DETR layers (installed offline; same architecture --
ext-mmcv ``BaseTransformerLayer`` / ``MultiScaleDeformableAttention`` / ``FFN`` are absent from the reference
tree and cannot run here).  HF parameters are mapped onto our reference-named ones; the sampling op runs
through the CPU oracle (``cpu_ops``), as it does through the HIP kernel on the GPU."""
import pytest
import torch

import dskd_amd  # noqa: F401
from dskd_amd.builder import build_transformer_layer

D, F, HEADS, LEVELS, POINTS = 64, 128, 8, 4, 4
SHAPES = [(9, 14), (5, 7), (3, 4), (2, 2)]


def _hf():
    transformers = pytest.importorskip("transformers")
    from transformers.models.deformable_detr import modeling_deformable_detr as m
    cfg = transformers.DeformableDetrConfig(d_model=D, encoder_attention_heads=HEADS, decoder_attention_heads=HEADS,
                                            encoder_n_points=POINTS, decoder_n_points=POINTS, num_feature_levels=LEVELS,
                                            encoder_ffn_dim=F, decoder_ffn_dim=F, dropout=0.0, activation_dropout=0.0,
                                            attention_dropout=0.0, activation_function="relu")
    return m, cfg


def _randomise(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 1 else 0.15))


def _inputs(B, seed, padded):
    g = torch.Generator().manual_seed(seed)
    N = sum(h * w for h, w in SHAPES)
    x = torch.randn(B, N, D, generator=g)
    pos = torch.randn(B, N, D, generator=g)
    ref = torch.rand(B, N, LEVELS, 2, generator=g)
    mask = torch.zeros(B, N, dtype=torch.bool)                    # True = padded (mmcv key_padding_mask)
    if padded:
        mask[1, torch.randperm(N, generator=g)[: N // 5]] = True
    starts = [0]
    for h, w in SHAPES[:-1]:
        starts.append(starts[-1] + h * w)
    return x, pos, ref, mask, starts


@pytest.mark.parametrize("padded", [False, True])
def test_encoder_layer_matches_transformers(cpu_ops, padded):
    """('self_attn', 'norm', 'ffn', 'norm') with deformable self-attention: query = x + pos, value = x with
    padded rows zeroed, softmax over levels x points, offsets / (W, H), residuals and post-norms."""
    m, cfg = _hf()
    hf = m.DeformableDetrEncoderLayer(cfg).eval()
    _randomise(hf, 3)
    ours = build_transformer_layer(dict(
        type="BaseTransformerLayer", attn_cfgs=dict(type="MultiScaleDeformableAttention", embed_dims=D, num_heads=HEADS,
                                                     num_levels=LEVELS, num_points=POINTS),
        ffn_cfgs=dict(type="FFN", embed_dims=D, feedforward_channels=F, num_fcs=2, ffn_drop=0.0,
                      act_cfg=dict(type="ReLU", inplace=True)),
        operation_order=("self_attn", "norm", "ffn", "norm"))).eval()
    sd = hf.state_dict()
    mapped = {}
    for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"):
        for k in ("weight", "bias"):
            mapped[f"attentions.0.{n}.{k}"] = sd[f"self_attn.{n}.{k}"]
    for k in ("weight", "bias"):
        mapped[f"norms.0.{k}"], mapped[f"norms.1.{k}"] = sd[f"self_attn_layer_norm.{k}"], sd[f"final_layer_norm.{k}"]
        mapped[f"ffns.0.layers.0.0.{k}"], mapped[f"ffns.0.layers.1.{k}"] = sd[f"mlp.fc1.{k}"], sd[f"mlp.fc2.{k}"]
    ours.load_state_dict(mapped, strict=True)
    x, pos, ref, mask, starts = _inputs(2, 5, padded)
    with torch.no_grad():
        want = hf(x, attention_mask=~mask, spatial_position_embeddings=pos, reference_points=ref,
                  spatial_shapes=torch.tensor(SHAPES), spatial_shapes_list=SHAPES, level_start_index=torch.tensor(starts))
        want = want[0] if isinstance(want, tuple) else want
        got = ours(x.permute(1, 0, 2), key=None, value=None, query_pos=pos.permute(1, 0, 2),
                   query_key_padding_mask=mask, spatial_shapes=SHAPES, reference_points=ref,
                   level_start_index=starts).permute(1, 0, 2)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    # and batch-first tokens, the layout our encoder runs in
    with torch.no_grad():
        got_bf = ours(x, key=None, value=None, query_pos=pos, query_key_padding_mask=mask, spatial_shapes=SHAPES,
                      reference_points=ref, level_start_index=starts, tokens_batch_first=True)
    torch.testing.assert_close(got_bf, want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("padded", [False, True])
def test_decoder_layer_matches_transformers(cpu_ops, padded):
    """('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'): multi-head self-attention over the queries
    (q = k = x + query_pos, v = x) and deformable cross-attention into the encoder memory (query = x +
    query_pos, padded memory rows zeroed), residuals and post-norms."""
    m, cfg = _hf()
    hf = m.DeformableDetrDecoderLayer(cfg).eval()
    _randomise(hf, 7)
    ours = build_transformer_layer(dict(
        type="DetrTransformerDecoderLayer",
        attn_cfgs=[dict(type="MultiheadAttention", embed_dims=D, num_heads=HEADS, dropout=0.0),
                   dict(type="MultiScaleDeformableAttention", embed_dims=D, num_heads=HEADS, num_levels=LEVELS,
                        num_points=POINTS)],
        ffn_cfgs=dict(type="FFN", embed_dims=D, feedforward_channels=F, num_fcs=2, ffn_drop=0.0,
                      act_cfg=dict(type="ReLU", inplace=True)),
        feedforward_channels=F, ffn_dropout=0.0,
        operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))).eval()
    sd = hf.state_dict()
    mapped = {"attentions.0.attn.in_proj_weight": torch.cat([sd[f"self_attn.{n}_proj.weight"] for n in "qkv"], 0),
              "attentions.0.attn.in_proj_bias": torch.cat([sd[f"self_attn.{n}_proj.bias"] for n in "qkv"], 0),
              "attentions.0.attn.out_proj.weight": sd["self_attn.o_proj.weight"],
              "attentions.0.attn.out_proj.bias": sd["self_attn.o_proj.bias"]}
    for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"):
        for k in ("weight", "bias"):
            mapped[f"attentions.1.{n}.{k}"] = sd[f"encoder_attn.{n}.{k}"]
    for k in ("weight", "bias"):
        mapped[f"norms.0.{k}"], mapped[f"norms.1.{k}"] = sd[f"self_attn_layer_norm.{k}"], sd[f"encoder_attn_layer_norm.{k}"]
        mapped[f"norms.2.{k}"] = sd[f"final_layer_norm.{k}"]
        mapped[f"ffns.0.layers.0.0.{k}"], mapped[f"ffns.0.layers.1.{k}"] = sd[f"mlp.fc1.{k}"], sd[f"mlp.fc2.{k}"]
    ours.load_state_dict(mapped, strict=True)
    memory, _, _, mask, starts = _inputs(2, 9, padded)
    g = torch.Generator().manual_seed(10)
    Q = 23
    x, qpos = torch.randn(2, Q, D, generator=g), torch.randn(2, Q, D, generator=g)
    ref = torch.rand(2, Q, LEVELS, 2, generator=g)                 # reference points already scaled by the valid ratios
    with torch.no_grad():
        want = hf(x, object_queries_position_embeddings=qpos, reference_points=ref, spatial_shapes=torch.tensor(SHAPES),
                  spatial_shapes_list=SHAPES, level_start_index=torch.tensor(starts), encoder_hidden_states=memory,
                  encoder_attention_mask=~mask)
        want = want[0] if isinstance(want, tuple) else want
        got = ours(x.permute(1, 0, 2), key=None, value=memory.permute(1, 0, 2), query_pos=qpos.permute(1, 0, 2),
                   key_padding_mask=mask, reference_points=ref, spatial_shapes=SHAPES,
                   level_start_index=starts).permute(1, 0, 2)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    with torch.no_grad():                                           # memory kept batch-first, as our transformer hands it over
        got_bf = ours(x.permute(1, 0, 2), key=None, value=memory, query_pos=qpos.permute(1, 0, 2), key_padding_mask=mask,
                      reference_points=ref, spatial_shapes=SHAPES, level_start_index=starts,
                      value_batch_first=True).permute(1, 0, 2)
    torch.testing.assert_close(got_bf, want, rtol=2e-4, atol=2e-4)
