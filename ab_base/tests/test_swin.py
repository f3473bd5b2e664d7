"""Swin backbone against an independent implementation: ``transformers``'
``SwinBackbone`` (installed offline).  Same architecture, different code; the weights of a random
HF model are mapped onto our reference-named parameters.  The only layout difference is the
channel order of patch merging: the reference (``nn.Unfold``) is channel-major ``c*4 + kh*2 + kw``,
the original / HF order is position-major ``[(0,0), (1,0), (0,1), (1,1)] x C``."""
import pytest
import torch

import dskd_amd  # noqa: F401
from dskd_amd.builder import BACKBONES
from dskd_amd.swin import SwinTransformer


def _hf_to_ours(hf_sd, depths):
    hf_sd = {(k[5:] if k.startswith("swin.") else k): v for k, v in hf_sd.items()}
    out = {}
    pre = "embeddings.patch_embeddings.projection."
    out["patch_embed.projection.weight"] = hf_sd[pre + "weight"]
    out["patch_embed.projection.bias"] = hf_sd[pre + "bias"]
    out["patch_embed.norm.weight"] = hf_sd["embeddings.norm.weight"]
    out["patch_embed.norm.bias"] = hf_sd["embeddings.norm.bias"]
    for i, depth in enumerate(depths):
        for j in range(depth):
            s, d = f"encoder.layers.{i}.blocks.{j}.", f"stages.{i}.blocks.{j}."
            out[d + "norm1.weight"], out[d + "norm1.bias"] = hf_sd[s + "layernorm_before.weight"], hf_sd[s + "layernorm_before.bias"]
            out[d + "norm2.weight"], out[d + "norm2.bias"] = hf_sd[s + "layernorm_after.weight"], hf_sd[s + "layernorm_after.bias"]
            a = s + "attention."
            out[d + "attn.w_msa.relative_position_bias_table"] = hf_sd[a + "relative_position_bias.relative_position_bias_table"]
            out[d + "attn.w_msa.qkv.weight"] = torch.cat([hf_sd[a + f"{n}_proj.weight"] for n in "qkv"], 0)
            out[d + "attn.w_msa.qkv.bias"] = torch.cat([hf_sd[a + f"{n}_proj.bias"] for n in "qkv"], 0)
            out[d + "attn.w_msa.proj.weight"], out[d + "attn.w_msa.proj.bias"] = hf_sd[a + "o_proj.weight"], hf_sd[a + "o_proj.bias"]
            out[d + "ffn.layers.0.0.weight"], out[d + "ffn.layers.0.0.bias"] = hf_sd[s + "mlp.fc1.weight"], hf_sd[s + "mlp.fc1.bias"]
            out[d + "ffn.layers.1.weight"], out[d + "ffn.layers.1.bias"] = hf_sd[s + "mlp.fc2.weight"], hf_sd[s + "mlp.fc2.bias"]
        if i < len(depths) - 1:
            s, d = f"encoder.layers.{i}.downsample.", f"stages.{i}.downsample."
            C = hf_sd[s + "norm.weight"].numel() // 4

            def reorder(t):        # [..., pos*C + c] (HF) -> [..., c*4 + kh*2 + kw] (reference)
                t4 = t.reshape(*t.shape[:-1], 4, C)            # pos: 0 (0,0), 1 (1,0), 2 (0,1), 3 (1,1)
                t4 = t4[..., [0, 2, 1, 3], :]                   # -> kh*2 + kw
                return t4.transpose(-1, -2).reshape(*t.shape[:-1], 4 * C)
            out[d + "norm.weight"], out[d + "norm.bias"] = reorder(hf_sd[s + "norm.weight"]), reorder(hf_sd[s + "norm.bias"])
            out[d + "reduction.weight"] = reorder(hf_sd[s + "reduction.weight"])
    for i in range(len(depths)):
        out[f"norm{i}.weight"] = hf_sd[f"hidden_states_norms.stage{i + 1}.weight"]
        out[f"norm{i}.bias"] = hf_sd[f"hidden_states_norms.stage{i + 1}.bias"]
    return out


@pytest.mark.parametrize("hw", [(224, 224), (250, 331)])       # the second: patch, merge and window padding
def test_swin_matches_transformers_backbone(hw):
    from transformers import SwinBackbone, SwinConfig
    torch.manual_seed(0)
    depths, heads, embed = (2, 2, 2, 2), (2, 4, 8, 16), 32
    cfg = SwinConfig(image_size=224, patch_size=4, num_channels=3, embed_dim=embed, depths=list(depths), num_heads=list(heads),
                     window_size=7, mlp_ratio=4.0, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     drop_path_rate=0.0, hidden_act="gelu", use_absolute_embeddings=False, layer_norm_eps=1e-5,
                     out_features=["stage1", "stage2", "stage3", "stage4"])
    hf = SwinBackbone(cfg).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():                        # non-trivial biases / norms / bias tables
            if p.dim() == 1 or "relative_position_bias_table" in n:
                p.add_(torch.randn_like(p) * 0.2)
    ours = SwinTransformer(embed_dims=embed, depths=depths, num_heads=heads, window_size=7, drop_path_rate=0.0).eval()
    missing, unexpected = ours.load_state_dict(_hf_to_ours(hf.state_dict(), depths), strict=False)
    assert not unexpected and all("relative_position_index" in k for k in missing), (missing, unexpected)
    x = torch.randn(2, 3, *hw)
    with torch.no_grad():
        ref = hf(x).feature_maps
        out = ours(x)
    assert len(out) == 4
    for a, b in zip(out, ref):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)


def test_official_checkpoint_conversion(tmp_path):
    """``convert_weights=True``: a checkpoint in the original Swin layout (here: written from the
    HF weights, whose patch merging is position-major like the original) initialises the backbone
    to the same function."""
    from transformers import SwinBackbone, SwinConfig
    torch.manual_seed(1)
    depths, heads, embed = (2, 2, 2, 2), (2, 4, 8, 16), 32
    hf = SwinBackbone(SwinConfig(embed_dim=embed, depths=list(depths), num_heads=list(heads), window_size=7,
                                 drop_path_rate=0.0, out_features=["stage1", "stage2", "stage3", "stage4"])).eval()
    sd = {(k[5:] if k.startswith("swin.") else k): v for k, v in hf.state_dict().items()}
    official = {"patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
                "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
                "patch_embed.norm.weight": sd["embeddings.norm.weight"], "patch_embed.norm.bias": sd["embeddings.norm.bias"],
                "head.weight": torch.zeros(10, 8 * embed)}
    for i, depth in enumerate(depths):
        for j in range(depth):
            s_, d_ = f"encoder.layers.{i}.blocks.{j}.", f"layers.{i}.blocks.{j}."
            a = s_ + "attention."
            official[d_ + "attn.relative_position_bias_table"] = sd[a + "relative_position_bias.relative_position_bias_table"]
            official[d_ + "attn.qkv.weight"] = torch.cat([sd[a + f"{n}_proj.weight"] for n in "qkv"], 0)
            official[d_ + "attn.qkv.bias"] = torch.cat([sd[a + f"{n}_proj.bias"] for n in "qkv"], 0)
            official[d_ + "attn.proj.weight"], official[d_ + "attn.proj.bias"] = sd[a + "o_proj.weight"], sd[a + "o_proj.bias"]
            for n, m in (("norm1", "layernorm_before"), ("norm2", "layernorm_after")):
                official[d_ + n + ".weight"], official[d_ + n + ".bias"] = sd[s_ + m + ".weight"], sd[s_ + m + ".bias"]
            for n in ("fc1", "fc2"):
                official[d_ + f"mlp.{n}.weight"], official[d_ + f"mlp.{n}.bias"] = sd[s_ + f"mlp.{n}.weight"], sd[s_ + f"mlp.{n}.bias"]
        if i < len(depths) - 1:
            for n in ("reduction.weight", "norm.weight", "norm.bias"):
                official[f"layers.{i}.downsample.{n}"] = sd[f"encoder.layers.{i}.downsample.{n}"]
    for i in range(4):
        official[f"norm{i}.weight"], official[f"norm{i}.bias"] = sd[f"hidden_states_norms.stage{i + 1}.weight"], sd[f"hidden_states_norms.stage{i + 1}.bias"]
    path = tmp_path / "swin_official.pth"
    torch.save({"model": official}, path)
    ours = SwinTransformer(embed_dims=embed, depths=depths, num_heads=heads, window_size=7, drop_path_rate=0.0,
                           pretrained=str(path), convert_weights=True).eval()
    ours.init_weights()
    x = torch.randn(1, 3, 160, 192)
    with torch.no_grad():
        for a, b in zip(ours(x), hf(x).feature_maps):
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("split", ["40_40", "70_10"])
def test_swin_detector_step_on_cpu(cpu_ops, split):
    """BASELINE configs[3] (Swin-T 40+40; 70+10 = the same trunk on the headline split): Swin-T + ChannelMapper([192, 384, 768]) + the DSKD head builds from
    the config file and runs one distillation step (CPU, oracle ops) with finite losses and
    gradients reaching the trainable backbone stages."""
    import copy
    import os
    from dskd_amd.builder import build_detector
    from dskd_amd.config import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", f"dskd_gfl_deformable_detr_swin_t_{split}.py"))
    assert (cfg.num_prev, cfg.num_curr) == tuple(int(v) for v in split.split("_"))
    if split == "40_40":      # configs[3] = the reference's 40+40 schedule (lr 4e-4) on the Swin-T trunk
        ref40 = Config.fromfile(os.path.join(root, "configs", "dskd_gfl_deformable_detr_r50_40_40.py"))
        assert cfg.optimizer[0]["lr"] == ref40.optimizer[0]["lr"] == 4e-4
        assert cfg.model.bbox_head.feats_distill == ref40.model.bbox_head.feats_distill
        assert cfg.model.bbox_head.cates_distill == ref40.model.bbox_head.cates_distill
    cfg.model.bbox_head.num_query = 30
    torch.manual_seed(0)
    m = build_detector(cfg.model)
    m.init_weights()
    t = copy.deepcopy(m)
    m.set_teacher(model=t)
    m.LableInPCNTask = {"prev": list(range(cfg.num_prev)), "curr": list(range(cfg.num_prev, 80)), "next": []}
    m.train()
    img = torch.randn(1, 3, 128, 160)
    metas = [dict(img_shape=(128, 160, 3), batch_input_shape=(128, 160), scale_factor=1.0)]
    out = m.train_step(dict(img=img, img_metas=metas, gt_bboxes=[torch.tensor([[10., 12., 90., 100.]])],
                            gt_labels=[torch.tensor([75])]))
    assert all(v == v and abs(v) < 1e6 for v in out["log_vars"].values()), out["log_vars"]
    for k in ("loss_cls", "loss_bbox", "loss_iou", "loss_dfl", "d4.loss_cls", "loss"):
        assert k in out["log_vars"]
    out["loss"].backward()
    assert m.backbone.stages[3].blocks[1].ffn.layers[1].weight.grad is not None
    assert m.neck.convs[0].conv.weight.grad is not None


@pytest.mark.parametrize("shift", [0, 3])
def test_window_partition_gradients(shift, monkeypatch):
    """The window partition / reverse are token gathers with a hand-written (gather) backward:
    gradients must equal autograd's generic index backward, padded sizes included."""
    from dskd_amd import swin
    torch.manual_seed(3)
    att = swin.ShiftWindowMSA(32, 4, 7, shift_size=shift).eval()
    with torch.no_grad():
        att.w_msa.relative_position_bias_table.normal_(std=0.5)
    x = torch.randn(2, 9 * 13, 32)
    gy = torch.randn(2, 9 * 13, 32)
    xa = x.clone().requires_grad_(True)
    att(xa, (9, 13)).backward(gy)
    ga = [xa.grad.clone()] + [p.grad.clone() for p in att.parameters()]
    for p in att.parameters():
        p.grad = None

    class Plain:
        @staticmethod
        def apply(t, fwd, bwd, n_in):
            return t[:, fwd]
    monkeypatch.setattr(swin, "_TokenGather", Plain)
    xb = x.clone().requires_grad_(True)
    att(xb, (9, 13)).backward(gy)
    gb = [xb.grad] + [p.grad for p in att.parameters()]
    for a, b in zip(ga, gb):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_swin_registry_surface_and_freezing():
    """Constructor kwargs of the reference configs, state-dict names, frozen stages, gradients."""
    m = BACKBONES.build(dict(type="SwinTransformer", embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                             window_size=7, mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                             drop_path_rate=0.2, patch_norm=True, out_indices=(1, 2, 3), with_cp=False,
                             convert_weights=True, frozen_stages=1, init_cfg=None))
    m.init_weights()
    keys = set(m.state_dict())
    for k in ("patch_embed.projection.weight", "patch_embed.norm.bias", "stages.0.blocks.1.attn.w_msa.relative_position_index",
              "stages.2.blocks.5.attn.w_msa.qkv.weight", "stages.1.blocks.0.ffn.layers.0.0.weight",
              "stages.1.blocks.0.ffn.layers.1.bias", "stages.0.downsample.reduction.weight", "stages.2.downsample.norm.weight",
              "norm1.weight", "norm3.bias"):
        assert k in keys, k
    assert "norm0.weight" not in keys and "stages.3.downsample.reduction.weight" not in keys
    assert sum(p.numel() for p in m.parameters()) == 27_520_506               # Swin-T (27.5 M) with norm1..3
    m.train()
    assert not m.patch_embed.projection.weight.requires_grad and not m.stages[0].blocks[0].norm1.weight.requires_grad
    assert m.stages[1].blocks[0].norm1.weight.requires_grad and not m.stages[0].training and m.stages[1].training
    outs = m(torch.randn(1, 3, 96, 130))
    assert [o.shape[1] for o in outs] == [192, 384, 768] and outs[0].shape[-2:] == (12, 17)
    sum(o.float().pow(2).mean() for o in outs).backward()
    assert m.stages[3].blocks[1].ffn.layers[1].weight.grad is not None and m.stages[0].blocks[0].attn.w_msa.qkv.weight.grad is None


@pytest.mark.gpu
def test_swin_gpu_matches_cpu():
    """Same weights and input on the MI355X (fused attention kernels, hipBLASLt) and on the CPU,
    fp32; plus the bench configuration (bf16 autocast) within bf16 tolerance, with gradients."""
    torch.manual_seed(2)
    m = SwinTransformer(embed_dims=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), out_indices=(1, 2, 3),
                        drop_path_rate=0.0).eval()
    m.init_weights()
    x = torch.randn(2, 3, 250, 331)
    with torch.no_grad():
        ref = m(x)
    mg = m.to("cuda:0")
    with torch.no_grad():
        out = mg(x.to("cuda:0"))
    for a, b in zip(out, ref):
        torch.testing.assert_close(a.cpu(), b, rtol=2e-3, atol=2e-3)
    mg.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out16 = mg(x.to("cuda:0"))
    for a, b in zip(out16, ref):
        assert a.dtype == torch.float32                      # LayerNorm outputs
        torch.testing.assert_close(a.cpu(), b, rtol=5e-2, atol=8e-2)
    sum(o.pow(2).mean() for o in out16).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in mg.parameters() if p.requires_grad)
