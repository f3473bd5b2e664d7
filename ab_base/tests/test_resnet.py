"""ResNet-50 backbone (``mmdet/models/backbones/resnet.py:631-659``, which needs
ext-mmcv to run) against an independent implementation: ``transformers``' ``ResNetBackbone`` (installed
offline, random weights, torchvision-style v1.5 bottleneck = mmdet ``style='pytorch'``: the stride sits
on the 3x3 convolution).  The HF parameters are mapped onto our reference-named ones (``conv1 / bn1 /
layer{i}.{j}.conv{k} / bn{k} / downsample.{0,1}``); C3-C5 must agree -- also through our frozen-BN
fold (BN in eval mode folded into the convolution weights, bias + ReLU epilogue)."""
import pytest
import torch

import dskd_amd  # noqa: F401
from dskd_amd.backbones import ResNet


def _hf_to_ours(hf_sd):
    out = {}
    bn = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    out["conv1.weight"] = hf_sd["embedder.embedder.convolution.weight"]
    for k in bn:
        out[f"bn1.{k}"] = hf_sd[f"embedder.embedder.normalization.{k}"]
    for i, depth in enumerate((3, 4, 6, 3)):
        for j in range(depth):
            s, d = f"encoder.stages.{i}.layers.{j}.", f"layer{i + 1}.{j}."
            for c in range(3):
                out[f"{d}conv{c + 1}.weight"] = hf_sd[f"{s}layer.{c}.convolution.weight"]
                for k in bn:
                    out[f"{d}bn{c + 1}.{k}"] = hf_sd[f"{s}layer.{c}.normalization.{k}"]
            if j == 0:
                out[f"{d}downsample.0.weight"] = hf_sd[f"{s}shortcut.convolution.weight"]
                for k in bn:
                    out[f"{d}downsample.1.{k}"] = hf_sd[f"{s}shortcut.normalization.{k}"]
    return out


def _resnet_names():
    """our ResNet parameter name -> HF ResNet parameter name (convolution weights only)."""
    out = {"conv1.weight": "embedder.embedder.convolution.weight"}
    for i, depth in enumerate((3, 4, 6, 3)):
        for j in range(depth):
            for c in range(3):
                out[f"layer{i + 1}.{j}.conv{c + 1}.weight"] = f"encoder.stages.{i}.layers.{j}.layer.{c}.convolution.weight"
            if j == 0:
                out[f"layer{i + 1}.{j}.downsample.0.weight"] = f"encoder.stages.{i}.layers.{j}.shortcut.convolution.weight"
    return out


@pytest.mark.parametrize("hw", [(96, 128), (75, 101)])          # the second: odd sizes through the strided convs / max-pool
def test_resnet50_matches_transformers_backbone(hw):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.ResNetConfig(out_features=["stage2", "stage3", "stage4"])
    assert cfg.layer_type == "bottleneck" and list(cfg.depths) == [3, 4, 6, 3] and not cfg.downsample_in_bottleneck
    torch.manual_seed(0)
    hf = transformers.ResNetBackbone(cfg).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                       # non-trivial BN statistics / affine parameters
        for name, buf in hf.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.75)
        for name, p in hf.named_parameters():
            if "normalization" in name:
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75 if name.endswith("weight")
                        else torch.randn(p.shape, generator=g) * 0.1)
    ours = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True, style="pytorch")
    missing, unexpected = ours.load_state_dict(_hf_to_ours(hf.state_dict()), strict=True)
    assert not missing and not unexpected
    x = torch.randn(2, 3, *hw, generator=g)
    with torch.no_grad():
        want = hf(x).feature_maps
    for mode in ("eval", "train"):                              # train(): BN stays in eval (norm_eval), stages <= 1 frozen
        getattr(ours, mode)()
        with torch.no_grad():
            got = ours(x)
        assert len(got) == 3
        for a, b in zip(got, want):
            assert a.shape == b.shape
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))
    # the frozen part carries no gradient, the rest does
    ours.train()
    ours(x)[-1].sum().backward()
    assert ours.conv1.weight.grad is None and ours.layer1[0].conv1.weight.grad is None
    assert ours.layer2[0].conv1.weight.grad is not None and ours.layer4[2].conv3.weight.grad is not None


def test_resnet50_gradients_match_transformers_in_float64():
    """Training mode, float64: gradients of a scalar of C3-C5 w.r.t. every trainable convolution weight
    (stages 2-4; our BN-fold backward multiplies the folded-weight gradient by the frozen scale) agree
    with autograd through the independent implementation to rounding; stem and stage 1 get none."""
    transformers = pytest.importorskip("transformers")
    cfg = transformers.ResNetConfig(out_features=["stage2", "stage3", "stage4"])
    torch.manual_seed(0)
    hf = transformers.ResNetBackbone(cfg).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, buf in hf.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.75)
    ours = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True, style="pytorch")
    ours.load_state_dict(_hf_to_ours(hf.state_dict()), strict=True)
    hf, ours = hf.double(), ours.double().train()
    x = torch.randn(2, 3, 64, 96, generator=g).double()
    x[1, :, 40:, :] = 0                                          # a padded image
    want = hf(x).feature_maps
    ws = [torch.randn(f.shape, generator=g).double() for f in want]
    sum((a * w).sum() for a, w in zip(want, ws)).backward()
    sum((a * w).sum() for a, w in zip(ours(x), ws)).backward()
    hp, op = dict(hf.named_parameters()), dict(ours.named_parameters())
    checked = 0
    for on, hn in _resnet_names().items():
        if on.startswith(("conv1", "layer1")):
            assert op[on].grad is None
            continue
        scale = float(hp[hn].grad.abs().max())
        assert float((op[on].grad - hp[hn].grad).abs().max()) <= 1e-10 * scale, on
        checked += 1
    assert checked == 42                                          # stages 2-4: (4 + 6 + 3) blocks x 3 convolutions + 3 shortcuts
