"""ChannelMapper neck (/root/reference/mmdet/models/necks/channel_mapper.py:10-100): one
k x k conv + norm per input level and ``num_outs - len(in_channels)`` extra 3x3 stride-2
convs on the last map.  Sub-module names follow ext-mmcv ``ConvModule`` (``conv``, ``gn``)
so reference checkpoints load."""
import torch
import torch.nn as nn

from . import native
from .builder import NECKS


class ConvModule(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, norm_cfg=None, act_cfg=None):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=norm_cfg is None)
        self.norm_name = None
        if norm_cfg is not None:
            t = norm_cfg["type"]
            if t == "GN":
                self.norm_name = "gn"
                self.add_module("gn", nn.GroupNorm(norm_cfg["num_groups"], cout))
            elif t == "BN":
                self.norm_name = "bn"
                self.add_module("bn", nn.BatchNorm2d(cout))
            else:
                raise KeyError(f"norm type {t} not supported")
        self.activate = nn.ReLU(inplace=True) if act_cfg is not None else None

    def forward(self, x):
        w = self.conv.weight
        if x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 \
                and self.conv.kernel_size == (1, 1) and x.dtype == torch.bfloat16:
            wl = w.to(torch.bfloat16)
            if native.conv1x1_ok(x, wl, self.conv):       # lateral 1x1 convolution on the MFMA GEMM (csrc/gemm_nt.hip)
                x = native.conv1x1(x, wl, self.conv.bias, None, False, self.conv.stride[0])
            else:
                x = self.conv(x)
        else:
            x = self.conv(x)
        if self.norm_name == "gn" and native.group_norm_cl_ok(x, self.gn):
            # channels_last GroupNorm(32, 256), with the ReLU of the GFL head's towers folded in: csrc/gn.hip
            return native.group_norm_cl(x, self.gn, relu=isinstance(self.activate, nn.ReLU))
        elif self.norm_name:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


@NECKS.register_module()
class ChannelMapper(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), num_outs=None, init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        self.extra_convs = None
        if num_outs is None:
            num_outs = len(in_channels)
        self.convs = nn.ModuleList(
            ConvModule(c, out_channels, kernel_size, padding=(kernel_size - 1) // 2, norm_cfg=norm_cfg, act_cfg=act_cfg)
            for c in in_channels)
        if num_outs > len(in_channels):
            self.extra_convs = nn.ModuleList()
            for i in range(len(in_channels), num_outs):
                cin = in_channels[-1] if i == len(in_channels) else out_channels
                self.extra_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg,
                                                   act_cfg=act_cfg))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):   # init_cfg: Xavier uniform on Conv2d (channel_mapper.py:55-56)
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        outs = [self.convs[i](inputs[i]) for i in range(len(inputs))]
        if self.extra_convs:
            for i, conv in enumerate(self.extra_convs):
                outs.append(conv(inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)


@NECKS.register_module()
class FPN(nn.Module):
    """Feature pyramid (/root/reference/mmdet/models/necks/fpn.py:10-204): 1x1 lateral convs on
    ``in_channels[start_level:end_level]``, top-down nearest-neighbour upsampling + add, a 3x3 conv per level, and
    ``num_outs - used_levels`` extra stride-2 3x3 convs fed by the last input (``'on_input'``), the last lateral
    (``'on_lateral'``) or the last output (``'on_output'``, the GFL configs).  Sub-module names (``lateral_convs.i.conv``,
    ``fpn_convs.i.conv``) follow the reference, so its checkpoints load."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        self.in_channels, self.out_channels, self.num_ins, self.num_outs = list(in_channels), out_channels, len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1 or end_level == self.num_ins - 1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level + 1
            assert end_level < self.num_ins and num_outs == end_level - start_level + 1
        self.start_level, self.end_level = start_level, end_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ("on_input", "on_lateral", "on_output")
        elif add_extra_convs:
            add_extra_convs = "on_input"
        self.add_extra_convs = add_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg, act_cfg=act_cfg))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == "on_input") \
                    else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):   # init_cfg: Xavier uniform on Conv2d (fpn.py:77-78)
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        import torch.nn.functional as F
        assert len(inputs) == len(self.in_channels)
        laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        used = len(laterals)
        for i in range(used - 1, 0, -1):
            if "scale_factor" in self.upsample_cfg:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], **self.upsample_cfg)
            else:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(used)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - used):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                if self.add_extra_convs == "on_input":
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == "on_lateral":
                    src = laterals[-1]
                else:
                    src = outs[-1]
                outs.append(self.fpn_convs[used](src))
                for i in range(used + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)
