"""ChannelMapper neck (/root/reference/mmdet/models/necks/channel_mapper.py:10-100): one
k x k conv + norm per input level and ``num_outs - len(in_channels)`` extra 3x3 stride-2
convs on the last map.  Sub-module names follow ext-mmcv ``ConvModule`` (``conv``, ``gn``)
so reference checkpoints load."""
import torch.nn as nn

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
        x = self.conv(x)
        if self.norm_name:
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
