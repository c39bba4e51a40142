"""U-Net disparity decoder (reference: model_layer/depth_decoder.py:13-112).  State-dict keys
`decoder.N.conv.conv.weight` in the reference's ModuleList order.  Convs run on MIOpen."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF


def upsample(tensor):
    return TF.interpolate(tensor, scale_factor=2, mode="nearest")


class Conv3x3(nn.Module):
    """reference: depth_decoder.py:36-50 (reflection or zero pad + 3x3 conv)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, inputs):
        return self.conv(self.pad(inputs))


class ConvBlock(nn.Module):
    """reference: depth_decoder.py:18-32 (Conv3x3 + ELU)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.elu = nn.ELU(inplace=True)

    def forward(self, inputs):
        return self.elu(self.conv(inputs))


class DepthDecoder(nn.Module):
    """reference: depth_decoder.py:54-112.  features (5 maps) -> {("disp", s): sigmoid [B,1,H>>s,W>>s]}."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.convs = OrderedDict()
        for index in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if index == 4 else self.num_ch_dec[index + 1]
            self.convs[("upconv", index, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[index])
            num_ch_in = self.num_ch_dec[index]
            if self.use_skips and index > 0:
                num_ch_in += self.num_ch_enc[index - 1]
            self.convs[("upconv", index, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[index])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        self.outputs = {}
        feature = input_features[-1]
        for index in range(4, -1, -1):
            feature = self.convs[("upconv", index, 0)](feature)
            feature = [upsample(feature)]
            if self.use_skips and index > 0:
                feature += [input_features[index - 1]]
            feature = torch.cat(feature, 1)
            feature = self.convs[("upconv", index, 1)](feature)
            if index in self.scales:
                # the disparity head and sigmoid stay float32 even under bf16 autocast: the photometric
                # kernels consume float32 disparity
                with torch.autocast(device_type=feature.device.type, enabled=False):
                    self.outputs[("disp", index)] = self.sigmoid(self.convs[("dispconv", index)](feature.float()))
        return self.outputs
