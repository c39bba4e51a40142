"""Pose heads (reference: model_layer/pose_decoder.py:13-98).  State-dict keys `net.N.{weight,bias}`."""
from collections import OrderedDict

import torch
import torch.nn as nn


class PoseDecoder(nn.Module):
    """reference: pose_decoder.py:13-58 -> (axisangle, translation), each [B, n_frames, 1, 3]."""

    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(int(self.num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        last_features = [f[-1] for f in input_features]
        cat_features = torch.cat([self.relu(self.convs["squeeze"](f)) for f in last_features], 1)
        out = cat_features
        for i in range(3):
            out = self.convs[("pose", i)](out)
            if i != 2:
                out = self.relu(out)
        out = out.float().mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]


class PoseCNN(nn.Module):
    """reference: pose_decoder.py:62-98 (the first conv is not followed by a ReLU there either)."""

    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        chans = [(3 * num_input_frames, 16, 7, 3), (16, 32, 5, 2), (32, 64, 3, 1), (64, 128, 3, 1),
                 (128, 256, 3, 1), (256, 256, 3, 1), (256, 256, 3, 1)]
        self.convs = {i: nn.Conv2d(ci, co, k, 2, p) for i, (ci, co, k, p) in enumerate(chans)}
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)
        self.num_convs = len(self.convs)
        self.relu = nn.ReLU(True)
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_images):
        output = self.convs[0](input_images)
        for index in range(self.num_convs - 1):
            output = self.relu(self.convs[index + 1](output))
        output = self.pose_conv(output).float().mean(3).mean(2)
        output = 0.01 * output.view(-1, self.num_input_frames - 1, 1, 6)
        return output[..., :3], output[..., 3:]
