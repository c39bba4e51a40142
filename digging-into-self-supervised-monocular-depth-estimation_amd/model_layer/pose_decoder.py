"""Pose heads (behaviour of reference model_layer/pose_decoder.py:13-98); parameter names `net.<n>.{weight,bias}`."""
import torch
import torch.nn as nn


class PoseDecoder(nn.Module):
    """Encoder features -> (axisangle, translation), each [B, frames, 1, 3], scaled by 0.01.
    1x1 squeeze to 256 -> two 3x3 (ReLU) -> 1x1 to 6*frames -> spatial mean."""

    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc, self.num_input_features = num_ch_enc, num_input_features
        self.num_frames_to_predict_for = num_frames_to_predict_for or (num_input_features - 1)
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),
            nn.Conv2d(256, 256, 3, stride, 1),
            nn.Conv2d(256, 6 * self.num_frames_to_predict_for, 1),
        ])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        squeezed = [self.relu(self.net[0](feats[-1])) for feats in input_features]
        x = torch.cat(squeezed, 1)
        x = self.relu(self.net[1](x))
        x = self.relu(self.net[2](x))
        x = self.net[3](x).float().mean(dim=(2, 3))
        x = 0.01 * x.view(-1, self.num_frames_to_predict_for, 1, 6)
        return x[..., :3], x[..., 3:]


class PoseCNN(nn.Module):
    """Seven strided convolutions on the stacked frames -> 1x1 pose head.  As in the reference, the first
    convolution is not followed by a ReLU (pose_decoder.py:86-89)."""

    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        spec = [(3 * num_input_frames, 16, 7), (16, 32, 5), (32, 64, 3), (64, 128, 3), (128, 256, 3), (256, 256, 3),
                (256, 256, 3)]
        self.net = nn.ModuleList([nn.Conv2d(ci, co, k, 2, k // 2) for ci, co, k in spec])
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)
        self.relu = nn.ReLU(True)

    def forward(self, input_images):
        x = self.net[0](input_images)
        for conv in list(self.net)[1:]:
            x = self.relu(conv(x))
        x = self.pose_conv(x).float().mean(dim=(2, 3))
        x = 0.01 * x.view(-1, self.num_input_frames - 1, 1, 6)
        return x[..., :3], x[..., 3:]
