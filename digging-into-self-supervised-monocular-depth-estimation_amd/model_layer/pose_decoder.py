"""Pose heads (behaviour of reference model_layer/pose_decoder.py:13-98); parameter names `net.<n>.{weight,bias}`."""
import torch
import torch.nn as nn
import torch.nn.functional as TF


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

    fused_tail = True        # False: every convolution with its bias + torch's ReLU / mean (A/B, parity tests)

    def _conv(self, k, x, relu=True):
        """net[k] + ReLU.  On a channels-last GPU map the convolution runs WITHOUT its bias and one hand-written pass adds the bias
        and applies the ReLU (its backward also reduces the bias gradient): mdx.functional.bias_act, csrc/pose_head_nhwc.hip."""
        conv = self.net[k]
        if self.fused_tail and conv.bias is not None:
            from mdx import functional as F
            from mdx.layout import is_channels_last
            if x.is_cuda and is_channels_last(conv.weight if conv.kernel_size != (1, 1) else x):
                y = TF.conv2d(x, conv.weight, None, conv.stride, conv.padding)
                if F.bias_act_ok(y):
                    return F.bias_act(y, conv.bias, relu=relu)
                y = y + conv.bias.to(y.dtype).view(1, -1, 1, 1)
                return self.relu(y) if relu else y
        y = conv(x)
        return self.relu(y) if relu else y

    def forward(self, input_features):
        squeezed = [self._conv(0, feats[-1]) for feats in input_features]
        x = torch.cat(squeezed, 1) if len(squeezed) > 1 else squeezed[0]
        x = self._conv(1, x)
        x = self._conv(2, x)
        head = self.net[3]
        if self.fused_tail and x.is_cuda and head.bias is not None:
            from mdx import functional as F
            from mdx.layout import is_channels_last
            y = TF.conv2d(x, head.weight, None, head.stride, head.padding)
            if is_channels_last(y) and y.dtype in (torch.float32, torch.bfloat16):
                # spatial mean, bias and the 0.01 in one launch (mean(conv + b) = mean(conv) + b)
                x = F.mean_bias(y, head.bias, scale=0.01).view(-1, self.num_frames_to_predict_for, 1, 6)
                return x[..., :3], x[..., 3:]
            x = (y + head.bias.to(y.dtype).view(1, -1, 1, 1)).float().mean(dim=(2, 3))
        else:
            x = head(x).float().mean(dim=(2, 3))
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
