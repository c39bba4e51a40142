"""U-Net disparity decoder (behaviour of reference model_layer/depth_decoder.py:13-112).

Five stages, deepest first; stage i = ConvBlock -> nearest x2 -> concat encoder skip (i > 0) -> ConvBlock, and a
sigmoid disparity head on the stages listed in `scales`.  The parameter names are the reference's
(`decoder.<n>.conv.conv.{weight,bias}`, n in the reference's ModuleList order: the ten stage blocks from stage 4
down to stage 0, then one head per scale), so checkpoints are interchangeable.  Convolutions run on MIOpen.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

STAGE_WIDTH = (16, 32, 64, 128, 256)


class Conv3x3(nn.Module):
    """3x3 convolution behind a one-pixel reflection (or zero) pad."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = (nn.ReflectionPad2d if use_refl else nn.ZeroPad2d)(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), kernel_size=3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """Conv3x3 followed by ELU."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.elu = nn.ELU(inplace=True)

    def forward(self, x):
        return self.elu(self.conv(x))


class DepthDecoder(nn.Module):
    fused_heads = True          # False: the heads as MIOpen convolution + bias + sigmoid (A/B, parity tests)
    thin_wgrad = True           # False: MIOpen's weight gradient for the 16-channel stage convolutions too

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array(STAGE_WIDTH)
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        blocks, self._stage = [], {}
        for i in (4, 3, 2, 1, 0):
            wide = int(self.num_ch_dec[i])
            fan_in = int(num_ch_enc[-1]) if i == 4 else int(self.num_ch_dec[i + 1])
            skip = int(num_ch_enc[i - 1]) if (use_skips and i > 0) else 0
            self._stage[i] = (len(blocks), len(blocks) + 1)
            blocks += [ConvBlock(fan_in, wide), ConvBlock(wide + skip, wide)]
        self._head = {}
        for s in scales:
            self._head[s] = len(blocks)
            blocks.append(Conv3x3(int(self.num_ch_dec[s]), num_output_channels))
        self.decoder = nn.ModuleList(blocks)
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        # csrc/glue.hip puts B*C on a 16-bit grid axis: (far) larger batches take the torch op sequence below
        if (input_features[-1].is_cuda and self._glue_ok()
                and input_features[-1].shape[0] * 2 * max(f.shape[1] for f in input_features) <= 65535):
            return self._forward_glue(input_features)
        self.outputs = {}
        x = input_features[-1]
        for i in (4, 3, 2, 1, 0):
            first, second = self._stage[i]
            x = TF.interpolate(self.decoder[first](x), scale_factor=2, mode="nearest")
            if self.use_skips and i > 0:
                x = torch.cat((x, input_features[i - 1]), 1)
            x = self.decoder[second](x)
            if i in self._head:
                # head + sigmoid stay float32 under bf16 autocast: the photometric kernels consume float32
                with torch.autocast(device_type=x.device.type, enabled=False):
                    self.outputs[("disp", i)] = self.sigmoid(self.decoder[self._head[i]](x.float()))
        return self.outputs

    def _glue_ok(self):
        return all(isinstance(m.pad, nn.ReflectionPad2d) for m in self.modules() if isinstance(m, Conv3x3))

    def _forward_glue(self, input_features):
        """Same network on the GPU with everything BETWEEN two convolutions -- ELU, nearest x2, skip concat,
        reflection pad -- as one hand-written pass (mdx.functional.decoder_glue, csrc/glue.hip): a convolution
        here produces the pre-activation map WITHOUT its bias and the next glue call adds the bias and applies the
        ELU on the way into the padded input of the next convolution (its backward also reduces d(bias)).  Parameters, their names and the results are those of forward()."""
        from mdx import functional as F
        from mdx.layout import to_layout, weight_layout
        self.outputs = {}
        cl = weight_layout(self)        # the decoder's own layout (mdx.layout): the encoder's maps are brought to it
        input_features = [to_layout(f, cl) for f in input_features]

        def conv(block, x):     # the stage convolution WITHOUT its bias: the glue call that consumes it adds it
            w = block.conv.conv.weight
            if self.thin_wgrad and torch.is_grad_enabled() and F.thin_conv_ok(x, w):
                return F.thin_conv3x3(x, w), block.conv.conv.bias      # 16 output channels: weight gradient by csrc/thinconv_nhwc.hip
            return TF.conv2d(x, w, None), block.conv.conv.bias
        padded = F.decoder_glue(input_features[-1], None, elu=False, upsample=False)
        for i in (4, 3, 2, 1, 0):
            first, second = self._stage[i]
            raw, bias = conv(self.decoder[first], padded)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            raw, bias = conv(self.decoder[second], F.decoder_glue(raw, skip, elu=True, upsample=True, bias=bias))
            padded = F.decoder_glue(raw, None, elu=True, upsample=False, bias=bias) if i > 0 else None
            if i in self._head:
                head = self.decoder[self._head[i]].conv
                # The hand-written head accumulates in float32 and writes float32 whatever its input's dtype: under bf16 autocast it
                # reads the bf16 padded map the next stage reads anyway (scale 0: a bf16 one of its own).  The MIOpen form keeps head
                # + sigmoid in float32 on a float32 copy of the map: the photometric kernels consume float32.
                if self.fused_heads and raw.dtype != torch.float32 and cl:
                    head_in = padded if padded is not None else F.decoder_glue(raw, None, elu=True, upsample=False, bias=bias)
                else:
                    head_in = padded if (padded is not None and padded.dtype == torch.float32) else \
                        F.decoder_glue(raw, None, elu=True, upsample=False, out_dtype=torch.float32, bias=bias)
                if self.fused_heads and F.disp_head_ok(head_in, head.weight):
                    # one output channel is no matrix-core problem: convolution + bias + sigmoid in one hand-written launch,
                    # data / weight / bias gradient in one more (csrc/disp_head_nhwc.hip)
                    self.outputs[("disp", i)] = F.disp_head(head_in, head.weight, head.bias)
                else:
                    with torch.autocast(device_type="cuda", enabled=False):
                        self.outputs[("disp", i)] = self.sigmoid(head(head_in))
        return self.outputs
