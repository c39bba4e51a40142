"""ResNet encoder (reference: model_layer/depth_encoder.py:14-101), restated without torchvision.

State-dict compatible with the reference / torchvision (`encoder.conv1.weight`,
`encoder.layer1.0.conv1.weight`, ..., `encoder.fc.weight`).  The convolutions run on PyTorch-ROCm
(MIOpen / hipBLASLt -> MFMA); no hand kernel here by design.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF



class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose `num_batches_tracked` counter lives on the host while training.

    The stock module launches one tiny add kernel per layer and step for that counter (60 launches per training
    step of this path), and with a fixed momentum nothing ever reads it.  Same parameters, buffers and state-dict
    keys; the buffer is brought up to date whenever the state dict is taken.  momentum=None (cumulative average,
    which does read the counter) keeps the stock behaviour."""

    fused_min_elements = 0          # per channel (B*H*W): below it act() takes the torch ops (a test / A-B switch)
    _batch_groups = 1               # default of every instance; batch_groups() sets it on the modules of ONE network

    @staticmethod
    def batch_groups(groups, network):
        """Context manager: inside it every BatchNorm2d OF `network` treats its input batch as `groups` consecutive
        sub-batches, each with its own batch statistics and its own running-statistics update -- the result of calling the
        network once per sub-batch, from ONE pass of the convolutions over the whole batch.  Per instance: no other
        network (and no other thread's forward) is affected."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            mods = [m for m in network.modules() if isinstance(m, BatchNorm2d)]
            prev = [m.__dict__.get("_batch_groups") for m in mods]
            for m in mods:
                m._batch_groups = int(groups)
            try:
                yield
            finally:
                for m, p in zip(mods, prev):
                    if p is None:
                        m.__dict__.pop("_batch_groups", None)
                    else:
                        m._batch_groups = p
        return scope()

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pending_batches = 0

    def forward(self, x):
        if self.momentum is None or not self.track_running_stats:
            return super().forward(x)
        groups = self._batch_groups if self.training else 1
        if groups > 1:
            return torch.cat([self._norm(c) for c in x.chunk(groups)])
        return self._norm(x)

    def _norm(self, x):
        if self.training:
            self._pending_batches += 1
        return TF.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                             self.momentum, self.eps)

    def act(self, x, residual=None, relu=True, fork=False):
        """relu(self(x) + residual) -- on the GPU in training mode as the fused kernels of csrc/norm.hip (planar maps) /
        csrc/norm_nhwc.hip (channels-last maps): statistics pass + normalise/add/ReLU pass, backward likewise; else as the
        torch ops.  fork: the result twice (mdx.functional.bn_act), one tensor per consumer."""
        # csrc/norm.hip: one launch each way for maps up to 24 K elements per channel (kept in registers between the
        # reduction and the apply step), two (statistics pass, apply pass) above -- tools/normbench.py
        fused = (self.training and x.is_cuda and self.track_running_stats and self.momentum is not None
                 and self.affine and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4
                 and torch.is_grad_enabled() and x.shape[0] % self._batch_groups == 0
                 and x.shape[0] * x.shape[2] * x.shape[3] >= self.fused_min_elements
                 # csrc/norm.hip puts (images of a group) x (8 K-element spans of a plane) on a 16-bit grid axis
                 and (x.shape[0] // self._batch_groups) * ((x.shape[2] * x.shape[3] + 8191) // 8192) <= 65535)
        if not fused:
            out = self(x)
            if residual is not None:
                out = out + residual
            out = TF.relu(out) if relu else out
            return (out, out) if fork else out
        from mdx import functional as F
        self._pending_batches += self._batch_groups
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        return F.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                        residual=residual, relu=relu, groups=self._batch_groups, fork=fork)

    def _flush_counter(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

def _pair(x):
    """A block's input: (for the first convolution, for the identity path) -- two tensors on one storage when the
    producer forked its output (BatchNorm2d.act(fork=True)), else the same tensor twice."""
    return x if isinstance(x, tuple) else (x, x)


class BasicBlock(nn.Module):
    expansion = 1
    fork_output = True      # hand the output on as a pair: the producer's backward then adds its two gradients itself

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        x, xid = _pair(x)
        identity = xid if self.downsample is None else self.downsample[1].act(self.downsample[0](xid), relu=False)
        out = self.bn1.act(self.conv1(x))
        return self.bn2.act(self.conv2(out), residual=identity, fork=self.fork_output)


class Bottleneck(nn.Module):
    expansion = 4
    fork_output = True

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        x, xid = _pair(x)
        identity = xid if self.downsample is None else self.downsample[1].act(self.downsample[0](xid), relu=False)
        out = self.bn1.act(self.conv1(x))
        out = self.bn2.act(self.conv2(out))
        return self.bn3.act(self.conv3(out), residual=identity, fork=self.fork_output)


_CFG = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
        101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResNet(nn.Module):
    """torchvision.models.ResNet layout (conv1/bn1/relu/maxpool/layer1-4/avgpool/fc)."""

    def __init__(self, block, layers, num_classes=1000, num_input_images=1):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _load_pretrained(model, num_layers, num_input_images):
    """ImageNet init (reference: depth_encoder.py:57-60,86).  There is no network here: weights are
    read from $MDX_RESNET_WEIGHTS/resnet{N}.pth when present, otherwise the random init stays."""
    path = os.path.join(os.environ.get("MDX_RESNET_WEIGHTS", ""), "resnet%d.pth" % num_layers)
    if not os.path.isfile(path):
        warnings.warn("pretrained=True but %s not found (offline): using random init" % (path or "weights"))
        return
    loaded = torch.load(path, map_location="cpu")
    if num_input_images > 1:
        loaded["conv1.weight"] = torch.cat([loaded["conv1.weight"]] * num_input_images, 1) / num_input_images
    model.load_state_dict(loaded)


def resnet_multiimage_input(num_layers, pretrained=True, num_input_images=1):
    """reference: model_layer/depth_encoder.py:44-61."""
    assert num_layers in [18, 50], "Can only run with 18 or 50 layer resnet"
    block, layers = _CFG[num_layers]
    model = ResNet(block, layers, num_input_images=num_input_images)
    if pretrained:
        _load_pretrained(model, num_layers, num_input_images)
    return model


class ResnetEncoder(nn.Module):
    """reference: model_layer/depth_encoder.py:65-101.  Returns the five feature maps."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if num_input_images > 1:
            self.encoder = resnet_multiimage_input(num_layers, pretrained, num_input_images)
        else:
            block, layers = _CFG[num_layers]
            self.encoder = ResNet(block, layers)
            if pretrained:
                _load_pretrained(self.encoder, num_layers, 1)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    fused_input = True       # False: (x - 0.45) / 0.225 as torch ops on the concatenated frames (A/B, parity tests)

    def _normalised_input(self, input_image):
        """(x - 0.45) / 0.225 (reference depth_encoder.py:89).  x: a tensor, or the step's not-yet-concatenated frame pairs
        (mdx.functional.FrameStack).  With a channels-last stem on the GPU the frames go through ONE pass that normalises and
        writes the channels-last map the first convolution reads (bf16 under autocast: the cast it would do itself)."""
        from mdx import functional as F
        from mdx.layout import weight_layout
        stack = input_image if isinstance(input_image, F.FrameStack) else None
        if stack is None and torch.is_tensor(input_image) and input_image.dim() == 4 and input_image.shape[1] == 3:
            stack = F.FrameStack([[input_image]])
        if (self.fused_input and stack is not None and stack.ok() and weight_layout(self.encoder.conv1)
                and stack.blocks[0][0].numel() * len(stack.blocks) * len(stack.blocks[0]) < (1 << 31)):
            dt = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else torch.float32
            if dt in (torch.float32, torch.bfloat16):
                return F.encoder_input(stack, 0.45, 0.225, dt)
        if isinstance(input_image, F.FrameStack):
            input_image = input_image.tensor()
        return (input_image - 0.45) / 0.225

    def forward(self, input_image):
        from mdx.layout import to_layout, weight_layout
        self.features = []
        x = self._normalised_input(input_image)
        # every map below has two consumers (the next layer's first convolution + its identity path, or the decoder's
        # skip connection + the next layer): producers hand their output on as a pair, see BatchNorm2d.act(fork=True)
        stem, stem_b = _pair(self.encoder.bn1.act(self.encoder.conv1(x), fork=BasicBlock.fork_output))
        self.features.append(stem)
        if (stem.is_cuda and stem.dtype in (torch.float32, torch.bfloat16)
                and stem.shape[0] * stem.shape[1] <= 65535):         # csrc/glue.hip: B*C on a 16-bit grid axis
            from mdx import functional as F      # gather-based backward instead of ATen's atomics (csrc/glue.hip)
            x = F.maxpool3s2(stem_b, fork=BasicBlock.fork_output)
        else:
            x = self.encoder.maxpool(stem_b)
        for layer in (self.encoder.layer1, self.encoder.layer2, self.encoder.layer3, self.encoder.layer4):
            # a stage's layout is its weights' layout (mdx.layout): a transposing copy only where two stages differ
            x = _pair(layer(to_layout(x, weight_layout(layer)) if x[0].is_cuda else x))
            self.features.append(x[0])
        return self.features
