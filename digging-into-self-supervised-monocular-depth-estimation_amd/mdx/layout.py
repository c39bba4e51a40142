"""Memory-layout plan of the networks: which stages keep their maps channels-last ([B][H][W][C] memory: MIOpen's
implicit-GEMM convolutions and csrc/*_nhwc.hip) and which planar ([B][C][H][W]: MIOpen's Winograd convolutions and
csrc/norm.hip / glue.hip).  A stage's layout is the layout of its convolution weights (torch picks a convolution's layout
from input OR weight, so maps are brought to the stage's layout where two stages of different layout meet).

A plan is "none", "all", or a comma list of stage names: stem, layer1..layer4 (of every ResNet encoder), decoder (the depth
decoder), pose (the pose decoder).  "auto" = DEFAULT_PLAN, what measured fastest on MI355X (network layout)."""
import torch

STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "decoder", "pose")
DEFAULT_PLAN = "all"
_CL = torch.channels_last


def parse_plan(plan):
    """-> the set of channels-last stages.  Accepts the historical booleans (True = "all")."""
    if plan in (None, False, "", "none", "0"):
        return frozenset()
    if plan is True or plan in ("1", "all"):
        return frozenset(STAGES)
    if plan == "auto":
        return parse_plan(DEFAULT_PLAN)
    names = [s.strip() for s in str(plan).split(",") if s.strip()]
    bad = [s for s in names if s not in STAGES]
    if bad:
        raise ValueError("channels_last plan: unknown stage(s) %s (stages: %s)" % (bad, ", ".join(STAGES)))
    return frozenset(names)


def _set(module, cl):
    module._mdx_channels_last = bool(cl)          # (weight_layout: a stage of 1x1 convolutions cannot tell from its strides)
    return module.to(memory_format=_CL if cl else torch.contiguous_format)


def apply_plan(models, plan):
    """models: {"encoder", "decoder", "pose_encoder", "pose_decoder", ...} (model_tool.loader.setting.model).  Sets the
    memory format of every convolution weight according to the plan, in place; returns the set of channels-last stages."""
    stages = parse_plan(plan)
    for key, m in models.items():
        if key in ("encoder", "pose_encoder"):
            net = m.encoder
            _set(net.conv1, "stem" in stages)
            for name in ("layer1", "layer2", "layer3", "layer4"):
                _set(getattr(net, name), name in stages)
        elif key == "decoder":
            _set(m, "decoder" in stages)
        else:
            _set(m, "pose" in stages)
    return stages


def is_channels_last(t):
    """4-D map whose memory is [B][H][W][C] and NOT also [B][C][H][W] (C == 1 or H == W == 1 maps are both: planar)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=_CL)


def weight_layout(module):
    """Is `module` a channels-last stage?  What apply_plan recorded on it, else the layout of its first convolution weight with a
    kernel larger than 1x1 -- a [O, I, 1, 1] weight is planar and channels-last at once (ResNet-50's blocks START with one: read
    as "planar" it sent every layer's input through a transposing copy, 45 strided copies = 4.6 ms per step at 320x1024)."""
    flag = getattr(module, "_mdx_channels_last", None)
    if flag is not None:
        return bool(flag)
    for p in module.parameters():
        if p.dim() == 4 and (p.shape[2] > 1 or p.shape[3] > 1):
            return is_channels_last(p)
    return False


def _in_layout(t, cl):
    return is_channels_last(t) == cl or (t.is_contiguous() and t.is_contiguous(memory_format=_CL))


def to_layout(t, cl):
    """t in the wanted layout (a transposing copy only where the layouts differ; differentiable).  A forked pair
    (BatchNorm2d.act(fork=True)) that needs the copy becomes the copy twice: both consumers read the converted map."""
    if isinstance(t, tuple):
        if all(_in_layout(x, cl) for x in t):
            return t
        c = to_layout(t[0], cl)
        return (c, c)
    if _in_layout(t, cl):
        return t
    return t.contiguous(memory_format=_CL if cl else torch.contiguous_format)
