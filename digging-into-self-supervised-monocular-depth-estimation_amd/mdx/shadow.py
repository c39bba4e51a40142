"""bf16 copies of the convolution weights for one training step, made by ONE launch.

Under `torch.autocast(bfloat16)` every convolution casts its float32 weight to bfloat16 on the way in and its weight gradient back to
float32 on the way out: two 5 us launches per convolution and step (124 for the two ResNet-18 networks, 240 for ResNet-50), each in
front of or behind a convolution on its network's chain.  `bf16_weights(modules)` makes all the copies with one multi-tensor launch
before the networks start, lets the modules see them in place of their parameters while the forward runs, and casts all the weight
gradients back with one launch when backward has produced the last of them (one autograd node with every weight as input and
output).  Parameters, optimiser state and checkpoints are untouched: the float32 tensors remain the parameters."""
import contextlib

import torch


class _CastAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *masters):
        outs = [torch.empty_like(m, dtype=torch.bfloat16) for m in masters]          # (preserve_format: channels-last stays)
        torch._foreach_copy_(outs, list(masters))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        have = [(i, g) for i, g in enumerate(grads) if g is not None]
        outs = [torch.empty_like(g, dtype=torch.float32) for _, g in have]
        if outs:
            torch._foreach_copy_(outs, [g for _, g in have])
        res = [None] * len(grads)
        for (i, _), o in zip(have, outs):
            res[i] = o
        return tuple(res)


def _weights(modules):
    """(module, name, parameter) of every float32 convolution weight with more than one output channel that is trained (the
    one-channel disparity heads read their float32 weight themselves: mdx.functional.disp_head)."""
    seen, out = set(), []
    for net in modules:
        for m in net.modules():
            for name, p in m._parameters.items():
                if (p is not None and id(p) not in seen and p.dim() == 4 and p.dtype == torch.float32 and p.requires_grad and p.is_cuda
                        and p.shape[0] > 1):
                    seen.add(id(p))
                    out.append((m, name, p))
    return out


@contextlib.contextmanager
def bf16_weights(modules):
    """While the block runs, `module.weight` of every convolution in `modules` is a bfloat16 copy that autograd links to the
    float32 parameter; the copies and, in backward, the float32 gradients are one launch each."""
    sites = _weights(modules)
    if not sites or not torch.is_grad_enabled():
        yield 0
        return
    shadows = _CastAll.apply(*[p for _, _, p in sites])
    try:
        for (m, name, _), s in zip(sites, shadows):
            m._parameters[name] = s
        yield len(sites)
    finally:
        for m, name, p in sites:
            m._parameters[name] = p
