"""TEST INFRASTRUCTURE: the photometric + smoothness loss on CPU tensors through the oracle, as autograd
functions -- lets the host-side training plumbing (networks, pose driver, DDP) run without a GPU."""
import numpy as np
import torch

from oracle import oracle as orc


class OracleScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, P, target, sources, invK, noise, automask):
        tgt, srcs = target.numpy(), [s.numpy() for s in sources]
        out = orc.photometric_fwd(disp.numpy(), tgt, srcs, invK.numpy(), P.numpy(), noise, automask=automask)
        ctx.save_for_backward(disp, P)
        ctx.stuff = (tgt, srcs, invK.numpy(), out["idx"], automask)
        n = tgt.shape[0] * tgt.shape[2] * tgt.shape[3]
        ctx.n = n
        return torch.tensor(out["sum"] / n, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        disp, P = ctx.saved_tensors
        tgt, srcs, invK, idx, automask = ctx.stuff
        gd, gP = orc.photometric_bwd(disp.numpy(), tgt, srcs, invK, P.numpy(), idx, float(g) / ctx.n, automask=automask)
        return torch.from_numpy(gd), torch.from_numpy(gP), None, None, None, None, None


class OracleSmooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, color):
        v, g = orc.smooth_loss(disp.numpy(), color.numpy(), need_grad=True)
        ctx.g = torch.from_numpy(g)
        return torch.tensor(v, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        return ctx.g * g, None


def cpu_compute_loss(opt, inputs, outputs, seed=0):
    """compute.compute_loss (reference processor.py:166-217) on CPU tensors, oracle-backed."""
    rng = np.random.RandomState(seed)
    target = inputs[("color", 0, 0)]
    sources = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
    K = inputs[("K", 0)]
    P = torch.stack([torch.matmul(K, inputs["stereo"] if f == "s" else outputs[("c2c", f, 0)])[:, :3, :]
                     for f in opt.frame_ids[1:]])
    B, _, H, W = target.shape
    total = 0
    for s in opt.scales:
        noise = rng.randn(B, len(sources), H, W).astype(np.float32)
        total = total + OracleScale.apply(outputs[("disp", s)], P, target, sources, inputs[("inv_K", 0)], noise,
                                          bool(opt.use_automasking))
        total = total + opt.disp_smoothness * OracleSmooth.apply(outputs[("disp", s)], inputs[("color", 0, s)]) / (2 ** s)
    return total / len(opt.scales)
