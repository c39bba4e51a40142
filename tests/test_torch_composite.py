"""CPU: the plain-PyTorch restatement of the loss path that bench.py times as its second CPU baseline line agrees with
the reference-made goldens (loss, arg-min indices, gradients) -- so the number it produces is the cost of the right work."""
import numpy as np
import pytest
import torch

import goldens
import torch_composite as tc


@pytest.mark.parametrize("name", ["mono_24x40_b2", "stereo_16x32_b1", "noautomask_16x32_b2", "multi_64x160_b2"])
def test_torch_composite_vs_golden(name):
    c = goldens.Case(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))     # noqa: E731
    disps = {s: t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)}
    colors = {s: t(c.color(0, s)) for s in range(c.n_scales)}
    Ts = [t(c.T(f)).requires_grad_(f != "s") for f in c.sources_ids]
    noises = [t(c["noise_s%d" % s]) for s in range(c.n_scales)] if c.automask else None
    loss, idxs = tc.loss_path(disps, colors, [t(c.color(f)) for f in c.sources_ids], t(c["K"]), t(c["inv_K"]), Ts,
                              noises, automask=c.automask)
    loss.backward()
    assert abs(float(loss) - float(c["loss"])) <= 1e-5 * abs(float(c["loss"]))
    for s in range(c.n_scales):
        if "idx_s%d" % s in c:
            assert (idxs[s].numpy() == c["idx_s%d" % s]).mean() > 0.999       # op order differs from the reference's in places
        g, ref = disps[s].grad.numpy(), c["grad_disp_s%d" % s]
        assert np.abs(g - ref).max() <= 2e-4 * np.abs(ref).max()
