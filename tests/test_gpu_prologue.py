"""GPU: the per-step prologue (csrc/photo_prologue.hip, mdx_photometric_prologue) and the training kernel fed by it
(mdx_photometric_train_pre).

  * injected noise (what the parity tests use): target statistics, identity maps and the best identity channel per scale
    are bit-identical to the per-op kernels / the torch formula, and the training kernel fed with them returns what the
    ident + noise form returns -- indices, to_optimise, sums, depth bit for bit, gradients too;
  * drawn noise (what a training step uses): N(0,1) -- moments, independence over pixels / frames / scales / steps --
    reproducible for a given {seed, offset}, advanced on the device by the step itself.
The reference draws torch.randn on the HOST (processor.py:195); no device generator can reproduce that stream, the
distribution is what the auto-mask depends on.
"""
import numpy as np
import pytest
import torch

from test_gpu_train import _synth_images

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


def _case(G, B, H, W, S, nscales, seed=3):
    colors, K, invK, Ts, rng = _synth_images(B, H, W, S, seed=seed)
    hw = [(H >> s, W >> s) if H % 8 == 0 and W % 8 == 0 else (H, W) for s in range(nscales)]
    disps = [rng.rand(B, 1, h, w).astype(np.float32) for h, w in hw]
    noises = [G.t(rng.randn(B, S, H, W).astype(np.float32)) for _ in range(nscales)]
    Kt = G.t(K)
    P = torch.stack([G.F.compose_projection(Kt, G.t(T)) for T in Ts])
    return G.t(colors[0]), [G.t(x) for x in colors[1:]], disps, noises, P, G.t(invK)


@pytest.mark.parametrize("B,H,W,S,nscales", [(2, 192, 640, 2, 4), (1, 192, 640, 3, 4), (1, 100, 150, 2, 1),
                                             (3, 64, 68, 1, 4), (1, 192, 640, 4, 2)])
def test_prologue_injected_noise_matches_per_op_results(G, B, H, W, S, nscales):
    tgt, srcs, disps, noises, P, invK = _case(G, B, H, W, S, nscales)
    pre = G.F.photometric_prologue(tgt, srcs, nscales, noises=noises, need_ident=True)
    ident = G.F.identity_loss(tgt, srcs)
    assert torch.equal(pre["ident"], ident)
    for s in range(nscales):
        v = ident + noises[s] * 1e-5               # processor.py:195: identity_loss + 0.00001 * randn (mul, then add)
        val, idx = torch.min(v, dim=1)                # first minimum
        assert torch.equal(pre["bidfi"][s][..., 0], val), "best identity value, scale %d" % s
        assert torch.equal(pre["bidfi"][s][..., 1].contiguous().view(torch.int32), idx.int()), "best identity channel, scale %d" % s
    # target statistics against the unfused SSIM pieces: mu_y = avgpool3(reflect-pad(y)), sigma_y = avgpool3(y*y) - mu_y^2
    # (torch's own kernels sum in another order: 1e-6; the bit-exact check is the training kernel's output below)
    pad = torch.nn.functional.pad(tgt, (1, 1, 1, 1), mode="reflect")
    mu = torch.nn.functional.avg_pool2d(pad, 3, 1)
    sg = torch.nn.functional.avg_pool2d(pad * pad, 3, 1) - mu * mu
    G.assert_close(pre["tstat"][..., :3].permute(0, 3, 1, 2), mu.cpu().numpy(), "mu_y", rel=1e-6)
    assert float((pre["tstat"][..., 3:].permute(0, 3, 1, 2) - sg).abs().max()) < 1e-6
    # the training kernel fed by the prologue == the ident + noise form
    kw = dict(automask=True, need_depth=True, need_to_opt=True)
    outs = []
    for use_pre in (False, True):
        d = [G.t(x).requires_grad_(True) for x in disps]
        Pt = P.clone().requires_grad_(True)
        o = G.F.photometric_train(d, Pt, tgt, srcs, invK, None if use_pre else ident, None if use_pre else noises,
                                  pre=pre if use_pre else None, **kw)
        (o["sums"].sum() / (B * H * W)).backward()
        outs.append((o, d, Pt))
    (a, da, Pa), (b, db, Pb) = outs
    assert torch.equal(a["sums"], b["sums"]) and torch.equal(a["depth"], b["depth"])
    for s in range(nscales):
        assert torch.equal(a["idx"][s], b["idx"][s]) and torch.equal(a["to_opt"][s], b["to_opt"][s])
        assert torch.equal(da[s].grad, db[s].grad), "d disp, scale %d" % s
    assert torch.equal(Pa.grad, Pb.grad)
    with torch.no_grad():     # forward-only form fed by the prologue
        c = G.F.photometric_train([G.t(x) for x in disps], P, tgt, srcs, invK, pre=pre, **kw)
    assert torch.equal(c["sums"], a["sums"].detach()) and all(torch.equal(c["idx"][s], a["idx"][s]) for s in range(nscales))


def test_prologue_without_automask(G):
    tgt, srcs, disps, noises, P, invK = _case(G, 1, 192, 640, 2, 2)
    pre = G.F.photometric_prologue(tgt, srcs, 2, automask=False)
    assert pre["bidfi"] is None and pre["ident"] is None
    kw = dict(automask=False, need_to_opt=True)
    a = G.F.photometric_train([G.t(x) for x in disps], P, tgt, srcs, invK, **kw)
    b = G.F.photometric_train([G.t(x) for x in disps], P, tgt, srcs, invK, pre=pre, **kw)
    assert torch.equal(a["sums"], b["sums"]) and all(torch.equal(a["idx"][s], b["idx"][s]) for s in range(2))


def _drawn(G, S, nscales, state, B=2, H=192, W=640, advance=True):
    """normals recovered through a target that equals its sources: the identity losses are then exactly 0 (n == d in the
    SSIM quotient, |y - x| = 0) and bid_s = min_f(1e-5 * n_f) exactly."""
    g = torch.Generator().manual_seed(5)
    img = torch.rand(B, 3, H, W, generator=g).cuda()
    pre = G.F.photometric_prologue(img, [img] * S, nscales, rng=state, need_ident=True, advance=advance)
    assert float(pre["ident"].abs().max()) == 0.0
    bid = torch.stack([x[..., 0] for x in pre["bidfi"]]).double() / float(np.float32(1e-5))
    fi = torch.stack([x[..., 1].contiguous().view(torch.int32) for x in pre["bidfi"]])
    return bid, fi


def test_drawn_noise_is_standard_normal_and_independent(G):
    st = G.F.noise_state("cuda:0", seed=1234)
    n, fi = _drawn(G, 1, 4, st)                       # S = 1: the four scales' draws themselves, [4, B, H, W]
    N = n[0].numel()
    assert int(fi.abs().max()) == 0
    for s in range(4):
        x = n[s].flatten()
        assert abs(float(x.mean())) < 5 / np.sqrt(N), ("mean", s, float(x.mean()))
        assert abs(float(x.var()) - 1) < 5 * np.sqrt(2.0 / N), ("variance", s, float(x.var()))
        assert abs(float((x ** 3).mean())) < 5 * np.sqrt(15.0 / N), ("skewness", s)
        assert abs(float((x ** 4).mean()) - 3) < 5 * np.sqrt(96.0 / N), ("kurtosis", s)
        assert 4.0 < float(x.abs().max()) < 6.5                                        # tails reach, nothing absurd
        # independence: neighbouring pixels (x and y), the other scales
        assert abs(float((n[s][:, :, 1:] * n[s][:, :, :-1]).mean())) < 5 / np.sqrt(N)
        assert abs(float((n[s][:, 1:] * n[s][:, :-1]).mean())) < 5 / np.sqrt(N)
        for s2 in range(s + 1, 4):
            assert abs(float((n[s] * n[s2]).mean())) < 5 / np.sqrt(N), ("scales correlate", s, s2)
    # the next step (the offset was advanced on the device) is another, uncorrelated draw; the same {seed, offset} the same
    assert st.tensor.tolist() == [1234, 1]
    n2, _ = _drawn(G, 1, 4, st)
    assert abs(float((n * n2).mean())) < 5 / np.sqrt(4 * N) and not torch.equal(n, n2)
    again, _ = _drawn(G, 1, 4, G.F.noise_state("cuda:0", seed=1234))
    assert torch.equal(again, n)
    other, _ = _drawn(G, 1, 4, G.F.noise_state("cuda:0", seed=1235))
    assert abs(float((other * n).mean())) < 5 / np.sqrt(4 * N)
    rank1, _ = _drawn(G, 1, 4, G.F.noise_state("cuda:0", seed=1234, stream=1))      # another data-parallel rank
    assert abs(float((rank1 * n).mean())) < 5 / np.sqrt(4 * N)


@pytest.mark.parametrize("S,emin", [(2, -0.5641895835), (3, -0.8462843753), (4, -1.0293753730)])
def test_drawn_noise_frames_are_independent(G, S, emin):
    """several source frames: the minimum over S independent normals has the known mean, every frame wins equally
    often -- for every scale (S * nscales up to 16 normals per pixel: four Philox calls)."""
    st = G.F.noise_state("cuda:0", seed=77)
    m, fi = _drawn(G, S, 4, st)
    N = m[0].numel()
    for s in range(4):
        assert abs(float(m[s].mean()) - emin) < 5 / np.sqrt(N), (s, float(m[s].mean()))
        share = torch.bincount(fi[s].flatten().long(), minlength=S).double() / N
        assert float((share - 1.0 / S).abs().max()) < 5 * np.sqrt(0.25 / N), share


def test_training_step_advances_the_noise_on_the_device(G):
    tgt, srcs, disps, noises, P, invK = _case(G, 1, 64, 96, 2, 4)
    srcs = [tgt, tgt]       # both identity losses are exactly 0: the noise alone decides between the identity channels
    st = G.F.noise_state("cuda:0", seed=9)
    idx = []
    for step in range(3):
        pre = G.F.photometric_prologue(tgt, srcs, 4, rng=st)
        out = G.F.photometric_train([G.t(x).requires_grad_(True) for x in disps], P, tgt, srcs, invK, pre=pre)
        idx.append(out["idx"][0].clone())
        assert st.tensor.tolist() == [9, step + 1]          # advanced by the step's own finishing kernel
    assert int(idx[0].max()) <= 1 and not torch.equal(idx[0], idx[1])     # identity wins everywhere; other noise, other winner
    with pytest.raises(G.F._lib.MdxError):
        G.F.photometric_prologue(tgt, srcs, 4)               # neither injected noise nor a generator state
