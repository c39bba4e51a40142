"""GPU tests (MI355X) of the network glue kernels (csrc/glue.hip) against the torch op sequences they replace:
decoder glue = ReflectionPad2d(1)(cat(interpolate(ELU(raw), x2), skip)), max-pool 3x3/2/1, and the DepthDecoder's
glue path against its op-by-op path.  float32: 1e-6 / 1e-5; bfloat16 storage: one bf16 ulp."""
import importlib

import pytest
import torch

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from mdx import functional
    return functional

def _leaf(t, dtype, cl):
    """A leaf on the GPU in the wanted memory layout (cl: channels-last, what csrc/*_nhwc.hip take)."""
    t = t.to("cuda", dtype)
    if cl:
        t = t.contiguous(memory_format=torch.channels_last)
    return t.requires_grad_(True)


def _layout_is(t, cl):
    return t.is_contiguous(memory_format=torch.channels_last) if cl else t.is_contiguous()


GLUE_CASES = [(2, 5, 3, 6, 10, True, True, False), (1, 4, 0, 7, 9, True, False, False), (2, 3, 0, 2, 2, False, False, False),
              (1, 16, 0, 48, 160, True, True, False), (3, 2, 4, 1, 1, True, True, False),
              # channels-last maps: channel counts in 16-byte vectors, odd sizes, one- and two-pixel maps, the fold ring
              (2, 8, 16, 6, 10, True, True, True), (1, 16, 0, 7, 9, True, False, True), (2, 8, 0, 2, 2, False, False, True),
              (1, 16, 0, 48, 160, True, True, True), (3, 8, 8, 1, 1, True, True, True), (2, 32, 64, 5, 3, True, True, True),
              (2, 96, 0, 3, 4, False, False, True), (1, 512, 0, 2, 3, False, False, True), (2, 256, 256, 2, 2, True, True, True)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", GLUE_CASES)
def test_decoder_glue_matches_torch_ops(F, cfg, dtype):
    """decoder_glue == ReflectionPad2d(1)(cat(interpolate(ELU(raw), x2 nearest), skip)), forward and backward -- planar
    maps (csrc/glue.hip) and channels-last maps (csrc/glue_nhwc.hip)."""
    B, C1, C2, h, w, elu, up, cl = cfg
    g = torch.Generator().manual_seed(11)
    raw = _leaf(torch.randn(B, C1, h, w, generator=g), dtype, cl)
    u = 2 if up else 1
    skip = _leaf(torch.randn(B, C2, u * h, u * w, generator=g), dtype, cl) if C2 else None
    bias = torch.randn(C1, generator=g).cuda().requires_grad_(True) if elu else None    # conv bias folded in
    out = F.decoder_glue(raw, skip, elu=elu, upsample=up, bias=bias)
    if cl and F.is_channels_last(raw):       # (a one-pixel map is planar and channels-last at once: it takes the planar kernels)
        assert _layout_is(out, True), "a channels-last input gives a channels-last result"
    raw2 = raw.detach().clone().requires_grad_(True)
    skip2 = skip.detach().clone().requires_grad_(True) if C2 else None
    bias2 = bias.detach().clone().requires_grad_(True) if bias is not None else None
    x = raw2 if bias is None else (raw2.float() + bias2.view(1, -1, 1, 1)).to(dtype)
    x = torch.nn.functional.elu(x) if elu else x
    if up:
        x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    if C2:
        x = torch.cat((x, skip2), 1)
    ref = torch.nn.ReflectionPad2d(1)(x)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert out.shape == ref.shape and out.dtype == ref.dtype
    torch.testing.assert_close(out.float(), ref.float(), rtol=tol, atol=tol)
    gout = torch.randn(ref.shape, generator=g).to("cuda", dtype)
    out.backward(gout)
    ref.backward(gout)
    gtol = 1e-5 if dtype == torch.float32 else 6e-2   # bf16: the reference rounds after every op, the kernel once
    torch.testing.assert_close(raw.grad.float(), raw2.grad.float(), rtol=gtol, atol=gtol)
    assert not (cl and F.is_channels_last(raw)) or _layout_is(raw.grad, True)
    if C2:
        torch.testing.assert_close(skip.grad.float(), skip2.grad.float(), rtol=gtol, atol=gtol)
    if bias is not None:
        scale = float(bias2.grad.abs().max()) + 1e-12
        assert float((bias.grad - bias2.grad).abs().max()) <= (1e-4 if dtype == torch.float32 else 3e-2) * scale


@pytest.mark.parametrize("cl", [False, True])
def test_decoder_glue_bf16_to_f32_head_input(F, cl):
    raw = _leaf(torch.randn(2, 16, 6, 9), torch.bfloat16, cl)
    out = F.decoder_glue(raw, None, elu=True, upsample=False, out_dtype=torch.float32)
    ref = torch.nn.ReflectionPad2d(1)(torch.nn.functional.elu(raw.detach().float()))
    assert out.dtype == torch.float32 and _layout_is(out, cl)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    gout = torch.randn(ref.shape, device="cuda")
    out.backward(gout)
    raw2 = raw.detach().float().requires_grad_(True)
    torch.nn.ReflectionPad2d(1)(torch.nn.functional.elu(raw2)).backward(gout)
    torch.testing.assert_close(raw.grad.float(), raw2.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("shape", [(2, 3, 8, 10), (1, 2, 7, 9), (2, 64, 96, 320), (1, 1, 1, 1), (1, 2, 2, 3), (2, 8, 7, 9), (1, 16, 2, 3),
                                   (2, 24, 5, 4), (1, 8, 1, 1)])
def test_maxpool3s2_matches_torch(F, shape, dtype, cl):
    """planar (csrc/glue.hip) and channels-last (csrc/glue_nhwc.hip; channel counts of 16-byte vectors -- other shapes fall
    back to the planar kernels whatever the layout) forms against ATen, ties and odd sizes included."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    x[..., ::3] = x[..., :1].clone()                      # ties: the first maximum of the window must win, as in ATen
    x = _leaf(x, dtype, cl)
    y = F.maxpool3s2(x)
    x2 = x.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.max_pool2d(x2, 3, 2, 1)
    assert y.shape == ref.shape
    assert torch.equal(y, ref)
    gout = torch.randn(ref.shape, generator=g).to("cuda", dtype)
    y.backward(gout)
    ref.backward(gout)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(x.grad.float(), x2.grad.float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
def test_forked_outputs_sum_their_gradients(F, dtype, cl):
    """maxpool3s2(fork=True) / bn_act(fork=True) return the result twice; the backward receives the two upstream gradients
    separately and adds them (inside the kernel for channels-last maps) == autograd's sum over the two consumers."""
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(2, 16, 9, 11, generator=g)
    w0, b0 = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g)
    for op in ("pool", "bn"):
        res = {}
        for fork in (True, False):
            x = _leaf(x0, dtype, cl)
            w, b = w0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
            rm, rv = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
            if op == "pool":
                out = F.maxpool3s2(x, fork=fork)
            else:
                out = F.bn_act(x, w, b, rm, rv, 1e-5, 0.1, relu=True, fork=fork)
            ya, yb = out if fork else (out, out)
            if fork:
                assert ya.data_ptr() == yb.data_ptr() and torch.equal(ya, yb)
            g.manual_seed(10)
            ga, gb = torch.randn(ya.shape, generator=g).to("cuda", dtype), torch.randn(ya.shape, generator=g).to("cuda", dtype)
            ((ya.float() * ga.float()).sum() + (yb.float() * gb.float()).sum()).backward()
            res[fork] = (x.grad.float(), w.grad, b.grad)
            # only the second output used: the first gradient is None
            x1 = _leaf(x0, dtype, cl)
            out1 = F.maxpool3s2(x1, fork=True) if op == "pool" else F.bn_act(x1, w0.cuda(), b0.cuda(), None, None, 1e-5, 0.1, fork=True)
            out1[1].float().square().sum().backward()
            assert torch.isfinite(x1.grad).all()
        tol = 1e-5 if dtype == torch.float32 else 3e-2
        torch.testing.assert_close(res[True][0], res[False][0], rtol=tol, atol=tol, msg=op)
        if op == "bn":
            torch.testing.assert_close(res[True][1], res[False][1], rtol=10 * tol, atol=10 * tol)
            torch.testing.assert_close(res[True][2], res[False][2], rtol=10 * tol, atol=10 * tol)


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("amp", [False, True])
def test_decoder_glue_path_equals_module_path(amp, cl):
    """DepthDecoder on the GPU (glue path) == the same module run op by op (the CPU code path, forced); cl: the whole
    network in channels-last memory (csrc/glue_nhwc.hip between the convolutions)."""
    from model_layer import ResnetEncoder, DepthDecoder
    torch.manual_seed(3)
    enc = ResnetEncoder(18, False).cuda()
    dec = DepthDecoder(enc.num_ch_enc).cuda()
    if cl:
        enc, dec = enc.to(memory_format=torch.channels_last), dec.to(memory_format=torch.channels_last)
    img = torch.rand(2, 3, 64, 96, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        feats = [f.detach().requires_grad_(True) for f in enc(img)]
        out = dec(feats)
        loss = sum(v.float().mean() for v in out.values())
    loss.backward()
    grads = {n: p.grad.clone() for n, p in dec.named_parameters()}
    fg = [f.grad.clone() for f in feats]
    dec.zero_grad()
    feats2 = [f.detach().clone().requires_grad_(True) for f in feats]
    dec._glue_ok = lambda: False
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out2 = dec(feats2)
        loss2 = sum(v.float().mean() for v in out2.values())
    loss2.backward()
    tol = 2e-5 if not amp else 3e-2
    for k in out:
        torch.testing.assert_close(out[k], out2[k], rtol=tol, atol=tol)
    for n, p in dec.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((p.grad - grads[n]).abs().max()) <= (5e-4 if not amp else 8e-2) * scale, n
    for a, b in zip(fg, feats2):
        scale = float(b.grad.abs().max()) + 1e-12
        assert float((a.float() - b.grad.float()).abs().max()) <= (5e-4 if not amp else 8e-2) * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("cfg", [(12, 64, 48, 160, True, True), (3, 7, 5, 9, False, True), (2, 16, 96, 320, False, True),
                                 (4, 32, 6, 20, True, False), (2, 8, 3, 3, False, False), (1, 5, 130, 67, True, True),
                                 (1, 8, 130, 67, True, True), (3, 24, 5, 9, True, True), (2, 512, 3, 5, False, True),
                                 (1, 2048, 2, 3, True, True), (5, 40, 1, 1, False, False),
                                 # channels-last one-launch form (>= 256 channels, rows that fit the forward's registers): full / partial
                                 # last sweep, a single row
                                 (12, 256, 12, 40, True, True), (3, 256, 7, 9, False, True), (2, 512, 2, 3, True, False)])
def test_bn_act_matches_torch(F, cfg, dtype, cl):
    """bn_act == relu(batch_norm(x, training=True) + residual): output, running statistics, all four gradients,
    against torch's CPU batch norm in float64 (MIOpen's GPU batch norm drops elements for H*W % 4 != 0 planes --
    db off by O(1) on the (130, 67) case below -- so it cannot be the oracle here)."""
    B, Cc, H, W, has_res, relu = cfg
    g = torch.Generator().manual_seed(21)
    x0 = (torch.randn(B, Cc, H, W, generator=g) * 1.7 + 0.4).to(dtype)
    res0 = torch.randn(B, Cc, H, W, generator=g).to(dtype) if has_res else None
    w0, b0 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm0, rv0 = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    gy0 = torch.randn(B, Cc, H, W, generator=g).to(dtype)

    def leaf(t, dev, dt):
        t = t.to(dev, dt)
        if cl and dev == "cuda" and t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        return t.requires_grad_(True)
    # float64 reference on the CPU (from the same, already rounded, inputs)
    x2, w2, b2 = leaf(x0, "cpu", torch.float64), leaf(w0, "cpu", torch.float64), leaf(b0, "cpu", torch.float64)
    res2 = leaf(res0, "cpu", torch.float64) if has_res else None
    rm2, rv2 = rm0.double(), rv0.double()
    pre = torch.nn.functional.batch_norm(x2, rm2, rv2, w2, b2, True, 0.1, 1e-5)
    if has_res:
        pre = pre + res2
    ref = torch.relu(pre) if relu else pre
    # a pre-activation within rounding of zero may fall on either side of the ReLU: give those elements no gradient
    f32 = dtype == torch.float32
    sure = (pre.detach().abs() > (1e-4 if f32 else 5e-2)).to(dtype) if relu else torch.ones_like(gy0)
    gy0 = gy0 * sure
    ref.backward(gy0.double())

    x, w, b = leaf(x0, "cuda", dtype), leaf(w0, "cuda", torch.float32), leaf(b0, "cuda", torch.float32)
    res = leaf(res0, "cuda", dtype) if has_res else None
    rm, rv = rm0.cuda(), rv0.cuda()
    y = F.bn_act(x, w, b, rm, rv, 1e-5, 0.1, residual=res, relu=relu)
    # channel counts that are not whole 16-byte vectors take the planar kernels whatever the layout (and come back planar)
    nhwc = cl and F.is_channels_last(x) and Cc % (4 if f32 else 8) == 0
    assert y.dtype == dtype and y.shape == ref.shape and _layout_is(y, nhwc)
    y.backward(gy0.cuda())
    assert not nhwc or _layout_is(x.grad, True)

    def close(a, bb, name, tol):
        scale = float(bb.abs().max()) + 1e-12
        err = float((a.detach().double().cpu() - bb.detach()).abs().max()) / scale
        assert err <= tol, "%s: %g" % (name, err)
    close(y, ref, "y", 1e-5 if f32 else 1e-2)
    close(rm, rm2, "running_mean", 1e-5)
    close(rv, rv2, "running_var", 1e-5)
    close(x.grad, x2.grad, "dx", 1e-4 if f32 else 3e-2)
    close(w.grad, w2.grad, "dgamma", 1e-4 if f32 else 3e-2)
    close(b.grad, b2.grad, "dbeta", 1e-4 if f32 else 3e-2)
    if has_res:
        close(res.grad, res2.grad, "dres", 1e-5 if f32 else 1e-2)


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("layers", [18, 50])
def test_encoder_fused_norm_path_equals_module_path(layers, cl, monkeypatch):
    """ResnetEncoder in training mode on the GPU (fused norm + max-pool kernels) == the same modules op by op."""
    from model_layer import ResnetEncoder
    from model_layer.depth_encoder import BatchNorm2d
    torch.manual_seed(5)
    monkeypatch.setattr(BatchNorm2d, "fused_min_elements", 0)      # every layer through the fused kernels
    enc = ResnetEncoder(layers, False).cuda().train()
    ref = ResnetEncoder(layers, False).cuda().train()
    ref.load_state_dict(enc.state_dict())
    if cl:
        enc = enc.to(memory_format=torch.channels_last)
    # 128x256: every map's H*W is a multiple of 4 (MIOpen's batch norm, the reference here, is off for other planes)
    img = torch.rand(4, 3, 128, 256, device="cuda")
    # no map changes its layout inside a network whose stages share one (ResNet-50's layers begin with a 1x1 convolution, whose
    # weight is planar and channels-last at once: read as planar it sent every layer's input through a transposing copy)
    import mdx.layout as L
    made = []
    real = torch.Tensor.contiguous
    monkeypatch.setattr(torch.Tensor, "contiguous", lambda t, *a, **k: (made.append(tuple(t.shape)) if t.dim() == 4 and t.shape[1] > 3 and not (
        t.is_contiguous(memory_format=k.get("memory_format", torch.contiguous_format))) else None, real(t, *a, **k))[1])
    feats = enc(img)
    monkeypatch.setattr(torch.Tensor, "contiguous", real)
    assert not made, "layout-changing copies of feature maps: %s" % made
    assert all(L.weight_layout(getattr(enc.encoder, "layer%d" % i)) == cl for i in (1, 2, 3, 4))
    if cl:
        assert all(f.is_contiguous(memory_format=torch.channels_last) for f in feats)
    loss = sum(f.mean() for f in feats)
    loss.backward()
    plain = BatchNorm2d.act
    try:
        BatchNorm2d.act = lambda self, x, residual=None, relu=True, fork=False: \
            (lambda o: (o, o) if fork else o)((lambda o: torch.relu(o) if relu else o)(self(x) if residual is None else self(x) + residual))
        feats2 = ref(img)
        sum(f.mean() for f in feats2).backward()
    finally:
        BatchNorm2d.act = plain
    tol = 2e-4 if layers == 18 else 2e-3        # 50 layers of float32 rounding differences add up
    for a, b in zip(feats, feats2):
        torch.testing.assert_close(a, b, rtol=tol, atol=tol)
    sa, sb = enc.state_dict(), ref.state_dict()
    for k in sa:
        torch.testing.assert_close(sa[k].float(), sb[k].float(), rtol=tol, atol=tol, msg=k)
    for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        if p.grad is None:
            assert q.grad is None
            continue
        # weights in front of a batch norm get gradients that are small differences of large terms (and MIOpen's
        # weight-gradient kernels sum with atomics): compare in the Frobenius norm, with an absolute floor
        err, scale = float((p.grad - q.grad).norm()), float(q.grad.norm())
        assert err <= (2e-2 if layers == 18 else 1e-1) * scale + 1e-5, "%s: err %.3g scale %.3g" % (n, err, scale)


def test_batched_pose_pairs_equal_the_loop_on_gpu():
    """forward_pose on the GPU: one batched pass (grouped fused batch norms) == one call per pair."""
    import types
    from model_layer import ResnetEncoder, PoseDecoder
    from model_tool.processor import compute
    opt = types.SimpleNamespace(frame_ids=[0, -1, 1], pose_frames="pair", pose_type="separate", batch=3)
    g = torch.Generator().manual_seed(1)
    inputs = {("color_aug", f, 0): torch.rand(3, 3, 64, 128, generator=g).cuda() for f in (0, -1, 1)}
    res = {}
    for batched in (True, False):
        torch.manual_seed(7)
        enc = ResnetEncoder(18, False, num_input_images=2).cuda().train()
        dec = PoseDecoder(enc.num_ch_enc, 1, 2).cuda().train()
        st = types.SimpleNamespace(model={"pose_encoder": enc, "pose_decoder": dec})
        opt.batch_pose_pairs = batched
        _, out = compute(opt, "cuda").forward_pose(dict(inputs), {}, st)
        sum(out[("c2c", f, 0)].square().sum() for f in (-1, 1)).backward()
        res[batched] = (out, st)
    (oa, sa), (ob, sb) = res[True], res[False]
    for f in (-1, 1):
        torch.testing.assert_close(oa[("c2c", f, 0)], ob[("c2c", f, 0)], rtol=1e-4, atol=1e-5)
    da, db = sa.model["pose_encoder"].state_dict(), sb.model["pose_encoder"].state_dict()
    for k in da:
        torch.testing.assert_close(da[k].float(), db[k].float(), rtol=1e-4, atol=1e-5, msg=k)
    for name in ("pose_encoder", "pose_decoder"):
        for (n, p), (_, q) in zip(sa.model[name].named_parameters(), sb.model[name].named_parameters()):
            if q.grad is None:
                continue
            err, scale = float((p.grad - q.grad).norm()), float(q.grad.norm())
            assert err <= 2e-2 * scale + 1e-6, "%s.%s: %g vs %g" % (name, n, err, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("shape", [(6, 8, 12, 40), (4, 16, 96, 320), (4, 256, 6, 20)])      # single-launch / multi-launch paths
def test_bn_act_groups_equal_separate_calls(F, shape, dtype, cl):
    """bn_act(groups=2) on a batch == bn_act on its two halves one after the other (outputs, running statistics,
    gradients): what lets both frame pairs go through the pose network in one batch."""
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(3)
    fmt = torch.channels_last if cl else torch.contiguous_format
    x0 = torch.randn(B, Cc, H, W, generator=g).to("cuda", dtype).contiguous(memory_format=fmt)
    r0 = torch.randn(B, Cc, H, W, generator=g).to("cuda", dtype).contiguous(memory_format=fmt)
    w0, b0 = (torch.rand(Cc, generator=g) + 0.5).cuda(), torch.randn(Cc, generator=g).cuda()
    gy = torch.randn(B, Cc, H, W, generator=g).to("cuda", dtype)
    out = {}
    for mode in ("grouped", "separate"):
        x, r = x0.clone(memory_format=torch.preserve_format).requires_grad_(True), r0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        if mode == "grouped":
            y = F.bn_act(x, w, b, rm, rv, 1e-5, 0.1, residual=r, relu=True, groups=2)
        else:
            y = torch.cat([F.bn_act(xc, w, b, rm, rv, 1e-5, 0.1, residual=rc, relu=True)
                           for xc, rc in zip(x.chunk(2), r.chunk(2))])
        y.backward(gy)
        out[mode] = (y.detach(), rm, rv, x.grad, r.grad, w.grad, b.grad)
    for a, bb, name in zip(out["grouped"], out["separate"], ("y", "running_mean", "running_var", "dx", "dres", "dgamma", "dbeta")):
        if name in ("dgamma", "dbeta"):      # summed in the kernel vs by autograd: same terms, different order
            torch.testing.assert_close(a, bb, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-3, msg=name)
        else:
            assert torch.equal(a, bb), name


@pytest.mark.parametrize("invert", [False, True])
def test_param2matrix_kernel_matches_torch_ops(F, invert):
    """mdx_param2matrix == the reference's op sequence (vector2translation / angle2rotation / matmul), forward and
    backward, incl. tiny angles (the 1e-5 in the normalisation) and the exact zero rotation."""
    from model_layer.warp import angle2rotation, vector2translation
    g = torch.Generator().manual_seed(2)
    aa0 = torch.randn(9, 1, 3, generator=g) * torch.tensor([1.0, 0.3, 0.01, 1e-3, 1e-5, 2.5, 0.0, 0.7, 3.0]).view(9, 1, 1)
    t0 = torch.randn(9, 1, 3, generator=g)
    gm = torch.randn(9, 4, 4, generator=g).cuda()

    def ref(aa, t):
        R = angle2rotation(aa)
        tt = t.clone()
        if invert:
            R, tt = R.transpose(1, 2), tt * -1
        T = vector2translation(tt)
        return torch.matmul(R, T) if invert else torch.matmul(T, R)
    a1, t1 = aa0.cuda().requires_grad_(True), t0.cuda().requires_grad_(True)
    a2, t2 = aa0.double().requires_grad_(True), t0.double().requires_grad_(True)
    M1, M2 = F.param2matrix(a1, t1, invert), ref(a2, t2)
    torch.testing.assert_close(M1.double().cpu(), M2, rtol=2e-6, atol=2e-6)
    M1.backward(gm)
    M2.backward(gm.double().cpu())
    torch.testing.assert_close(t1.grad.double().cpu(), t2.grad, rtol=1e-5, atol=1e-5)
    ok = aa0.abs().sum((1, 2)) > 0                     # at the exact origin |a| has no gradient: both give a finite value
    torch.testing.assert_close(a1.grad.double().cpu()[ok], a2.grad[ok], rtol=2e-4, atol=2e-4)
    assert torch.isfinite(a1.grad).all()


HEAD_CASES = [(2, 16, 6, 10), (1, 16, 33, 70), (3, 32, 5, 7), (2, 64, 9, 4), (1, 128, 13, 21), (2, 256, 3, 5), (12, 16, 48, 160),
              (2, 16, 1, 1), (1, 32, 2, 65)]


@pytest.mark.parametrize("wcl", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", HEAD_CASES)
def test_disp_head_matches_conv_bias_sigmoid(F, cfg, dtype, wcl):
    """disp_head == sigmoid(conv2d(x, w, b)) with one output channel (depth_decoder.py:73-74,108-110), forward and the three
    gradients, weight in either memory format.  float32 maps: 2e-6 / 1e-5 relative to the largest entry; bfloat16 maps against
    the float32 convolution of the SAME (bf16-rounded) input: the kernel accumulates in float32."""
    B, Cc, h, w = cfg
    if dtype == torch.bfloat16 and Cc % 8:
        pytest.skip("bfloat16 vectors hold 8 channels")
    g = torch.Generator().manual_seed(3)
    x = _leaf(torch.randn(B, Cc, h + 2, w + 2, generator=g), dtype, True)
    wt = (0.2 * torch.randn(1, Cc, 3, 3, generator=g)).cuda()
    if wcl:
        wt = wt.contiguous(memory_format=torch.channels_last)
    wt.requires_grad_(True)
    b = torch.randn(1, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(B, 1, h, w, generator=g).cuda()
    assert F.disp_head_ok(x, wt)
    y = F.disp_head(x, wt, b)
    assert y.dtype == torch.float32 and y.shape == (B, 1, h, w)
    gx, gw, gb = torch.autograd.grad(y, (x, wt, b), gy)
    assert gx.dtype == dtype and _layout_is(gx, True) and gw.shape == wt.shape and gw.stride() == wt.stride()
    xr = x.detach().double().requires_grad_(True)
    wr, br = wt.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yr = torch.sigmoid(torch.nn.functional.conv2d(xr, wr, br))
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), gy.double())

    def close(a, ref, tol, what):
        d = float((a.double() - ref).abs().max())
        assert d <= tol * max(1e-30, float(ref.abs().max())), (what, d, float(ref.abs().max()))
    close(y, yr, 2e-6, "disp")
    close(gx, gxr, 2e-6 if dtype == torch.float32 else 8e-3, "gx")          # gx is stored in x's dtype
    close(gw, gwr, 1e-5, "gw")
    close(gb, gbr, 1e-5, "gb")


def test_disp_head_refuses_what_it_cannot_take(F):
    x = torch.randn(1, 24, 6, 6).cuda().contiguous(memory_format=torch.channels_last)      # 6 vectors: not a power of two
    w = torch.randn(1, 24, 3, 3).cuda()
    assert not F.disp_head_ok(x, w)
    with pytest.raises(Exception):
        F.disp_head(x, w)
    assert not F.disp_head_ok(torch.randn(1, 16, 6, 6).cuda(), torch.randn(1, 16, 3, 3).cuda())          # planar map
    assert not F.disp_head_ok(x[:, :16].contiguous(memory_format=torch.channels_last), torch.randn(2, 16, 3, 3).cuda())


def test_decoder_with_fused_heads_equals_miopen_heads():
    """DepthDecoder with the hand-written heads against the same decoder with MIOpen convolution + bias + sigmoid heads:
    disparities and every parameter gradient."""
    from model_layer.depth_decoder import DepthDecoder
    from mdx.layout import apply_plan
    torch.manual_seed(0)
    enc = [64, 64, 128, 256, 512]
    dec = DepthDecoder(enc).cuda()
    apply_plan({"decoder": dec}, "all")
    feats = [torch.randn(2, c, 32 >> i, 64 >> i).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for i, c in enumerate(enc)]
    res = []
    for fused in (True, False):
        dec.fused_heads = fused
        dec.zero_grad(set_to_none=True)
        out = dec(feats)
        loss = sum((out[("disp", s)] * (s + 1.0)).square().mean() for s in range(4))
        grads = torch.autograd.grad(loss, list(dec.parameters()) + feats)
        res.append(([out[("disp", s)].detach() for s in range(4)], grads))
    dec.fused_heads = True
    (da, ga), (db, gb) = res
    for s in range(4):
        assert float((da[s] - db[s]).abs().max()) <= 2e-6, s
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 2e-5 * max(1e-30, float(b.abs().max())) + 1e-9


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(24, 256, 6, 20), (2, 8, 3, 5), (3, 64, 1, 2), (1, 512, 7, 2), (5, 16, 33, 9)])
def test_bias_act_matches_torch(F, shape, dtype, relu):
    """bias_act == relu(x + b) (what follows a pose-decoder convolution that ran without its bias), forward, dx and db."""
    g = torch.Generator().manual_seed(9)
    x = _leaf(torch.randn(*shape, generator=g), dtype, True)
    b = torch.randn(shape[1], generator=g).cuda().requires_grad_(True)
    gy = torch.randn(*shape, generator=g).to("cuda", dtype).contiguous(memory_format=torch.channels_last)
    y = F.bias_act(x, b, relu=relu)
    gx, gb = torch.autograd.grad(y, (x, b), gy)
    assert _layout_is(y, True) and _layout_is(gx, True) and y.dtype == dtype and gb.dtype == torch.float32
    xr, br = x.detach().float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = xr + br.view(1, -1, 1, 1)
    yr = torch.relu(yr) if relu else yr
    if dtype == torch.bfloat16:
        yr = yr.detach().to(dtype).float() + (yr - yr.detach())  # the kernel rounds y once, like this
    gxr, gbr = torch.autograd.grad(yr, (xr, br), gy.float())
    assert torch.equal(y.float(), yr.detach()) if dtype == torch.float32 else float((y.float() - yr.detach()).abs().max()) <= 1e-2
    assert torch.equal(gx.float(), gxr)
    assert float((gb - gbr).abs().max()) <= 1e-5 * max(1.0, float(gbr.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(24, 12, 6, 20), (2, 6, 1, 2), (3, 12, 5, 7), (4, 64, 3, 3)])
def test_mean_bias_matches_torch(F, shape, dtype):
    """mean_bias == 0.01 * (x + b).mean((2, 3)) (pose_decoder.py:51-53), forward, dx and db."""
    g = torch.Generator().manual_seed(10)
    x = _leaf(torch.randn(*shape, generator=g), dtype, True)
    b = torch.randn(shape[1], generator=g).cuda().requires_grad_(True)
    go = torch.randn(shape[0], shape[1], generator=g).cuda()
    out = F.mean_bias(x, b, scale=0.01)
    gx, gb = torch.autograd.grad(out, (x, b), go)
    xr, br = x.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    outr = 0.01 * (xr + br.view(1, -1, 1, 1)).mean(dim=(2, 3))
    gxr, gbr = torch.autograd.grad(outr, (xr, br), go.double())
    assert out.dtype == torch.float32 and _layout_is(gx, True) and gx.dtype == dtype
    assert float((out.double() - outr).abs().max()) <= 2e-6 * float(outr.abs().max())
    assert float((gx.double() - gxr).abs().max()) <= (2e-6 if dtype == torch.float32 else 8e-3) * float(gxr.abs().max())
    assert float((gb.double() - gbr).abs().max()) <= 2e-6 * float(gbr.abs().max())


def test_pose_decoder_fused_tail_equals_torch_ops(monkeypatch):
    """PoseDecoder with bias_act / mean_bias behind bias-free convolutions against the module's plain form.  Gradients are
    compared on an input for which every ReLU takes the same branch in both forms: a pre-activation within the 2e-7 rounding
    difference of the kink takes the other one, and its whole gradient with it (about one input in five has such an element)."""
    from model_layer.pose_decoder import PoseDecoder
    from mdx.layout import apply_plan
    torch.manual_seed(0)
    dec = PoseDecoder([64, 64, 128, 256, 512], 1, 2).cuda()
    apply_plan({"pose_decoder": dec}, "all")
    acts = []
    plain = PoseDecoder._conv

    def spy(self, k, x, relu=True):
        y = plain(self, k, x, relu)
        acts.append(y.detach() > 0)
        return y
    monkeypatch.setattr(PoseDecoder, "_conv", spy)
    for seed in range(20):
        g = torch.Generator().manual_seed(seed)
        feat = torch.randn(6, 512, 6, 20, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res, masks = [], []
        for fused in (True, False):
            dec.fused_tail = fused
            del acts[:]
            a, t = dec([[feat]])
            masks.append(list(acts))
            grads = torch.autograd.grad((a * a).sum() + (t * 3.0).sum(), list(dec.parameters()) + [feat])
            res.append((a.detach(), t.detach(), grads))
            if fused:
                assert a._base is not None and a._base is t._base      # what processor._pose_head_output looks for
        (a1, t1, g1), (a2, t2, g2) = res
        assert float((a1 - a2).abs().max()) <= 2e-6 * float(a2.abs().max()) and float((t1 - t2).abs().max()) <= 2e-6 * float(t2.abs().max())
        if all(torch.equal(u, v) for u, v in zip(*masks)):
            break
    else:
        pytest.fail("twenty inputs in a row with a ReLU at its kink")
    dec.fused_tail = True
    for x, y in zip(g1, g2):
        assert float((x - y).abs().max()) <= 2e-5 * max(1e-30, float(y.abs().max())) + 1e-10


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("blocks,groups", [(1, 1), (2, 2), (1, 2), (2, 1)])
def test_encoder_input_equals_cat_normalise(F, blocks, groups, dtype):
    """encoder_input == ((cat of the frame pairs) - 0.45) / 0.225 laid out channels-last (depth_encoder.py:89 behind
    processor.py:61-75), bit for bit in float32 (ATen's GPU division by a scalar is a multiplication by its reciprocal)."""
    g = torch.Generator().manual_seed(2)
    frames = [[torch.rand(3, 3, 10, 14, generator=g).cuda() for _ in range(groups)] for _ in range(blocks)]
    stack = F.FrameStack(frames)
    assert stack.ok()
    out = F.encoder_input(stack, 0.45, 0.225, dtype)
    ref = ((stack.tensor() - 0.45) / 0.225).to(dtype)
    assert out.shape == ref.shape and _layout_is(out, True) and out.dtype == dtype
    assert torch.equal(out, ref)
    assert not F.FrameStack([[frames[0][0].requires_grad_(True)]]).ok()


@pytest.mark.parametrize("wcl", [False, True])
@pytest.mark.parametrize("cfg", [(2, 16, 8, 12), (1, 32, 5, 8), (3, 16, 7, 4), (2, 32, 33, 20), (12, 16, 48, 160)])
def test_thin_conv3x3_weight_gradient_matches_float64(F, cfg, wcl):
    """thin_conv3x3: the MFMA weight gradient of the decoder's 16-output-channel 3x3 convolutions (csrc/thinconv_nhwc.hip) against
    the float64 convolution's; forward and data gradient are MIOpen's and are checked against it too."""
    B, Cin, h, w = cfg
    g = torch.Generator().manual_seed(4)
    x = _leaf(torch.randn(B, Cin, h + 2, w + 2, generator=g), torch.float32, True)
    wt = (0.1 * torch.randn(16, Cin, 3, 3, generator=g)).cuda()
    if wcl:
        wt = wt.contiguous(memory_format=torch.channels_last)
    wt.requires_grad_(True)
    gy = torch.randn(B, 16, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    assert F.thin_conv_ok(x, wt)
    y = F.thin_conv3x3(x, wt)
    gx, gw = torch.autograd.grad(y, (x, wt), gy)
    assert gw.shape == wt.shape and gw.stride() == wt.stride()
    xr, wr = x.detach().double().cpu().requires_grad_(True), wt.detach().double().cpu().contiguous().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double().cpu())
    for a, r, tol in ((y, yr, 2e-6), (gx, gxr, 2e-6), (gw, gwr, 2e-6)):
        assert float((a.double().cpu() - r).abs().max()) <= tol * float(r.abs().max())
    assert not F.thin_conv_ok(x, torch.randn(8, Cin, 3, 3).cuda())                       # other output widths: MIOpen
    assert not F.thin_conv_ok(torch.randn(1, 16, 6, 9).cuda().contiguous(memory_format=torch.channels_last), wt[:, :16])  # w % 4
