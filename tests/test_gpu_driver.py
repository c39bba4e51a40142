"""GPU: the step driver `compute` (image2warping + compute_loss, reference processor.py:139-218) end to end
against the reference's golden loss and gradients -- with the one-launch training kernel (all scales, forward +
gradient), in per-scale fused mode (one forward + one backward kernel per scale) and in the op-by-op
mode that runs the reference's sequence on the fine-grained kernels (what a maintainer gets by swapping only
model_layer / model_loss under the reference's own processor.py)."""
import types

import numpy as np
import pytest
import torch

import goldens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


def _setup(G, c, fused, fused_train=True, fused_tail=True):
    from model_tool.processor import compute
    from model_layer import Depth2PointCloud, PointCloud2Pixel
    from model_loss import ReprojectionLoss, SmoothLoss
    opt = types.SimpleNamespace(scales=list(range(c.n_scales)), frame_ids=c.frame_ids, height=c.H, width=c.W,
                                min_depth=0.1, max_depth=100.0, disp_smoothness=1e-3, use_automasking=c.automask,
                                batch=c.B, pose_type="separate", pose_frames="pair", fused=fused, noise="device",
                                fused_train=fused_train, fused_tail=fused_tail)
    st = types.SimpleNamespace(inv_projection={0: Depth2PointCloud(c.B, c.H, c.W)},
                               for_projection={0: PointCloud2Pixel(c.B, c.H, c.W)},
                               loss={"reprojection": ReprojectionLoss(), "edge_aware": SmoothLoss()})
    inputs, outputs = {}, {}
    for f in c.frame_ids:
        inputs[("color", f, 0)] = G.t(c.color(f))
    for s in range(c.n_scales):
        inputs[("color", 0, s)] = G.t(c.color(0, s))
        if c.automask:
            inputs[("noise", s)] = G.t(c["noise_s%d" % s])
        outputs[("disp", s)] = G.t(c["disp_s%d" % s]).requires_grad_(True)
    inputs[("K", 0)], inputs[("inv_K", 0)] = G.t(c["K"]), G.t(c["inv_K"])
    for f in c.sources_ids:
        if f == "s":
            inputs["stereo"] = G.t(c.T(f))
        else:
            outputs[("c2c", f, 0)] = G.t(c.T(f)).requires_grad_(True)
    return compute(opt, G.DEV), st, inputs, outputs


@pytest.mark.parametrize("mode", ["train_kernel", "train_kernel_scalar_ops", "per_scale", "op_by_op"])
@pytest.mark.parametrize("name", goldens.CASES)
def test_compute_driver_vs_golden(G, name, mode):
    """train_kernel: the step's default (one autograd node for the whole loss, mdx.functional.train_loss);
    train_kernel_scalar_ops: the same kernels with the reference's scalar ops after them (opt.fused_tail = False)."""
    c = goldens.Case(name)
    fused = mode != "op_by_op"
    cp, st, inputs, outputs = _setup(G, c, fused, fused_train=mode.startswith("train_kernel"),
                                     fused_tail=(mode == "train_kernel"))
    inputs, outputs = cp.image2warping(inputs, outputs, st)
    outputs = cp.compute_loss(inputs, outputs, st)
    outputs["loss"].backward()
    G.assert_close(outputs["loss"], c["loss"], "loss", rel=1e-5)
    for s in range(c.n_scales):
        G.assert_close(outputs[("disp", s)].grad, c["grad_disp_s%d" % s], "grad disp s%d" % s, elem=1e-4)
        if "idx_s%d" % s in c:
            assert (outputs[("automask", s)].cpu().numpy() == c["idx_s%d" % s]).all(), "auto-mask s%d" % s
    for f in c.sources_ids:
        if f != "s":
            G.assert_close(outputs[("c2c", f, 0)].grad, c["grad_T_%s" % f], "grad T %s" % f)
    G.assert_bitexact(outputs[("depth", 0, 0)], c["depth_s0"], "depth scale 0")
    if not fused and ("warp_%s_s0" % c.sources_ids[0]) in c:
        for f in c.sources_ids:
            G.assert_bitexact(outputs[("warp_color", f, 0)], c["warp_%s_s0" % f], "warp_color %s" % f)


@pytest.mark.parametrize("name", goldens.CASES)
def test_loss_tail_one_node_equals_scalar_ops(G, name):
    """mdx.functional.train_loss (one launch finishes the scalar, one launch fans the gradient out) against the reference's
    scalar ops behind the same kernels: loss and disparity gradients bit for bit, matrix gradients to rounding (the scales'
    projection gradients are summed in index order instead of by ATen's reduction)."""
    c = goldens.Case(name)
    res = []
    for tail in (True, False):
        cp, st, inputs, outputs = _setup(G, c, True, fused_train=True, fused_tail=tail)
        inputs, outputs = cp.image2warping(inputs, outputs, st)
        outputs = cp.compute_loss(inputs, outputs, st)
        (outputs["loss"] * 1.7).backward()                 # an upstream gradient that is not 1
        res.append(outputs)
    a, b = res
    assert torch.equal(a["loss"], b["loss"]), (float(a["loss"]), float(b["loss"]))
    for s in range(c.n_scales):
        assert torch.equal(a[("disp", s)].grad, b[("disp", s)].grad), "grad disp s%d" % s
        assert torch.equal(a[("automask", s)], b[("automask", s)])
    for f in c.sources_ids:
        if f != "s":
            ga, gb = a[("c2c", f, 0)].grad, b[("c2c", f, 0)].grad
            assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max()) + 1e-12, "grad T %s" % f
    assert torch.equal(a[("depth", 0, 0)], b[("depth", 0, 0)])


def test_pose_projection_equals_slices_param2matrix_compose(G):
    """mdx.functional.pose_projection (the pose head's output -> matrices and projections of both source frames, one launch each
    way) against the reference's row slice + [:, 0] + param2matrix + K @ T per frame (processor.py:61-83, :143-160)."""
    from mdx import functional as F
    g = torch.Generator().manual_seed(5)
    n, Fr = 5, 2
    base = (0.05 * torch.randn(2 * n, Fr, 1, 6, generator=g)).to(G.DEV)
    K = torch.eye(4).repeat(n, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.58 * 64, 1.92 * 32, 32.0, 16.0
    K = K.to(G.DEV)
    wP = torch.randn(2, n, 3, 4, generator=g).to(G.DEV)
    wT = torch.randn(2, n, 4, 4, generator=g).to(G.DEV)
    for frame in (0, 1):
        for with_T in (False, True):
            raw1 = base.clone().requires_grad_(True)
            T1, P1 = F.pose_projection(raw1, K, [0, n], [frame, frame], [1, 0])
            ((P1 * wP).sum() + ((T1 * wT).sum() if with_T else 0.0)).backward()
            raw2 = base.clone().requires_grad_(True)
            aa, tr = raw2[..., :3], raw2[..., 3:]
            Ts, Ps = [], []
            for k, inv in enumerate((True, False)):
                a, t = aa[k * n:(k + 1) * n], tr[k * n:(k + 1) * n]
                Ts.append(F.param2matrix(a[:, frame].float(), t[:, frame].float(), invert=inv))
                Ps.append(F.compose_projection(K, Ts[-1]))
            T2, P2 = torch.stack(Ts), torch.stack(Ps)
            ((P2 * wP).sum() + ((T2 * wT).sum() if with_T else 0.0)).backward()
            assert torch.equal(T1, T2) and torch.equal(P1, P2)
            d = float((raw1.grad - raw2.grad).abs().max())
            assert d <= 2e-6 * float(raw2.grad.abs().max()), (frame, with_T, d)
            assert float(raw1.grad[:, 1 - frame].abs().max()) == 0.0        # the entry no source reads
    with pytest.raises(Exception):
        F.pose_projection(base, K, [0, n + 1], [0, 0], [1, 0])              # rows beyond the head's output


def test_training_step_runs_and_learns(G):
    """setting + compute + Adam on the synthetic contract: loss finite and decreasing over a few steps."""
    import importlib
    import sys
    sys.path.insert(0, ".")
    bench = importlib.import_module("bench")
    from model_tool import setting, compute
    torch.manual_seed(0)
    opt = bench.make_opt(2, height=64, width=96)
    st, cp = setting(opt, G.DEV), compute(opt, G.DEV)
    st.set_train()
    inputs = bench.one_batch(st, G.DEV)
    losses = []
    for _ in range(6):
        o = {}
        i, o = cp.forward_depth(inputs, o, st)
        i, o = cp.forward_pose(i, o, st)
        i, o = cp.image2warping(i, o, st)
        o = cp.compute_loss(i, o, st)
        st.optim["optimizer"].zero_grad(set_to_none=True)
        o["loss"].backward()
        st.optim["optimizer"].step()
        losses.append(float(o["loss"].detach()))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    assert o[("automask", 0)].dtype == torch.uint8 and o[("depth", 0, 0)].shape == (2, 1, 64, 96)


def test_training_trajectory_fused_network_kernels_vs_torch_ops(G, monkeypatch):
    """Six optimiser steps with every hand-written network kernel on (decoder glue, max-pool, fused batch norm,
    pose pairs in one batch) against the same six steps with the torch op sequences they replace: same losses."""
    import importlib
    import sys
    sys.path.insert(0, ".")
    bench = importlib.import_module("bench")
    from model_tool import setting, compute
    from model_layer.depth_encoder import BatchNorm2d
    from model_layer.depth_decoder import DepthDecoder
    from mdx import functional as F

    def run(native):
        with monkeypatch.context() as m:
            if not native:
                m.setattr(BatchNorm2d, "fused_min_elements", 1 << 60)          # torch batch_norm + add + relu
                m.setattr(DepthDecoder, "_glue_ok", lambda self: False)         # ELU / interpolate / cat / pad ops
                m.setattr(F, "maxpool3s2", lambda x, fork=False: (lambda y: (y, y) if fork else y)(
                    torch.nn.functional.max_pool2d(x, 3, 2, 1)))
            torch.manual_seed(0)
            opt = bench.make_opt(2, height=64, width=128)
            opt.batch_pose_pairs = native
            opt.noise = "cpu"                      # the same noise stream in both runs
            st, cp = setting(opt, G.DEV), compute(opt, G.DEV)
            st.set_train()
            inputs = bench.one_batch(st, G.DEV)
            torch.manual_seed(1)
            losses = []
            for _ in range(6):
                o = {}
                i, o = cp.forward_depth(inputs, o, st)
                i, o = cp.forward_pose(i, o, st)
                i, o = cp.image2warping(i, o, st)
                o = cp.compute_loss(i, o, st)
                st.optim["optimizer"].zero_grad(set_to_none=True)
                o["loss"].backward()
                st.optim["optimizer"].step()
                losses.append(float(o["loss"].detach()))
            return losses
    a, b = run(True), run(False)
    assert all(np.isfinite(a)) and all(np.isfinite(b))
    np.testing.assert_allclose(a[:3], b[:3], rtol=2e-4, atol=1e-6)     # float32 rounding / atomics-order differences ...
    np.testing.assert_allclose(a, b, rtol=1e-2, atol=1e-5)             # ... which Adam amplifies step by step


def test_kitti_tree_training_and_eigen_style_evaluation(G, tmp_path):
    """N2 + N3 on a synthetic KITTI-raw tree: `setting` with dataset='kitti_mono' feeds the step; the evaluation
    protocol (reference model_test.py:61-119) runs end to end with random weights."""
    import importlib
    import os
    import fake_kitti
    from model_tool import setting, compute
    bench = importlib.import_module("bench")
    names = fake_kitti.make(str(tmp_path), n_frames=6)
    os.makedirs(os.path.join(str(tmp_path), "splits", "fake"))
    for split in ("train", "val", "test"):
        open(os.path.join(str(tmp_path), "splits", "fake", split + "_files.txt"), "w").write("\n".join(names) + "\n")
    opt = bench.make_opt(2, height=192, width=640)
    opt.dataset, opt.datapath, opt.splits, opt.datatype = "kitti_mono", str(tmp_path), os.path.join(str(tmp_path), "splits"), "fake"
    torch.manual_seed(0)
    st, cp = setting(opt, G.DEV), compute(opt, G.DEV)
    st.set_train()
    inputs = next(iter(st.train_dataloader))
    o = {}
    i, o = cp.forward_depth(inputs, o, st)
    i, o = cp.forward_pose(i, o, st)
    i, o = cp.image2warping(i, o, st)
    o = cp.compute_loss(i, o, st)
    o["loss"].backward()
    assert np.isfinite(float(o["loss"].detach()))
    from model_loss import compute_depth_metric
    m = compute_depth_metric(i, o, "torch")
    assert all(np.isfinite(float(v)) for v in m)
    import model_test
    res = model_test.inference(opt, encoder=st.raw_model["encoder"], decoder=st.raw_model["decoder"])
    assert set(res) == set(model_test.METRICS) and all(np.isfinite(v) for v in res.values())
    assert 0 <= res["a1"] <= res["a2"] <= res["a3"] <= 1


def _trainer_losses(graph, automask=False, noise="device", n=6, lr_change_at=4, amp="none", channels_last="auto", batches=None,
                    frame_ids=(0, -1, 1), overlap_pose=True, shadow_weights=True):
    import importlib
    bench = importlib.import_module("bench")
    from model_train import trainer
    torch.manual_seed(0)
    opt = bench.make_opt(2, height=64, width=96, amp=amp, frame_ids=frame_ids)
    opt.use_automasking, opt.graph, opt.synthetic_length, opt.max_steps, opt.miopen_find = automask, graph, 16, 0, False
    opt.noise, opt.channels_last, opt.overlap_pose, opt.shadow_weights = noise, channels_last, overlap_pose, shadow_weights
    tr = trainer(opt)
    tr.setting.set_train()
    if batches is None:
        batches = list(tr.setting.train_dataloader)[:n]
    tr.last_batches = batches
    torch.manual_seed(1)
    losses = []
    for i, b in enumerate(batches):
        if i == lr_change_at:
            for g in tr.setting.optim["optimizer"].param_groups:       # what StepLR does at an epoch boundary
                if torch.is_tensor(g["lr"]):
                    g["lr"].fill_(1e-6)
                else:
                    g["lr"] = 1e-6
        losses.append(float(tr.train_step(dict(b))["loss"].detach()))
    enc = tr.setting.raw_model["encoder"]
    return losses, int(enc.state_dict()["encoder.bn1.num_batches_tracked"]), tr


def _same_trajectory(a, b):
    assert all(np.isfinite(a)) and all(np.isfinite(b)), (a, b)
    np.testing.assert_allclose(a[:3], b[:3], rtol=2e-4, atol=1e-6)     # float32 rounding / atomics-order differences ...
    np.testing.assert_allclose(a, b, rtol=1e-2, atol=1e-5)             # ... which Adam amplifies step by step


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_channels_last_follows_the_planar_trajectory(G, graph):
    """The layout plan (mdx/layout.py) changes where the bytes of a map lie, not the numbers: six steps with every stage
    channels-last (csrc/norm_nhwc.hip, glue_nhwc.hip between MIOpen's NHWC convolutions), with a mixed plan (a transposing
    copy where stages of different layout meet) and with planar maps follow one trajectory -- eager and captured."""
    planar, n0, _ = _trainer_losses(graph, channels_last="none")
    nhwc, n1, tr = _trainer_losses(graph, channels_last="all")
    mixed, n2, _ = _trainer_losses(graph, channels_last="stem,layer2,decoder")
    assert n0 == n1 == n2 == 6
    assert tr.setting.channels_last_stages == frozenset(("stem", "layer1", "layer2", "layer3", "layer4", "decoder", "pose"))
    enc = tr.setting.raw_model["encoder"].encoder
    assert enc.layer2[0].conv1.weight.is_contiguous(memory_format=torch.channels_last) and not enc.layer2[0].conv1.weight.is_contiguous()
    _same_trajectory(nhwc, planar)
    _same_trajectory(mixed, planar)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("amp", ["none", "bf16"])
def test_pose_network_beside_depth_network_follows_the_sequential_trajectory(G, graph, amp):
    """opt.overlap_pose (default): the separate pose network runs on a side stream beside the depth network -- forward in
    trainer.batch_process, backward because autograd runs a node's backward on its forward's stream -- and joins before the loss
    kernels.  Same numbers as one network after the other, eager and captured (fork / join become graph edges), fp32 and bf16."""
    seq, n0, tr0 = _trainer_losses(graph, overlap_pose=False, amp=amp)
    par, n1, tr1 = _trainer_losses(graph, overlap_pose=True, amp=amp)
    assert n0 == n1 == 6 and tr1._pose_stream is not None and tr0._pose_stream is None
    if amp == "none":
        _same_trajectory(par, seq)
    else:
        np.testing.assert_allclose(par, seq, rtol=5e-2, atol=1e-4)
    pa, pb = dict(tr1.setting.raw_model["pose_decoder"].named_parameters()), dict(tr0.setting.raw_model["pose_decoder"].named_parameters())
    for k in pa:       # six Adam steps of lr 1e-4: the weights agree far inside one step's change
        assert float((pa[k] - pb[k]).abs().max()) <= (6e-4 if amp == "none" else 3e-3), k


@pytest.mark.parametrize("graph", [False, True])
def test_bf16_weight_shadows_follow_the_autocast_trajectory(G, graph, monkeypatch):
    """mdx/shadow.py: the bf16 copies of the convolution weights made by ONE launch before the networks fork (and the weight
    gradients cast back by one launch at the end of backward) against autocast's cast around every convolution -- the same roundings,
    so the same losses up to the order of MIOpen's atomics; and the modules see their float32 parameters again afterwards."""
    import mdx.shadow as shadow
    made = []
    plain = shadow._CastAll.forward

    def spy(ctx, *masters):
        made.append(len(masters))
        return plain(ctx, *masters)
    monkeypatch.setattr(shadow._CastAll, "forward", staticmethod(spy))
    a, n0, tr0 = _trainer_losses(graph, amp="bf16", shadow_weights=True)
    assert made and made[0] >= 40, made                 # every convolution weight of the four networks, the one-channel heads left out
    del made[:]
    b, n1, tr1 = _trainer_losses(graph, amp="bf16", shadow_weights=False)
    assert not made and n0 == n1 == 6
    np.testing.assert_allclose(a, b, rtol=5e-2, atol=1e-4)
    np.testing.assert_allclose(a[:2], b[:2], rtol=2e-3, atol=1e-5)
    for net in tr0.setting.raw_model.values():
        for p in net.parameters():
            assert isinstance(p, torch.nn.Parameter) and p.dtype == torch.float32
            assert not p.requires_grad or (p.grad is not None and p.grad.dtype == torch.float32)


def test_trainer_graph_replay_matches_eager(G):
    """model_train.trainer with opt.graph: the step captured into one hipGraph and replayed follows the eager loop's
    TRAJECTORY -- the warm-up steps capture needs are undone (weights, batch-norm statistics and counters, Adam moments
    and step counts), so the first replay is step 1 -- and follows a learning-rate change (auto-masking off: its noise
    stream differs between a captured and an eager generator)."""
    eager, n_eager, _ = _trainer_losses(False)
    graph, n_graph, tr = _trainer_losses(True)
    assert tr._graphed is not None
    assert n_graph == n_eager == 6             # neither the warm-up nor the capture pass counts as a step
    _same_trajectory(graph, eager)
    step = [float(v["step"]) for v in tr.setting.optim["optimizer"].state.values()]
    assert step and all(v == 6.0 for v in step), sorted(set(step))
    # an eager step from the same state, on the stream the graph was captured on (its gradient-accumulation nodes live
    # there; on the default stream torch warns about the mismatch -- tools/diag_accgrad_warning.py: the trainer's own
    # flow, warm-up + capture + replays, raises no such warning)
    with torch.cuda.stream(tr._graphed.stream):
        o1 = float(tr._eager_step(dict(tr._graphed.static))["loss"].detach())
    torch.cuda.current_stream().wait_stream(tr._graphed.stream)
    o2 = float(tr._graphed(dict(tr._graphed.static))["loss"].detach())
    assert np.isfinite(o1) and np.isfinite(o2) and abs(o1 - o2) < 0.05 * abs(o1)


def test_trainer_mono_stereo_captured_step_matches_eager(G):
    """BASELINE configs[4] (frame_ids 0, -1, 1, "s": three source frames -- the training kernel's LOW form with its (u, v) ring in
    the workspace and its accumulators in LDS) through the trainer: six steps captured into a hipGraph and replayed follow the
    eager steps, with auto-masking on (the in-kernel noise offset is restored around the warm-up: same draws)."""
    fids = (0, -1, 1, "s")
    eager, n_e, tr_e = _trainer_losses(False, automask=True, frame_ids=fids)
    graph, n_g, tr_g = _trainer_losses(True, automask=True, frame_ids=fids)
    assert tr_g._graphed is not None and tr_e._graphed is None and n_e == n_g == 6
    assert len(tr_g.opt.frame_ids) == 4 and tr_g.compute.noise_offset() == 6
    _same_trajectory(graph, eager)
    assert eager[-1] < eager[0]


def test_trainer_host_noise_is_never_captured(G):
    """--noise cpu (the reference's torch.randn on the host + copy, processor.py:195) cannot be part of a hipGraph: the
    host draw would run once, at capture time.  The trainer then runs eager whatever opt.graph says, and the run is
    the eager run (same host RNG stream)."""
    eager, _, _ = _trainer_losses(False, automask=True, noise="cpu", n=4)
    asked, _, tr = _trainer_losses(True, automask=True, noise="cpu", n=4)
    assert tr._graphed is None and not tr.can_graph()
    _same_trajectory(asked, eager)


@pytest.fixture
def rccl_group_of_one():
    import torch.distributed as dist
    import bench
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % bench.free_port(), rank=0, world_size=1,
                            device_id=torch.device(torch.cuda.current_device()))
    yield dist
    dist.destroy_process_group()


def test_trainer_captured_step_with_rccl_exchange(G, rccl_group_of_one):
    """The data-parallel step on ONE GPU: a process group of one rank over RCCL makes `setting` build the flat gradient
    buffer and issue its bucketed all-reduce from inside backward (model_tool/parallel.py).  Eager and captured (the
    exchange inside the hipGraph) follow the single-process trajectory."""
    plain, _, _ = None, None, None
    eager, n_eager, tr_e = _trainer_losses(False)
    assert tr_e.setting.sync is not None and tr_e.setting.sync.backend == "nccl" and tr_e._graphed is None
    sync = tr_e.setting.sync
    assert len(sync.buckets) >= 2 and all(p.grad.data_ptr() == sync.flat.data_ptr() + 4 * sync.offsets[id(p)] for p in sync.params)
    graph, n_graph, tr_g = _trainer_losses(True)
    assert tr_g.setting.sync is not None and tr_g._graphed is not None and tr_g.can_graph()
    assert len(tr_g.setting.sync.buckets) == 1        # a captured step exchanges one bucket (model_tool/parallel.py)
    assert n_graph == n_eager == 6
    _same_trajectory(graph, eager)
    # the metrics of an epoch go through the job-wide mean (one all-reduce over RCCL)
    log = {k: [] for k in tr_g.control.metric_name}
    log["loss"] = [torch.tensor(1.0, device=G.DEV), torch.tensor(2.0, device=G.DEV)]
    assert abs(tr_g.control.epoch_means(log)["loss"] - 1.5) < 1e-12
    for tr in (tr_e, tr_g):
        tr.setting.sync.detach()


@pytest.mark.parametrize("amp", ["none", "bf16"])
def test_trainer_split_capture_keeps_rccl_out_of_the_graph(G, rccl_group_of_one, monkeypatch, amp):
    """What a multi-rank job runs by default: graph A = forward, backward, gradients gathered into the flat buffer; the
    all-reduce issued EAGERLY between the replays (no RCCL call inside a capture); graph B = Adam.  Forced here with a group of
    one rank (MDX_DP_SPLIT=1; with several ranks it is the form whenever MDX_DP_GRAPH is not 1): same trajectory as the eager
    data-parallel step, the exchange really runs (the flat buffer is what Adam reads), batch-norm counters advance."""
    eager, n_eager, tr_e = _trainer_losses(False, amp=amp)
    monkeypatch.setenv("MDX_DP_SPLIT", "1")
    calls = []
    real = rccl_group_of_one.all_reduce
    monkeypatch.setattr(rccl_group_of_one, "all_reduce", lambda *a, **k: (calls.append(torch.cuda.is_current_stream_capturing()), real(*a, **k))[1])
    split, n_split, tr_s = _trainer_losses(True, amp=amp)
    g = tr_s._graphed
    assert g is not None and g.split and g.graph_apply is not None and len(tr_s.setting.sync.buckets) == 1
    assert n_split == n_eager == 6
    assert calls and not any(calls), "an all-reduce ran inside a capture"
    assert sum(1 for c in calls) >= 6          # one per replayed step (+ the warm-up steps' bucketed ones)
    if amp == "none":
        _same_trajectory(split, eager)
    else:
        np.testing.assert_allclose(split, eager, rtol=5e-2, atol=1e-4)
    sync = tr_s.setting.sync
    assert all(p.grad.data_ptr() == sync.flat.data_ptr() + 4 * sync.offsets[id(p)] for p in sync.params)
    for tr in (tr_e, tr_s):
        tr.setting.sync.detach()


def test_synchronous_collectives_refuse_capture_on_gpu(G, rccl_group_of_one):
    """VERDICT r3 weak #6 (gpurun_out/r3b/dist_graph_blocking.err: core dump): a blocking collective inside a capture must
    raise in the caller -- and the process, the process group and the GPU must be usable afterwards."""
    from model_tool import parallel
    x = torch.ones(4, device=G.DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    lin = torch.nn.Linear(2, 2).to(G.DEV)                  # (built outside: an upload inside a capture is its own error)
    torch.cuda.synchronize()
    for call in (lambda: parallel.mean_over_ranks([1.0, 2.0], G.DEV), lambda: parallel.broadcast_state([lin])):
        g = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="synchronous collective"):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                x.add_(1.0)
                assert parallel.capturing()
                call()
        del g
    torch.cuda.synchronize()
    assert not parallel.capturing()
    # outside a capture the same calls work, over RCCL
    assert parallel.mean_over_ranks([1.0, 2.0], G.DEV) == [1.0, 2.0]
    parallel.broadcast_state([lin])
    y = torch.ones(3, device=G.DEV)
    rccl_group_of_one.all_reduce(y)
    torch.cuda.synchronize()
    assert float(y.sum()) == 3.0


def test_trainer_channels_last_data_parallel_fused_adam(G):
    """--channels_last + a process group: the flat buffer's gradient views carry the parameters' own strides, so the fused
    Adam pairs the right elements (contiguous views made it refuse the lists / mispair).  The data-parallel
    steps (1-rank RCCL group) follow the single-process steps."""
    import torch.distributed as dist
    import bench
    plain, _, tr_p = _trainer_losses(False, channels_last="all")
    assert tr_p.setting.sync is None
    assert any(p.dim() == 4 and not p.is_contiguous() for p in tr_p.setting.parameters)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % bench.free_port(), rank=0, world_size=1,
                            device_id=torch.device(torch.cuda.current_device()))
    try:
        # the same batches (a process group brings a DistributedSampler: another order)
        dp, _, tr_d = _trainer_losses(False, channels_last="all", batches=tr_p.last_batches)
        sync = tr_d.setting.sync
        assert sync is not None
        assert all(p.grad.stride() == p.stride() and p.grad.data_ptr() == sync.flat.data_ptr() + 4 * sync.offsets[id(p)]
                   for p in sync.params)
        _same_trajectory(dp, plain)
        for a, b in zip(tr_d.setting.parameters, tr_p.setting.parameters):
            assert a.stride() == b.stride()
            assert float((a - b).abs().max()) <= 6 * 2e-4       # six Adam steps of lr 1e-4, sign flips near zero allowed
        sync.detach()
    finally:
        dist.destroy_process_group()


def test_trainer_bf16_networks_hand_float32_to_the_loss_path(G):
    """configs[2] / configs[3] run the networks under bf16 autocast (VERDICT r3 weak #1c).  One step: what the networks hand
    the loss path is float32 after the driver's casts, the kernels' indices on THOSE tensors are the oracle's bit for bit
    and the loss is the oracle's; the gradient reaches the bf16 networks.  Then six steps, captured and eager."""
    import importlib
    from oracle import oracle as orc
    bench = importlib.import_module("bench")
    from model_train import trainer
    torch.manual_seed(0)
    B, H, W = 2, 64, 96
    opt = bench.make_opt(B, height=H, width=W, amp="bf16")
    opt.synthetic_length, opt.max_steps, opt.miopen_find, opt.graph = 8, 0, False, False
    tr = trainer(opt)
    tr.setting.set_train()
    inputs = bench.one_batch(tr.setting, G.DEV)
    rng = np.random.RandomState(3)
    noises = [rng.randn(B, 2, H, W).astype(np.float32) for _ in opt.scales]
    for s in opt.scales:
        inputs[("noise", s)] = G.t(noises[s])
    outputs = tr.batch_process(inputs)
    assert any(outputs[("disp", s)].dtype == torch.bfloat16 for s in opt.scales) or \
        outputs["features"][-1].dtype == torch.bfloat16, "the networks did not run under bf16 autocast"
    Kn, iKn = inputs[("K", 0)].cpu().numpy(), inputs[("inv_K", 0)].cpu().numpy()
    tgt = inputs[("color", 0, 0)].cpu().numpy()
    srcs = [inputs[("color", f, 0)].cpu().numpy() for f in opt.frame_ids[1:]]
    for f in opt.frame_ids[1:]:
        assert outputs[("c2c", f, 0)].dtype == torch.float32
    assert outputs[("P", 0)].dtype == torch.float32 and outputs["loss"].dtype == torch.float32
    P_ref = np.stack([orc.compose_projection(Kn, outputs[("c2c", f, 0)].detach().cpu().numpy()) for f in opt.frame_ids[1:]])
    G.assert_bitexact(outputs[("P", 0)], P_ref, "P from the bf16 pose network's float32 matrices")
    total = 0.0
    for s in opt.scales:
        disp = outputs[("disp", s)].detach().float().cpu().numpy()        # the very tensor compute_loss hands the kernel
        ref = orc.photometric_fwd(disp, tgt, srcs, iKn, P_ref, noises[s])
        assert (outputs[("automask", s)].cpu().numpy() == ref["idx"]).all(), "auto-mask indices s%d (bf16 networks)" % s
        if s == 0:
            G.assert_bitexact(outputs[("depth", 0, 0)], ref["depth"], "depth")
        sm = orc.smooth_loss(disp, inputs[("color", 0, s)].cpu().numpy())
        total += ref["sum"] / float(B * H * W) + opt.disp_smoothness * sm / (2 ** s)
    G.assert_close(outputs["loss"], np.float64(total / len(opt.scales)), "loss (bf16 networks, float32 loss path)", rel=1e-5)
    outputs["loss"].backward()
    for key in ("encoder", "decoder", "pose_encoder", "pose_decoder"):
        gs = [p.grad for p in tr.setting.raw_model[key].parameters() if p.requires_grad]
        assert all(g is not None and g.dtype == torch.float32 and bool(torch.isfinite(g).all()) for g in gs), key
        assert any(float(g.abs().max()) > 0 for g in gs), key
    del tr
    eager, n_e, _ = _trainer_losses(False, amp="bf16")
    graph, n_g, trg = _trainer_losses(True, amp="bf16")
    assert trg._graphed is not None and n_e == n_g == 6
    assert all(np.isfinite(eager)) and all(np.isfinite(graph))
    np.testing.assert_allclose(graph[:3], eager[:3], rtol=5e-3, atol=1e-5)      # bf16 convolutions: non-deterministic kernels,
    np.testing.assert_allclose(graph, eager, rtol=5e-2, atol=1e-4)              # 8 bits of mantissa; Adam amplifies


def test_in_kernel_noise_offset_is_run_state(G, tmp_path, monkeypatch):
    """the {seed, offset} of the in-kernel noise generator is part of the run's state.  (1) graphed_step's
    warm-up puts the offset back: after n replays it is n, and -- auto-masking ON, noise drawn in the kernel -- the
    captured run follows the eager one (same draws).  (2) control.save stores it, control.resume restores it."""
    eager, _, tr_e = _trainer_losses(False, automask=True)
    graph, _, tr_g = _trainer_losses(True, automask=True)
    assert tr_g._graphed is not None and tr_g.compute.draws_in_kernel()
    assert tr_e.compute.noise_offset() == 6 and tr_g.compute.noise_offset() == 6
    _same_trajectory(graph, eager)
    monkeypatch.chdir(tmp_path)
    log = {k: [] for k in tr_e.control.metric_name}
    tr_e.opt.epoch = 2
    tr_e.control.save(1, log, log, tr_e.setting, compute=tr_e.compute)
    _, _, fresh = _trainer_losses(False, automask=True, n=1)
    assert fresh.compute.noise_offset() == 1
    fresh.opt.epoch = 2
    assert fresh.control.resume(fresh.setting, 2, compute=fresh.compute) == 2
    assert fresh.compute.noise_offset() == 6


def test_trainer_on_kitti_tree_gpu_image_prep_equals_pillow_loader(G, tmp_path):
    """model_train.trainer on a KITTI-raw tree (JPEG frames): with gpu_image_prep the workers hand over decoded frames
    and csrc/imgproc.hip builds the step's entries -- the SAME entries, so the training steps give the Pillow loader's
    losses (same random draws; auto-masking off: its noise is drawn on the device).  Then the loop
    people run: prefetcher + graph replay + validation, with worker processes."""
    import importlib
    import os
    import random
    import fake_kitti
    from model_train import trainer
    bench = importlib.import_module("bench")
    names = fake_kitti.make(str(tmp_path), n_frames=10)
    os.makedirs(os.path.join(str(tmp_path), "splits", "fake"))
    for split in ("train", "val", "test"):
        open(os.path.join(str(tmp_path), "splits", "fake", split + "_files.txt"), "w").write("\n".join(names) + "\n")

    def make(prep, graph, workers):
        opt = bench.make_opt(2, height=192, width=640, workers=workers)
        opt.dataset, opt.datapath, opt.datatype = "kitti_mono", str(tmp_path), "fake"
        opt.splits = os.path.join(str(tmp_path), "splits")
        opt.use_automasking, opt.graph, opt.max_steps, opt.miopen_find = False, graph, 3, False
        opt.gpu_image_prep, opt.uint8_loader, opt.collate_step_keys = prep, True, True
        return opt

    losses, entries = {}, {}
    for prep in ("true", "false"):
        torch.manual_seed(0)
        random.seed(0)
        tr = trainer(make(prep, False, 0))
        tr.setting.set_train()
        assert bool(getattr(tr.setting.train_dataloader.dataset, "gpu_prep", False)) == (prep == "true")
        out = []
        for step, b in enumerate(tr.batches(tr.setting.train_dataloader)):
            if prep == "true":
                assert ("color", 0, 0) in b and b[("color", 0, 0)].is_cuda and ("raw", 0) not in b    # prepared by the prefetcher
            out.append(float(tr.train_step(b)["loss"].detach()))
            if step == 0:       # what the step consumed (the Pillow path's uint8 entries were divided by 255 on the GPU, in place)
                entries[prep] = {k: v.detach().clone() for k, v in b.items()
                                 if isinstance(k, tuple) and k[0] in ("color", "color_aug", "K", "inv_K") and tr.compute._step_reads(k)}
            if step == 2:
                break
        losses[prep] = out
    # (1) the ENTRIES are identical, bit for bit -- this is the claim; the losses only follow from it
    assert set(entries["true"]) == set(entries["false"]) and len(entries["true"]) >= 11
    for k in entries["true"]:
        a, b_ = entries["true"][k], entries["false"][k]
        assert a.dtype == b_.dtype == torch.float32 and a.shape == b_.shape, (k, a.dtype, b_.dtype, a.shape, b_.shape)
        assert torch.equal(a, b_), ("entry differs between the GPU and the Pillow path", k, int((a != b_).sum()),
                                    a.numel(), float((a - b_).abs().max()))
    # (2) the losses agree as far as MIOpen lets two runs of ONE trainer on ONE batch agree: its forward convolution
    # kernel for encoder.layer2.0.conv1 is not run-to-run deterministic (profiles/r03_diag_two_trainers.json,
    # tools/diag_two_trainers.py: bit-equal weights and inputs, losses 0.20923434 ... 0.20923445 over eight passes);
    # later steps inherit the difference through the weights
    assert abs(losses["true"][0] - losses["false"][0]) <= 5e-5 * abs(losses["false"][0]), losses
    assert np.allclose(losses["true"], losses["false"], rtol=2e-4), losses
    # the full loop: worker processes, side-stream upload + preparation, hipGraph replay, validation, checkpoint
    torch.manual_seed(0)
    opt = make("true", True, 2)
    opt.save = os.path.join(str(tmp_path), "ckpt")
    tr = trainer(opt)
    tr.train()
    assert tr._graphed is not None


def test_trainer_restarts_from_its_checkpoint(G, tmp_path, monkeypatch):
    """--resume: a trainer started with resume=2 loads the files the first run wrote after its second epoch and runs
    only the remaining epoch (graph replay on: the captured Adam reads the restored moments and learning rate)."""
    import importlib
    from model_train import trainer
    bench = importlib.import_module("bench")
    monkeypatch.chdir(tmp_path)

    def make(epochs, resume):
        opt = bench.make_opt(2, height=64, width=96)
        opt.synthetic_length, opt.max_steps, opt.miopen_find, opt.graph = 8, 2, False, True
        opt.epoch, opt.scheduler_step, opt.save, opt.resume = epochs, 1, "restart", resume
        return opt

    torch.manual_seed(0)
    first = trainer(make(2, 0))
    first.train()
    saved = {k: v.clone() for k, v in first.setting.raw_model["decoder"].state_dict().items()}
    torch.manual_seed(5)
    second = trainer(make(3, 2))
    epochs_run = []
    original = second.control.print
    second.control.print = lambda epoch, *a: (epochs_run.append(epoch), original(epoch, *a))[1]
    second.train()
    assert epochs_run == [2]                                           # only the third epoch ran
    assert second.setting.optim["scheduler"].last_epoch == 3
    moved = [k for k, v in second.setting.raw_model["decoder"].state_dict().items() if not torch.equal(v, saved[k])]
    assert moved                                                       # it trained on from the restored weights
    lr = second.setting.optim["optimizer"].param_groups[0]["lr"]
    assert abs(float(lr) - 1e-4 * 0.1 ** 3) < 1e-12
