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


def _setup(G, c, fused, fused_train=True):
    from model_tool.processor import compute
    from model_layer import Depth2PointCloud, PointCloud2Pixel
    from model_loss import ReprojectionLoss, SmoothLoss
    opt = types.SimpleNamespace(scales=list(range(c.n_scales)), frame_ids=c.frame_ids, height=c.H, width=c.W,
                                min_depth=0.1, max_depth=100.0, disp_smoothness=1e-3, use_automasking=c.automask,
                                batch=c.B, pose_type="separate", pose_frames="pair", fused=fused, noise="device",
                                fused_train=fused_train)
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


@pytest.mark.parametrize("mode", ["train_kernel", "per_scale", "op_by_op"])
@pytest.mark.parametrize("name", goldens.CASES)
def test_compute_driver_vs_golden(G, name, mode):
    c = goldens.Case(name)
    fused = mode != "op_by_op"
    cp, st, inputs, outputs = _setup(G, c, fused, fused_train=(mode == "train_kernel"))
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
                m.setattr(F, "maxpool3s2", lambda x: torch.nn.functional.max_pool2d(x, 3, 2, 1))
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


def _trainer_losses(graph, automask=False, noise="device", n=6, lr_change_at=4):
    import importlib
    bench = importlib.import_module("bench")
    from model_train import trainer
    torch.manual_seed(0)
    opt = bench.make_opt(2, height=64, width=96)
    opt.use_automasking, opt.graph, opt.synthetic_length, opt.max_steps, opt.miopen_find = automask, graph, 16, 0, False
    opt.noise = noise
    tr = trainer(opt)
    tr.setting.set_train()
    batches = list(tr.setting.train_dataloader)[:n]
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
