"""CPU: host-side mirror of the reference interface -- networks (shapes, state-dict keys), pose matrices
against golden vectors, options, the synthetic data contract, metrics."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

import goldens

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
import model_layer  # noqa: E402
import model_loss   # noqa: E402
from model_layer.warp import param2matrix, vector2translation, angle2rotation  # noqa: E402


def test_export_lists_match_reference():
    # reference: model_layer/__init__.py:1-11, model_loss/__init__.py:1-3, model_tool/__init__.py:1-3
    for name in ["ResnetEncoder", "DepthDecoder", "PoseCNN", "PoseDecoder", "interpolate", "grid_sample",
                 "disparity2depth", "param2matrix", "Depth2PointCloud", "PointCloud2Pixel"]:
        assert hasattr(model_layer, name), name
    for name in ["ReprojectionLoss", "SmoothLoss", "compute_depth_error", "compute_depth_metric"]:
        assert hasattr(model_loss, name), name
    import model_tool
    for name in ["setting", "control", "compute"]:
        assert hasattr(model_tool, name), name


def test_param2matrix_matches_reference_golden():
    a = goldens.api()
    aa, tr = torch.from_numpy(a["p2m_aa"]).requires_grad_(True), torch.from_numpy(a["p2m_tr"]).requires_grad_(True)
    for inv in (False, True):
        M = param2matrix(aa, tr, invert=inv)
        np.testing.assert_allclose(M.detach().numpy(), a["p2m_M_%d" % inv], rtol=0, atol=1e-7)
        ga, gt = torch.autograd.grad(M, (aa, tr), torch.from_numpy(a["p2m_gM_%d" % inv]))
        np.testing.assert_allclose(ga.numpy(), a["p2m_gaa_%d" % inv], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gt.numpy(), a["p2m_gtr_%d" % inv], rtol=1e-5, atol=1e-6)
    T = vector2translation(torch.tensor([[[1.0, 2.0, 3.0]]]))
    assert T.shape == (1, 4, 4) and T[0, 0, 3] == 1 and T[0, 2, 3] == 3 and T[0, 3, 3] == 1
    R = angle2rotation(torch.zeros(2, 1, 3))
    np.testing.assert_allclose(R.numpy(), np.repeat(np.eye(4, dtype=np.float32)[None], 2, 0), atol=1e-7)


def test_network_shapes_and_state_dict_keys():
    enc = model_layer.ResnetEncoder(18, False)
    feats = enc(torch.rand(1, 3, 64, 96))
    assert [f.shape[1] for f in feats] == [64, 64, 128, 256, 512]
    assert [tuple(f.shape[2:]) for f in feats] == [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    keys = enc.state_dict().keys()
    for k in ["encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer1.0.conv1.weight",
              "encoder.layer2.0.downsample.0.weight", "encoder.layer4.1.bn2.bias", "encoder.fc.weight"]:
        assert k in keys, k
    assert sum(p.numel() for p in enc.parameters()) == 11689512
    dec = model_layer.DepthDecoder(enc.num_ch_enc)
    out = dec(feats)
    assert sorted(out) == [("disp", s) for s in range(4)]
    assert [tuple(out[("disp", s)].shape) for s in range(4)] == [(1, 1, 64 >> s, 96 >> s) for s in range(4)]
    assert sum(p.numel() for p in dec.parameters()) == 3152724
    dk = list(dec.state_dict().keys())
    assert dk[0] == "decoder.0.conv.conv.weight" and dk[-1] == "decoder.13.conv.bias"
    penc = model_layer.ResnetEncoder(18, False, 2)
    assert penc.encoder.conv1.weight.shape == (64, 6, 7, 7)
    pdec = model_layer.PoseDecoder(penc.num_ch_enc, 1, 2)
    aa, tr = pdec([penc(torch.rand(2, 6, 64, 96))])
    assert aa.shape == (2, 2, 1, 3) and tr.shape == (2, 2, 1, 3)
    assert sum(p.numel() for p in pdec.parameters()) == 1314572
    assert list(pdec.state_dict().keys())[:2] == ["net.0.weight", "net.0.bias"]
    r50 = model_layer.ResnetEncoder(50, False)
    assert list(r50.num_ch_enc) == [64, 256, 512, 1024, 2048]
    with pytest.raises(ValueError):
        model_layer.ResnetEncoder(19, False)
    cnn = model_layer.PoseCNN(2)
    aa, tr = cnn(torch.rand(1, 6, 64, 96))
    assert aa.shape == (1, 1, 1, 3)


def test_options_parse_lists_and_defaults():
    from model_option import options
    o = options([])
    assert o.frame_ids == [0, -1, 1] and o.scales == [0, 1, 2, 3] and o.batch == 12 and o.learning_rate == 1e-4
    o = options(["--frame_ids", "0 -1 1 s", "--use_automasking", "false", "--amp", "bf16"])
    assert o.frame_ids == [0, -1, 1, "s"] and o.use_automasking is False and o.amp == "bf16"


def test_synthetic_dataset_contract():
    from model_tool.synthetic import SyntheticKITTI
    it = SyntheticKITTI(2, [0, -1, 1, "s"], 64, 96)[1]
    for f in (0, -1, 1, "s"):
        for s in range(4):
            assert it[("color", f, s)].shape == (3, 64 >> s, 96 >> s)
            assert it[("color_aug", f, s)].dtype == torch.float32
    assert it[("K", 0)].shape == (4, 4) and it[("inv_K", 0)].shape == (4, 4) and it["stereo"].shape == (4, 4)
    np.testing.assert_allclose((it[("K", 0)] @ it[("inv_K", 0)]).numpy(), np.eye(4), atol=1e-4)
    assert it[("depth", 0)].shape == (1, 375, 1242)


def test_depth_metrics():
    gt = np.random.RandomState(0).uniform(1, 80, 1000)
    pred = gt * np.random.RandomState(1).uniform(0.8, 1.2, 1000)
    n = model_loss.compute_depth_error(gt, pred, "numpy")
    t = model_loss.compute_depth_error(torch.from_numpy(gt), torch.from_numpy(pred), "torch")
    np.testing.assert_allclose(np.array(n), np.array([float(v) for v in t]), rtol=1e-6)
    assert abs(n[0] - np.mean(np.abs(gt - pred) / gt)) < 1e-12
    perfect = model_loss.compute_depth_error(gt, gt, "numpy")
    assert perfect[0] == 0 and perfect[4] == 1.0
    inputs = {("depth", 0): torch.zeros(2, 1, 375, 1242)}
    inputs[("depth", 0)][:, :, 200:300, 100:1100] = 10.0
    outputs = {("depth", 0, 0): torch.full((2, 1, 192, 640), 5.0)}
    import model_loss.model_metric as mm
    # 40 % of the crop window is valid here: more than the monitor's fixed-size buffer holds by default -> NaN, loudly
    assert all(np.isnan(float(v)) for v in model_loss.compute_depth_metric(inputs, outputs, "torch"))
    keep, mm.METRIC_CAPACITY = mm.METRIC_CAPACITY, 1.0
    try:
        m = model_loss.compute_depth_metric(inputs, outputs, "torch")
    finally:
        mm.METRIC_CAPACITY = keep
    assert float(m[0]) < 1e-6   # median scaling removes the factor 2


def test_pose_driver_pair_ordering_and_invert_flag():
    """compute.forward_pose (reference processor.py:99-114): [f,0] for f<0 (inverted), [0,f] for f>0."""
    from model_tool.processor import compute
    opt = types.SimpleNamespace(frame_ids=[0, -1, 1, "s"], pose_frames="pair", pose_type="separate", batch=1,
                                batch_pose_pairs=False)      # the reference's loop: one call per pair
    seen = []

    class Enc(torch.nn.Module):
        def forward(self, x):
            seen.append(x[:, ::3, 0, 0].clone())
            return [x]

    class Dec(torch.nn.Module):
        def forward(self, feats):
            return torch.full((1, 2, 1, 3), 0.01), torch.full((1, 2, 1, 3), 0.02)
    st = types.SimpleNamespace(model={"pose_encoder": Enc(), "pose_decoder": Dec()})
    inputs = {("color_aug", f, 0): torch.full((1, 3, 4, 4), float(v)) for f, v in ((0, 0.0), (-1, -1.0), (1, 1.0), ("s", 9.0))}
    _, out = compute(opt, "cpu").forward_pose(inputs, {}, st)
    assert seen[0].flatten().tolist() == [-1.0, 0.0] and seen[1].flatten().tolist() == [0.0, 1.0]
    assert ("c2c", "s", 0) not in out and out[("c2c", -1, 0)].shape == (1, 4, 4)
    aa, tr = torch.full((1, 1, 3), 0.01), torch.full((1, 1, 3), 0.02)
    np.testing.assert_allclose(out[("c2c", -1, 0)].numpy(), param2matrix(aa, tr, True).numpy())
    np.testing.assert_allclose(out[("c2c", 1, 0)].numpy(), param2matrix(aa, tr, False).numpy())
    # default: the same pairs, in the same order, stacked along the batch of ONE call
    opt.batch_pose_pairs = True
    del seen[:]

    class Dec2(torch.nn.Module):
        def forward(self, feats):
            n = feats[0][0].shape[0]
            return torch.full((n, 2, 1, 3), 0.01), torch.full((n, 2, 1, 3), 0.02)
    st = types.SimpleNamespace(model={"pose_encoder": Enc(), "pose_decoder": Dec2()})
    _, out2 = compute(opt, "cpu").forward_pose(inputs, {}, st)
    assert len(seen) == 1 and seen[0].tolist() == [[-1.0, 0.0], [0.0, 1.0]]
    for f in (-1, 1):
        np.testing.assert_allclose(out2[("c2c", f, 0)].numpy(), out[("c2c", f, 0)].numpy())


def test_miopen_find_db_install(tmp_path, monkeypatch):
    """mdx/tuning.py: a private writable copy of the shipped gfx950 find-db per rank; explicit env wins; opt-out."""
    import tempfile
    from mdx import tuning
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    d = tuning.install_miopen_db(rank=3)
    assert "_r3_" in os.path.basename(d) and os.environ["MIOPEN_USER_DB_PATH"] == d
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    d2 = tuning.install_miopen_db(rank=3)                   # a second job with the same rank: its own directory
    assert d2 != d and sorted(os.listdir(d2)) == sorted(os.listdir(d))
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", d)
    shipped = sorted(f for f in os.listdir(os.path.join(tuning.PACKAGE_DIR, "miopen_db")) if f.endswith(".txt"))
    assert shipped and sorted(os.listdir(d)) == shipped
    for name in shipped:   # MIOpen's text format: one "key=value" record per line, keyed by gfx950
        assert name.startswith(("gfx950", "batchnorm_gfx950"))
        with open(os.path.join(d, name)) as fh:
            assert all("=" in line for line in fh if line.strip())
    assert tuning.install_miopen_db(rank=0) == d            # explicit MIOPEN_USER_DB_PATH is respected
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("MDX_MIOPEN_DB", "0")
    assert tuning.install_miopen_db(rank=0) is None and "MIOPEN_USER_DB_PATH" not in os.environ
    # another MIOpen build than the one the db was recorded with: warn, report "no db" (MIOpen would ignore the files)
    monkeypatch.delenv("MDX_MIOPEN_DB")
    monkeypatch.setattr(tuning, "miopen_db_tag", lambda: "9_9_9_20990101-1-1-gdeadbeef00")
    with pytest.warns(UserWarning, match="another MIOpen build"):
        assert tuning.install_miopen_db(rank=0) is None


def test_batchnorm_host_counter_keeps_state_dict_contract():
    """model_layer.depth_encoder.BatchNorm2d: same numbers and state-dict keys as nn.BatchNorm2d; the
    num_batches_tracked buffer is current whenever the state dict is taken."""
    from model_layer.depth_encoder import BatchNorm2d
    torch.manual_seed(0)
    a, b = BatchNorm2d(5), torch.nn.BatchNorm2d(5)
    b.load_state_dict(a.state_dict())
    for _ in range(3):
        x = torch.randn(4, 5, 6, 7)
        torch.testing.assert_close(a(x), b(x))
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        torch.testing.assert_close(sa[k], sb[k])
    assert int(sa["num_batches_tracked"]) == 3
    a.eval(), b.eval()
    x = torch.randn(2, 5, 3, 3)
    torch.testing.assert_close(a(x), b(x))
    assert int(a.state_dict()["num_batches_tracked"]) == 3


def _pose_setting(seed=7):
    from model_layer import ResnetEncoder, PoseDecoder
    torch.manual_seed(seed)
    enc = ResnetEncoder(18, False, num_input_images=2).train()
    dec = PoseDecoder(enc.num_ch_enc, 1, 2).train()
    return types.SimpleNamespace(model={"pose_encoder": enc, "pose_decoder": dec})


def test_batched_pose_pairs_equal_the_per_pair_loop():
    """compute.forward_pose: both frame pairs through the separate pose network in ONE batch (batch norms grouped per
    pair) == the reference's loop of one call per pair (processor.py:61-83): poses, running statistics, gradients."""
    from model_tool.processor import compute
    opt = types.SimpleNamespace(frame_ids=[0, -1, 1], pose_frames="pair", pose_type="separate", batch=2)
    g = torch.Generator().manual_seed(1)
    inputs = {("color_aug", f, 0): torch.rand(2, 3, 64, 96, generator=g) for f in (0, -1, 1)}
    res = {}
    for batched in (True, False):
        st = _pose_setting()
        opt.batch_pose_pairs = batched
        cp = compute(opt, "cpu")
        _, out = cp.forward_pose(dict(inputs), {}, st)
        loss = sum(out[("c2c", f, 0)].square().sum() for f in (-1, 1))
        loss.backward()
        res[batched] = (out, st)
    (oa, sa), (ob, sb) = res[True], res[False]
    for f in (-1, 1):
        for key in ("R", "T", "c2c"):
            torch.testing.assert_close(oa[(key, f, 0)], ob[(key, f, 0)], rtol=1e-5, atol=1e-6)
    da, db = sa.model["pose_encoder"].state_dict(), sb.model["pose_encoder"].state_dict()
    for k in da:
        torch.testing.assert_close(da[k].float(), db[k].float(), rtol=1e-5, atol=1e-6, msg=k)
    assert int(da["encoder.bn1.num_batches_tracked"]) == 2
    for name in ("pose_encoder", "pose_decoder"):
        for (n, p), (_, q) in zip(sa.model[name].named_parameters(), sb.model[name].named_parameters()):
            if q.grad is None:
                continue
            err, scale = float((p.grad - q.grad).norm()), float(q.grad.norm())
            assert err <= 1e-3 * scale + 1e-7, "%s.%s: %g vs %g" % (name, n, err, scale)


def test_checkpoint_resume_restores_weights_optimizer_and_schedule(tmp_path, monkeypatch):
    """control.save writes the reference-named weight files (+ state<N>.pt); control.resume restores networks, Adam
    moments and the StepLR position: a step after the restart equals the step the uninterrupted run takes."""
    import types
    from model_tool import setting, compute, control
    from cpu_loss import cpu_compute_loss
    monkeypatch.chdir(tmp_path)

    def make(seed):
        o = types.SimpleNamespace(dataset="synthetic", datatype="x", datapath="", splits="", batch=2, height=64, width=96,
                                  scales=[0, 1, 2, 3], frame_ids=[0, -1, 1], min_depth=0.1, max_depth=100.0,
                                  disp_smoothness=1e-3, use_automasking=True, pose_type="separate", pose_frames="pair",
                                  num_layers=18, weight_init=False, learning_rate=1e-3, scheduler_step=1, epoch=4,
                                  save="resume_test", num_workers=0, synthetic_length=4, fused=True, noise="device",
                                  amp="none")
        torch.manual_seed(seed)
        return o, setting(o, "cpu"), compute(o, "cpu"), control(o, "cpu")

    def step(o, st, cp, inputs):
        out = {}
        i, out = cp.forward_depth(dict(inputs), out, st)
        i, out = cp.forward_pose(i, out, st)
        st.optim["optimizer"].zero_grad(set_to_none=True)
        cpu_compute_loss(o, i, out, seed=3).backward()
        st.optim["optimizer"].step()

    o, st, cp, ct = make(1)
    st.set_train()
    inputs = next(iter(st.train_dataloader))
    for _ in range(2):
        step(o, st, cp, inputs)
        st.optim["scheduler"].step()
    empty = {k: [] for k in ct.metric_name}
    ct.save(1, empty, empty, st)                                   # after the second epoch
    files = sorted(os.listdir(os.path.join("model_save", "resume_test")))
    assert {"encoder2.pt", "decoder2.pt", "pose_encoder2.pt", "pose_decoder2.pt", "state2.pt"} <= set(files)
    o2, st2, cp2, ct2 = make(99)                                   # different initial weights
    assert ct2.resume(st2, 2) == 2
    st2.set_train()
    assert st2.optim["scheduler"].get_last_lr() == st.optim["scheduler"].get_last_lr() != [1e-3]
    for key in st.raw_model:
        for (n, a), (_, b) in zip(st.raw_model[key].state_dict().items(), st2.raw_model[key].state_dict().items()):
            assert torch.equal(a, b), (key, n)
    step(o, st, cp, inputs)
    step(o2, st2, cp2, inputs)
    for key in st.raw_model:
        for (n, a), (_, b) in zip(st.raw_model[key].named_parameters(), st2.raw_model[key].named_parameters()):
            assert torch.allclose(a, b, rtol=0, atol=1e-7), (key, n)
    with pytest.raises(FileNotFoundError):
        ct2.resume(st2, 3)
