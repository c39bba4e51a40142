#!/usr/bin/env python3
"""bench.py -- training throughput of the self-supervised depth step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = BASELINE.json configs[1]: ResNet-18 depth encoder/decoder + separate ResNet-18 pose network
(two source frames), the fused photometric loss at 4 scales, backward, Adam -- fp32, 192x640, batch 12 per
GPU, synthetic KITTI-shaped inputs resident in HBM.  Data parallel: one process per GPU, DDP over RCCL;
per-GPU work is fixed (weak scaling).  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline      the dominant hand-written kernel: mdx::photometric_train_kernel (all scales, forward + gradient, one
                launch per step).  achieved = algorithmic bytes per launch (per scale, forward + backward
                formula, summed over the scales one launch processes) / launch duration; the duration is the mean over
                every launch of the timed steps, from HIP events the library records on the launch stream right before
                and after the kernel (mdx_photometric_train's timing hook).  The kernel is VALU-issue bound, not
                HBM bound: the issue-side numbers of the committed PMC pass stand next to the HBM fraction.
  cpu_baseline  the same step on the host cores: torch-CPU networks + the CPU oracle for the loss path
                (kind "port"), on a bounded sample (batch 4, configs[0])
"""
import argparse
import importlib
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_opt(batch, height=192, width=640, frame_ids=(0, -1, 1), num_layers=18, amp="none", workers=0):
    o = types.SimpleNamespace()
    o.dataset, o.datatype, o.datapath, o.splits = "synthetic", "kitti_eigen_zhou", "", ""
    o.batch, o.height, o.width = batch, height, width
    o.scales, o.frame_ids = [0, 1, 2, 3], list(frame_ids)
    o.min_depth, o.max_depth, o.disp_smoothness = 0.1, 100.0, 1e-3
    o.use_automasking, o.pose_type, o.pose_frames = True, "separate", "pair"
    o.num_layers, o.weight_init = num_layers, False          # random init: no checkpoints offline
    o.learning_rate, o.scheduler_step, o.epoch, o.save = 1e-4, 15, 1, "bench"
    o.num_workers, o.synthetic_length = workers, 2 * batch
    o.fused, o.noise, o.amp, o.channels_last = True, "device", amp, "auto"
    o.fused_train = True
    return o


def one_batch(setting, device):
    batch = next(iter(setting.train_dataloader))
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def alg_bytes(B, H, W, S, scale, bwd=False):
    """target RGB + S source RGB (each unique byte once) + uint8 index + low-res disparity."""
    n_lo = B * (H >> scale) * (W >> scale)
    return B * H * W * (12 + 12 * S + 1) + n_lo * (8 if bwd else 4)


def cpu_baseline(batch=4, steps=10):
    """The same training step on the host: torch-CPU networks + the CPU oracle for the loss path (10 timed steps of
    batch 4, ~10-20 s), and -- SURVEY 8d's two loss-path lines -- the loss path alone (forward + backward, networks
    excluded) through the oracle and through the plain-PyTorch composite of tests/torch_composite.py."""
    from oracle import oracle as orc
    from model_layer import ResnetEncoder, DepthDecoder, PoseDecoder, param2matrix
    from model_tool.synthetic import SyntheticKITTI
    H, W, S = 192, 640, 2
    # the GPU box gives one GPU's share of the host: 16 cores (gpurun); never oversubscribe
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)   # the oracle's OpenMP loops
    except OSError:
        pass
    torch.manual_seed(0)
    enc, pose_enc = ResnetEncoder(18, False), ResnetEncoder(18, False, 2)
    dec, pose_dec = DepthDecoder(enc.num_ch_enc), PoseDecoder(pose_enc.num_ch_enc, 1, 2)
    params = [p for m in (enc, dec, pose_enc, pose_dec) for p in m.parameters()]
    optim = torch.optim.Adam(params, 1e-4)
    ds = SyntheticKITTI(batch, [0, -1, 1], H, W)
    items = [ds[i] for i in range(batch)]
    inputs = {k: torch.stack([it[k] for it in items]) for k in items[0]}
    tgt = inputs[("color", 0, 0)].numpy()
    srcs = [inputs[("color", f, 0)].numpy() for f in (-1, 1)]
    invK = inputs[("inv_K", 0)].numpy()
    n = batch * H * W

    class OracleScale(torch.autograd.Function):
        @staticmethod
        def forward(ctx, disp, P, noise):
            out = orc.photometric_fwd(disp.numpy(), tgt, srcs, invK, P.numpy(), noise)
            ctx.save_for_backward(disp, P)
            ctx.idx = out["idx"]
            return torch.tensor(out["sum"] / n, dtype=torch.float32)

        @staticmethod
        def backward(ctx, g):
            disp, P = ctx.saved_tensors
            gd, gP = orc.photometric_bwd(disp.numpy(), tgt, srcs, invK, P.numpy(), ctx.idx, float(g) / n)
            return torch.from_numpy(gd), torch.from_numpy(gP), None

    class OracleSmooth(torch.autograd.Function):
        @staticmethod
        def forward(ctx, disp, color):
            v, g = orc.smooth_loss(disp.numpy(), color.numpy(), need_grad=True)
            ctx.g = torch.from_numpy(g)
            return torch.tensor(v, dtype=torch.float32)

        @staticmethod
        def backward(ctx, g):
            return ctx.g * g, None

    rng = np.random.RandomState(0)
    t0 = None
    for step in range(steps + 1):
        if step == 1:
            t0 = time.perf_counter()
        feats = enc(inputs[("color_aug", 0, 0)])
        disps = dec(feats)
        Ps = []
        for f in (-1, 1):
            a, b = (f, 0) if f < 0 else (0, f)
            pin = torch.cat([inputs[("color_aug", a, 0)], inputs[("color_aug", b, 0)]], 1)
            aa, tr = pose_dec([pose_enc(pin)])
            T = param2matrix(aa[:, 0], tr[:, 0], invert=(f < 0))
            Ps.append(torch.matmul(inputs[("K", 0)], T)[:, :3, :])
        P = torch.stack(Ps)
        loss = 0
        for s in range(4):
            noise = rng.randn(batch, S, H, W).astype(np.float32)
            loss = loss + OracleScale.apply(disps[("disp", s)], P, noise)
            loss = loss + 1e-3 * OracleSmooth.apply(disps[("disp", s)], inputs[("color", 0, s)]) / (2 ** s)
        loss = loss / 4
        optim.zero_grad()
        loss.backward()
        optim.step()
    dt = time.perf_counter() - t0
    out = {"value": batch * steps / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "batch %d (configs[0]) x %d steps after 1 warm-up: torch-CPU nets + oracle loss path" % (batch, steps)}
    # the loss path alone, forward + backward, on fixed network outputs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_composite as tc
    dd = {s: disps[("disp", s)].detach().clone().requires_grad_(True) for s in range(4)}
    Tm = [torch.eye(4).repeat(batch, 1, 1).requires_grad_(True) for _ in range(S)]
    colors = {s: inputs[("color", 0, s)] for s in range(4)}
    srcs_t = [inputs[("color", f, 0)] for f in (-1, 1)]
    Kt, invKt = inputs[("K", 0)], inputs[("inv_K", 0)]

    def run_composite():
        loss, _ = tc.loss_path(dd, colors, srcs_t, Kt, invKt, Tm)
        loss.backward()

    def run_oracle():
        Pm = torch.stack([torch.matmul(Kt, T)[:, :3, :] for T in Tm])
        loss = 0
        for s in range(4):
            noise = rng.randn(batch, S, H, W).astype(np.float32)
            loss = loss + OracleScale.apply(dd[s], Pm, noise) + 1e-3 * OracleSmooth.apply(dd[s], colors[s]) / (2 ** s)
        (loss / 4).backward()
    for name, fn, reps in (("torch_composite", run_composite, 4), ("oracle", run_oracle, 6)):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out["loss_path_" + name] = {"value": batch * reps / (time.perf_counter() - t0), "unit": "images/s",
                                    "sample": "loss path only (4 scales, forward + backward), batch %d x %d" % (batch, reps)}
    # BASELINE configs[0] through the PRODUCT: setting / compute on the "cpu" device (the reference's device pick on a machine
    # without a GPU, model_train.py:28) -- torch-CPU networks + the package's plain-PyTorch op composite (mdx/composite.py),
    # the reference-shaped op-by-op path; nothing from oracle/
    from model_tool import setting, compute
    opt = make_opt(batch, height=H, width=W)
    opt.synthetic_length = batch
    torch.manual_seed(0)
    st, cp = setting(opt, "cpu"), compute(opt, "cpu")
    st.set_train()
    reps = 3
    for step in range(reps + 1):
        if step == 1:
            t0 = time.perf_counter()
        o = {}
        i, o = cp.forward_depth(dict(inputs), o, st)
        i, o = cp.forward_pose(i, o, st)
        i, o = cp.image2warping(i, o, st)
        o = cp.compute_loss(i, o, st)
        st.optim["optimizer"].zero_grad(set_to_none=True)
        o["loss"].backward()
        st.optim["optimizer"].step()
    out["product_cpu_step"] = {"value": batch * reps / (time.perf_counter() - t0), "unit": "images/s", "cores": torch.get_num_threads(),
                               "sample": "configs[0]: batch %d x %d steps after 1 warm-up through model_tool.setting / compute on "
                                         "device 'cpu' (torch-CPU nets + mdx/composite.py)" % (batch, reps)}
    return out


# written by tools/pmc_bench.sh + tools/pmc_to_json.py: one file per BASELINE workload shape (configs[1], [4], [3])
PMC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r05_bench_kernel_pmc.json", "r05_pmc_c4.json", "r05_pmc_c3.json")]
# written by tools/issue_bound.py (no GPU needed): the training kernel's loop mix priced with the measured issue cycles per class
ISSUE_BOUND_FILE = os.path.join(ROOT, "profiles", "r05_issue_bound.json")
SIMDS, CLOCK_MHZ = 1024, 2400.0          # MI355X: 256 CUs x 4 SIMDs; the clock the micro-benchmarks' cycle prices assume


def roofline_blocks(args, opt, frame_ids, tsum):
    """`roofline` (+ `roofline_other`) of the JSON line from the HIP-event timings of the timed steps' own launches."""
    from mdx import functional as F   # noqa: F401
    nS = len(frame_ids) - 1
    nsc = len(opt.scales)
    per_scale = {n: sum(alg_bytes(args.batch, opt.height, opt.width, nS, sc, bwd=(n == "bwd"))
                        for sc in range(nsc)) for n in ("fwd", "bwd")}
    k = {}
    if tsum.get("train", (0, 0))[1]:
        k["train"] = {"ms": 1e-3 * tsum["train"][0], "launches": tsum["train"][1],
                      "bytes": float(per_scale["fwd"] + per_scale["bwd"]),
                      "name": "mdx::photometric_train_kernel<%d>" % nS}
    for n, kn in (("fwd", "mdx::photometric_fwd_coef_kernel<%d>"), ("bwd", "mdx::photometric_bwd_coef_kernel<%d>")):
        if tsum.get(n, (0, 0))[1]:
            k[n] = {"ms": 1e-3 * tsum[n][0], "launches": tsum[n][1], "bytes": per_scale[n] / float(nsc),
                    "name": kn % nS}
    if not k:
        return {}
    for n in k:
        k[n]["GBs"] = k[n]["bytes"] / (k[n]["ms"] * 1e-3) / 1e9
    dom = max(k, key=lambda n: k[n]["ms"] * k[n]["launches"])
    # Counters of the committed rocprofv3 --pmc passes over THIS command's own launches (tools/pmc_bench.sh profiles
    # `python bench.py` with --kernel-include-regex on the photometric kernels: same tensors as the timed steps).  They
    # describe one build of the library: tied to it by the hash of libmdx_hip.so, or -- a rebuild elsewhere embeds other
    # paths -- by the hash of its sources and flags, and to the workload by its shape; anything else reports null rather
    # than a stale number.
    traffic = issue = None
    import hashlib
    import importlib.util
    from mdx import LIB_PATH
    spec = importlib.util.spec_from_file_location("_mdx_build", os.path.join(os.path.dirname(LIB_PATH), "build.py"))
    _build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(_build)
    try:
        want = [args.batch, opt.height, opt.width, nS, nsc]
        pmc = next(d for d in (json.load(open(f)) for f in PMC_FILES if os.path.exists(f)) if d.get("shape") == want)
        so = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]
        same_build = pmc.get("lib_sha16") == so or (pmc.get("source_sha16") == _build.source_sha16()
                                                    and "MDX_LIB" not in os.environ)
        ent = next((v for kk, v in pmc["kernels"].items() if kk.startswith(k[dom]["name"].split("<")[0])
                    and ("<%d" % nS) in kk), None)
        same_shape = pmc.get("shape") == [args.batch, opt.height, opt.width, nS, nsc]
        if ent and same_shape and same_build:
            traffic = ent.get("traffic_bytes")
            issue = {kk: ent[kk] for kk in ("valu_wave_insts", "valu_busy_us", "wait_any_frac",
                                            "wait_inst_frac", "active_frac", "kernel_us_profiled") if kk in ent}
            issue["source"] = pmc.get("source")
    except (OSError, ValueError, KeyError, StopIteration):
        pass
    launch_us = 1e3 * k[dom]["ms"]
    hbm_frac = k[dom]["GBs"] / HBM_PEAK_GBS
    # What bounds this kernel is vector-instruction ISSUE, not HBM: measured = SQ_INSTS_VALU / (SIMDs x launch
    # cycles), wave-instructions per SIMD-cycle, from the committed counter pass; bound = 1 / (mean issue cycles of the loop's
    # instruction mix), tools/issue_bound.py (static ISA mix x per-class cycles measured with tools/int_rate.hip).  Both are
    # tied to this build by the source hash and are null otherwise.
    valu_rate = bound = None
    if issue and issue.get("valu_wave_insts") and issue.get("kernel_us_profiled"):
        valu_rate = issue["valu_wave_insts"] / (SIMDS * issue["kernel_us_profiled"] * CLOCK_MHZ)
    try:
        ib = json.load(open(ISSUE_BOUND_FILE))
        if ib.get("source_sha16") == _build.source_sha16() and ("<%d" % nS) in k[dom]["name"] and "ILi%d" % nS in ib.get("kernel", ""):
            bound = ib["issue_bound_inst_per_simd_cycle"]
    except (OSError, ValueError, KeyError):
        pass
    out = {"roofline": {"kernel": k[dom]["name"],
                        "bound": "hbm",        # the roofline `frac` is quoted against (BASELINE.json: HBM roofline of this kernel)
                        "achieved": k[dom]["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_frac, "traffic": traffic,
                        # measured HBM bytes (the counters' FETCH + WRITE, corrected as the guide prescribes) over the same launch time
                        "traffic_frac": (None if traffic is None else traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS),
                        "traffic_source": (None if traffic is None else "committed rocprofv3 --pmc passes over this command (profiles/r05_*; "
                                           "tied to this library build and workload shape by hash), not measured in this run"),
                        "valu_inst_per_simd_cycle": valu_rate, "issue_bound_inst_per_simd_cycle": bound,
                        "issue_frac": (valu_rate / bound if valu_rate and bound else None),
                        "issue": issue,
                        "launch_us": launch_us, "alg_bytes_per_launch": k[dom]["bytes"],
                        "launches_timed": k[dom]["launches"],
                        "alg_bytes": "SURVEY 8d per scale (fwd: B*H*W*(12+12S+1)+B*h*w*4, bwd: ...+B*h*w*8), summed "
                                     "over the %d scale(s) x {fwd,bwd} one launch covers" % (nsc if dom == "train" else 1),
                        "timing": "HIP events recorded by the library around this kernel in the timed steps"}}
    others = [n for n in k if n != dom]
    if others:
        out["roofline_other"] = [{"kernel": k[n]["name"], "achieved": k[n]["GBs"], "frac": k[n]["GBs"] / HBM_PEAK_GBS,
                                  "launch_us": 1e3 * k[n]["ms"], "alg_bytes_per_launch": k[n]["bytes"]} for n in others]
    return out


def network_kernel_roofline(args, opt):
    """`roofline_other` entries of the hand-written kernels BETWEEN the convolutions (batch norm + add + ReLU, decoder glue):
    streaming kernels, for which the HBM roof is the right roof.  Measured live on the step's own largest maps in the layout
    the step uses: K calls captured in a hipGraph and replayed (GPU time per call including the gaps between its launches),
    bytes = every map once per pass that has to touch it (forward x [+ res] -> y: 2-3 N; backward dy, y, x -> dx [+ dres]:
    4-5 N).  The whole family per shape: tools/netbench.py -> profiles/r05_netbench_*.txt."""
    from mdx import functional as F
    from mdx.layout import parse_plan
    cl = "layer1" in parse_plan(opt.channels_last if opt.channels_last != "auto" else None) or opt.channels_last in ("auto", "all", True)
    dt = torch.bfloat16 if args.amp == "bf16" else torch.float32
    fmt = torch.channels_last if cl else torch.contiguous_format
    K, B = 8, args.batch

    def graph_us(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(K):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / (10 * K)

    out = []
    widths = (64, 64) if args.num_layers < 50 else (64, 256)
    for name, C, h, w, has_res in (("stem", widths[0], args.height // 2, args.width // 2, False),
                                   ("layer1", widths[1], args.height // 4, args.width // 4, True)):
        x = torch.randn(B, C, h, w, device="cuda").to(dt).contiguous(memory_format=fmt).requires_grad_(True)
        res = torch.randn(B, C, h, w, device="cuda").to(dt).contiguous(memory_format=fmt).requires_grad_(True) if has_res else None
        wt, bs = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        gy = torch.randn(B, C, h, w, device="cuda").to(dt).contiguous(memory_format=fmt)
        ins = [x, wt, bs] + ([res] if has_res else [])

        def fwd():
            return F.bn_act(x, wt, bs, rm, rv, 1e-5, 0.1, residual=res, relu=True)
        tf = graph_us(fwd)
        tfb = graph_us(lambda: torch.autograd.grad(fwd(), ins, gy))
        n = x.numel() * x.element_size()
        nbytes = n * ((2 + has_res) + (4 + has_res))
        out.append({"kernel": "mdx::%sbn_* forward + backward (%s map %dx%dx%dx%d, %s, %s)" % (
                        "nhwc::" if cl else "", name, B, C, h, w, "channels-last" if cl else "planar", str(dt).split(".")[-1]),
                    "achieved": nbytes / tfb / 1e3, "frac": nbytes / tfb / 1e3 / HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm",
                    "us_forward": tf, "us_backward": tfb - tf, "alg_bytes": nbytes,
                    "timing": "hipGraph replay of %d calls, HIP events, live in this run" % K})
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port=None):
    """The command that starts n ranks of this script on this node (one process per GPU, RCCL rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without an external launcher: start N worker processes as CHILDREN -- before this
    process has made any GPU call (torch.cuda.device_count() does not initialise the GPU; the process is never
    replaced) -- relay rank 0's JSON line and return the children's exit code."""
    import subprocess
    rehearsal = os.environ.get("MDX_DIST_BACKEND", "nccl") != "nccl" or "--selftest-launcher" in argv
    have = torch.cuda.device_count()
    if n > have and not rehearsal:
        sys.stderr.write("bench.py: --gpus %d asked for but this node exposes %d GPU(s); refusing to oversubscribe "
                         "(MDX_DIST_BACKEND=gloo rehearses the launch path with several ranks on one GPU)\n" % (n, have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    if "--eager" not in argv and os.environ.get("MDX_HW_QUEUES", "") != "0":
        # the ranks run the captured step: two hardware queues (model_train.py; measured with a process group of one rank:
        # resident 742 / loop 722 images/s at 2 queues, 744 / 669 at 4, 514 / 723 at 8)
        env.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("MDX_HW_QUEUES") or "2")
    return subprocess.run(launch_command(n, argv), env=env).returncode


def selftest_launcher(gpus):
    """No GPU: every rank joins a gloo group and the rank count is verified by an all-reduce (tests/test_bench_launcher.py)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.distributed.init_process_group("gloo")
    ones = torch.ones(1)
    if world > 1:
        torch.distributed.all_reduce(ones)
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_verified": int(ones[0]), "asked": gpus}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0 if int(ones[0]) == gpus else 3


def trainer_loop(args, frame_ids, steps, warmup, workers, raw):
    """The loop people run (model_train.trainer): batches from the DataLoader (worker processes, pinned memory, uploaded
    one step ahead on a side stream), train_step, and control.metric on every step -- beside the resident-input figure."""
    from model_train import trainer
    opt = make_opt(args.batch, height=args.height, width=args.width, frame_ids=frame_ids, num_layers=args.num_layers,
                   amp=args.amp, workers=workers)
    opt.fused_train = not args.per_scale_kernels
    opt.grad_comm, opt.bucket_mb = args.grad_comm, args.bucket_mb
    opt.channels_last = args.channels_last
    opt.overlap_pose = not args.no_overlap_pose
    world = int(os.environ.get("WORLD_SIZE", "1"))
    opt.synthetic_length = (steps + warmup + 4) * args.batch * world
    opt.synthetic_pool = 4 * args.batch          # the stand-in dataset must not be what is measured
    opt.uint8_loader = not args.float_loader     # colours uint8 through the host pipeline, x/255 on the GPU
    opt.collate_step_keys = not args.float_loader
    opt.graph = not args.trainer_eager
    # decoded KITTI-sized frames through the loader; flip / Lanczos pyramid / jitter / ToTensor on the GPU (csrc/imgproc.hip)
    opt.synthetic_raw, opt.gpu_image_prep = raw, "true" if raw else "false"
    if raw:
        opt.synthetic_pool = 2 * args.batch      # a decoded frame is 1.4 MB: keep the workers' pools small
    opt.max_steps, opt.miopen_find = 0, args.miopen_find
    if os.environ.get("MDX_SWITCH_INTERVAL"):
        sys.setswitchinterval(float(os.environ["MDX_SWITCH_INTERVAL"]))
    tr = trainer(opt)
    tr.setting.set_train()
    log = {k: [] for k in tr.control.metric_name}
    it = iter(tr.batches(tr.setting.train_dataloader))
    for _ in range(warmup):
        b = next(it)
        log = tr.control.metric(b, tr.train_step(b), log)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b = next(it)
        log = tr.control.metric(b, tr.train_step(b), log)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if torch.distributed.is_initialized():
        tmax = torch.tensor([dt], device=tr.device if tr.setting.sync.backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax[0])
    vals = tr.control.epoch_means(log)
    graphed = tr._graphed is not None
    exchange = None if tr.setting.sync is None else {"buckets": len(tr.setting.sync.buckets), "in_graph": graphed}
    del it, b
    # the persistent loader workers and the captured graph of this trainer must not outlive the measurement
    import gc
    tr._graphed = None
    tr.setting.train_dataloader = tr.setting.valid_dataloader = None
    del tr
    gc.collect()
    torch.cuda.synchronize()
    return {"value": world * args.batch * steps / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "n_gpus": world, "gradient_exchange": exchange, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
            "workers": workers, "uint8_loader": bool(opt.uint8_loader), "hip_graph": graphed,
            "raw_frames": raw, "what": "model_train.trainer: DataLoader (" + ("decoded 1242x375 frames, Lanczos pyramid / jitter / ToTensor on the GPU, " if raw else "prepared uint8 entries, ") + "pinned, side-stream upload) -> train_step -> control.metric every step", "abs_rel_monitor": vals.get("abs_rel")}


def trainer_loop_child(feed, port=None):
    """`python bench.py --trainer-loop-child raw|prepared <same shape arguments>` in a fresh process; its one JSON line.
    In a multi-rank job every rank starts its own child (never a re-exec) and the children form a job of their own on
    the next rendezvous port: the loop is measured with the gradient exchange in it."""
    import subprocess
    keep = [a for a in sys.argv[1:] if a not in ("--one-loop",)]
    cmd = [sys.executable, os.path.abspath(__file__), "--trainer-loop-child", feed] + keep
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    # (several ranks: the loop's step is captured only on request -- MDX_DP_GRAPH=1, which `--graph` sets -- see
    # model_tool/parallel.py: dp_graph_allowed; the eager data-parallel step is fastest at the runtime's default)
    if ((multi or "--dist" in keep) and "--trainer-eager" not in keep and os.environ.get("MDX_HW_QUEUES", "") != "0"):
        # what model_train.py does for a data-parallel run (see there): two hardware queues for the loop's process, so that
        # the graph's RCCL branch and the prefetcher's stream do not share one; this process (the resident step) keeps the default
        env.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("MDX_HW_QUEUES") or "2")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        env["MASTER_PORT"] = str(port)             # a free port rank 0 picked and broadcast to the job
    else:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420 if multi else 900)
    except subprocess.TimeoutExpired:
        return {"error": "trainer-loop child timed out"}
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    if out.returncode != 0 or (not lines and int(os.environ.get("RANK", "0")) == 0):
        return {"error": "trainer-loop child failed (rc %d): %s" % (out.returncode, out.stderr.decode()[-600:])}
    return json.loads(lines[-1]) if lines else {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed steps ...
    ap.add_argument("--warmup", type=int, default=20)      # ... after 20 warm-up steps (2.5 s in all at configs[1])
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--amp", type=str, default="none", choices=["none", "bf16"])
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--num-layers", type=int, default=18)
    ap.add_argument("--frame-ids", type=str, default="0 -1 1", help='e.g. "0 -1 1 s" for mono+stereo (configs[4])')
    ap.add_argument("--channels-last", type=str, nargs="?", const="all", default="auto",
                    help='stages of the networks with channels-last (NHWC) maps: "none", "all", "auto" (the default plan) or a list of '
                         'stem,layer1..layer4,decoder,pose (mdx/layout.py)')
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark (MIOpen find mode)")
    ap.add_argument("--no-miopen-db", action="store_true", help="ignore the shipped gfx950 find-db (MIOpen heuristics)")
    ap.add_argument("--graph", action="store_true",
                    help="(the default since round 5; kept so that older command lines still work) the step captured into a "
                         "hipGraph and replayed.  Several ranks: the split form -- forward + backward + gather in one graph, the "
                         "all-reduce issued eagerly, Adam in a second graph; MDX_DP_GRAPH=1 puts the exchange inside ONE graph")
    ap.add_argument("--eager", action="store_true", help="the step launched kernel by kernel instead of replayed from a hipGraph")
    ap.add_argument("--dist", action="store_true",
                    help="with --gpus 1: run the data-parallel path anyway (a process group of ONE rank over RCCL: flat "
                         "gradient buffer, bucketed all-reduce issued inside backward)")
    ap.add_argument("--bucket-mb", type=int, default=0,
                    help="size of a gradient all-reduce bucket (0 = one bucket for a captured step, 32 MB eager)")
    ap.add_argument("--grad-comm", type=str, default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce")
    ap.add_argument("--per-scale-kernels", action="store_true",
                    help="round-1 path: one fused forward + one backward kernel per scale instead of the one-launch "
                         "training kernel")
    ap.add_argument("--no-prologue", action="store_true",
                    help="round-2 form of the photometric path: identity kernel + torch.randn + per-scale target statistics "
                         "instead of the per-step prologue kernel (A/B)")
    ap.add_argument("--no-fused-tails", action="store_true",
                    help="round-4 form of the step's small ops (A/B): the scalar tail of the loss as torch ops, the pose head's output "
                         "sliced per frame into param2matrix + K @ T, the disparity heads as MIOpen convolution + bias + sigmoid, the pose "
                         "decoder's bias / ReLU / mean as ATen ops")
    ap.add_argument("--no-trainer-loop", action="store_true",
                    help="skip the second measurement (the DataLoader-fed trainer loop, reported as trainer_loop)")
    ap.add_argument("--float-loader", action="store_true",
                    help="trainer loop with the reference's float32 colours through the DataLoader (4x the host bytes)")
    ap.add_argument("--trainer-eager", action="store_true",
                    help="trainer loop without the hipGraph replay of the step (model_option --graph 0)")
    ap.add_argument("--one-loop", action="store_true",
                    help="trainer loop only with decoded frames (skip the second run fed with ready 192x640 entries)")
    ap.add_argument("--workers", type=int, default=0,
                    help="DataLoader workers of the trainer-loop measurement (0 = sized from the host share of this rank: "
                         "12 for fp32, up to 24 with --amp bf16 whose step consumes ~900 samples/s)")
    ap.add_argument("--no-overlap-pose", action="store_true",
                    help="pose network AFTER the depth network instead of beside it on a side stream (A/B: model_option --overlap_pose 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trainer-loop-child", type=str, default="", choices=["", "raw", "prepared"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))        # nothing has touched the GPU yet
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s -- start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    if args.selftest_launcher:
        raise SystemExit(selftest_launcher(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local = local % torch.cuda.device_count()          # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    if args.workers <= 0:
        cores = len(os.sched_getaffinity(0))
        try:                                   # a container's CPU quota (cgroup v2) counts, not the host's core count
            quota = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota[0] != "max":
                cores = max(1, min(cores, int(round(int(quota[0]) / int(quota[1])))))
        except (OSError, ValueError, IndexError):
            pass
        share = max(2, cores // max(1, min(world, torch.cuda.device_count())))
        # an fp32 step consumes ~750 samples/s, a bf16 step ~1500 (round 5); one loader set on 16 cores delivers 700-860 with 12
        # workers, 1117 with 16, 1214 with 24 (a few workers more than cores cover the decoders' waits: profiles/r04_loader_8ranks.json)
        args.workers = max(2, min(24, share + share // 2) if args.amp == "bf16" else min(16, share))
    backend = os.environ.get("MDX_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; "gloo" only to rehearse
    if world > 1 or args.dist:
        kw = {}
        if world == 1:
            kw = {"init_method": "tcp://127.0.0.1:%d" % free_port(), "rank": 0, "world_size": 1}
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(device), **kw)
        else:
            torch.distributed.init_process_group(backend, **kw)
    distributed = torch.distributed.is_initialized()
    pkg = importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
    miopen_db = None if args.no_miopen_db else pkg.install_miopen_db(rank)   # tuned conv solvers (gfx950 find-db)

    torch.manual_seed(1234 + rank)
    frame_ids = [t if t == "s" else int(t) for t in args.frame_ids.split()]
    if args.trainer_loop_child:
        res = trainer_loop(args, frame_ids, args.steps, args.warmup, args.workers, args.trainer_loop_child == "raw")
        if rank == 0:
            print(json.dumps(res))
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    from model_train import trainer, graphed_step
    opt = make_opt(args.batch, height=args.height, width=args.width, frame_ids=frame_ids, num_layers=args.num_layers,
                   amp=args.amp)
    opt.channels_last = args.channels_last
    opt.overlap_pose = not args.no_overlap_pose
    opt.fused_train = not args.per_scale_kernels
    opt.grad_comm, opt.bucket_mb = args.grad_comm, args.bucket_mb
    opt.prologue = not args.no_prologue
    opt.fused_tail = not args.no_fused_tails
    opt.shadow_weights = os.environ.get("MDX_SHADOW_WEIGHTS", "1") != "0"       # (A/B of mdx/shadow.py in bf16 runs)
    if os.environ.get("MDX_THIN_WGRAD", "1") == "0":                             # (A/B of csrc/thinconv_nhwc.hip)
        from model_layer.depth_decoder import DepthDecoder as _DD
        _DD.thin_wgrad = False
    if args.no_fused_tails:
        from model_layer.depth_decoder import DepthDecoder
        from model_layer.pose_decoder import PoseDecoder
        DepthDecoder.fused_heads = False
        DepthDecoder.thin_wgrad = False
        PoseDecoder.fused_tail = False
    # MIOpen picks the tuned solvers from the shipped find-db in immediate mode already (fp32: same images/s as find
    # mode).  Find mode proper (--miopen-find) returns at once on a db hit but searches for minutes on a miss, so it
    # is never on by default; bf16 networks gain from it (a few more solvers are only reachable through find).
    opt.miopen_find, opt.graph = args.miopen_find, not args.eager
    # the product's own step (model_train.trainer._eager_step / graphed_step) on one batch that stays resident in HBM
    tr = trainer(opt)
    st, cp = tr.setting, tr.compute
    st.set_train()
    inputs = one_batch(st, device)

    def step():
        return tr._eager_step(inputs)["loss"]

    def fence():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph = None
    if args.graph and not tr.can_graph():
        raise SystemExit("bench.py: --graph needs a capturable step (device-side noise, RCCL process group)")
    if not args.eager and tr.can_graph():
        # hipGraph (what model_train.py runs by default): side-effect-free warm-up on a side stream, capture of one step, then
        # every timed step is one graph launch (several ranks: two, with the all-reduce between them) -- ~1900 kernel launches
        # leave the host's critical path; since the pose network runs beside the depth network an eager step is host-bound
        graph = graphed_step(tr, inputs)
        for _ in range(args.warmup):
            graph({})

        def run():
            return graph({})["loss"]             # inputs stay where they are: replay only
    else:
        run = step
        for _ in range(args.warmup):
            step()
    # HIP events around every fused photometric kernel of the timed steps (recorded by the library on the launch
    # stream, right before / after the kernel): the roofline numbers below come from the very launches that are timed
    from mdx import functional as F
    timing = {"fwd": [], "bwd": [], "train": []} if (rank == 0 and not args.no_roofline and graph is None) else None
    fence()
    F.TIMING = timing
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    fence()
    dt = time.perf_counter() - t0
    F.TIMING = None
    if not args.no_roofline and graph is not None:
        # graph replay launches no Python: time the kernels over a few eager steps after the timed region instead
        # (every rank: the eager step contains the exchange)
        timing = {"fwd": [], "bwd": [], "train": []}
        F.TIMING = timing if rank == 0 else None
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        F.TIMING = None
    ranks_verified = 1
    rank_ms = [1e3 * dt / args.steps]
    if distributed:
        tdev = device if backend == "nccl" else "cpu"
        # every rank's own time over the timed steps: the line's value uses the MAX (the contract); min / max side by side
        # make a straggler visible in the first multi-GPU record
        mine = torch.tensor([dt], device=tdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        rank_ms = [1e3 * float(t[0]) / args.steps for t in every]
        tmax = torch.tensor([dt], device=tdev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax[0])
        ones = torch.ones(1, device=tdev)
        torch.distributed.all_reduce(ones)           # every rank really took part (RCCL when backend is nccl)
        ranks_verified = int(ones[0])
    loss_val = float(loss.detach())
    failed = False

    line = None
    if rank == 0:
        line = {
            "metric": "images/sec training, KITTI %dx%d batch %d/GPU" % (args.height, args.width, args.batch), "value": world * args.batch * args.steps / dt,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.amp == "none" else "bf16-nets/f32-loss", "data": "synthetic",
            "config": {"workload": "%skitti_eigen_zhou-shaped %dx%d, batch %d/GPU, ResNet%d depth + separate ResNet%d "
                                   "pose, frame_ids %s, 4 scales, automask, Adam, %s"
                                   % ("configs[1]: " if (args.height, args.width, args.num_layers, args.amp,
                                                         args.frame_ids) == (192, 640, 18, "none", "0 -1 1") else "",
                                      args.height, args.width, args.batch, args.num_layers, args.num_layers,
                                      str(frame_ids).replace(" ", ""), "fp32" if args.amp == "none" else "bf16 nets"),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world},
            "ranks_verified": ranks_verified, "final_loss": loss_val, "hip_graph": bool(graph is not None),
            "hip_graph_form": (None if graph is None else "split: forward+backward+gather | eager all-reduce | Adam" if graph.split
                               else "one graph" + (" incl. the gradient exchange" if st.sync is not None else "")),
            "miopen_find_db": bool(miopen_db),
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "argmax_rank": int(max(range(len(rank_ms)), key=rank_ms.__getitem__))},
            "channels_last": sorted(getattr(st, "channels_last_stages", ())),
            "pose_beside_depth": bool(tr._pose_beside_depth()),
        }
        if st.sync is not None:
            line["gradient_exchange"] = {"backend": "rccl" if st.sync.backend == "nccl" else st.sync.backend,
                                         "buckets": len(st.sync.buckets), "bytes": 4 * st.sync.flat.numel(),
                                         "comm_dtype": args.grad_comm, "in_graph": bool(graph is not None and not graph.split),
                                         "what": "flat gradient buffer; bucketed all-reduce(mean) issued from inside backward"}
        if not args.no_roofline and timing is not None:
            line.update(roofline_blocks(args, opt, frame_ids, F.timing_summary(timing)))
            try:
                line.setdefault("roofline_other", []).extend(network_kernel_roofline(args, opt))
            except Exception as exc:  # noqa: BLE001  (a side measurement never takes the line down)
                line.setdefault("roofline_other", []).append({"error": "network kernel measurement failed: %r" % (exc,)})
    if not args.no_trainer_loop:
        graph = None
        del tr, st, cp, inputs
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        # twice, each in its own child process per rank (a second loop in the same process inherits the first one's
        # loader workers and allocator state and measured 5 % low): fed with decoded 1242x375 frames (the data path of
        # a real run: pyramid / jitter / ToTensor are extra GPU work the resident figure does not contain) and with
        # ready entries (the loop alone)
        ports = torch.tensor([free_port(), free_port()] if rank == 0 else [0, 0], dtype=torch.int64,
                             device=device if (distributed and backend == "nccl") else "cpu")
        if distributed:
            torch.distributed.broadcast(ports, 0)      # the children of every rank meet on rank 0's choice
        for k, (key, feed) in enumerate((("trainer_loop", "raw"), ("trainer_loop_prepared_frames", "prepared"))):
            if key != "trainer_loop" and args.one_loop:
                continue
            res = trainer_loop_child(feed, port=int(ports[k]))
            # one rank: the line promises this measurement.  Several ranks: the contract's figure (the resident step, above)
            # stands on its own; a loop that did not run is reported in the line, not turned into a failed job
            failed = failed or ("error" in res and world == 1)
            if rank == 0:
                line[key] = res
                if "error" in res:
                    line["degraded"] = True       # a measurement this line names did not run (several ranks: reported, exit code 0)
                if "value" in res:
                    res["vs_resident"] = res["value"] / line["value"]
    if rank == 0:
        if not args.no_trainer_loop and world == 1:
            # the data layer's share: host cost per sample with and without the GPU image preparation
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import loader_cost
            line["image_prep"] = loader_cost.measure(samples=12, batch=args.batch, reps=10, height=args.height,
                                                     width=args.width, frames=[f for f in frame_ids], workers=args.workers)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if distributed:
        flag = torch.tensor([1.0 if failed else 0.0], device=device if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)     # every rank leaves with the same code
        failed = bool(flag[0] > 0)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if failed:
        raise SystemExit(4)       # a measurement this line promises did not run


if __name__ == "__main__":
    main()
