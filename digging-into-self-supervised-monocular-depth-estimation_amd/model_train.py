"""Training entry (reference: model_train.py:24-101): `trainer(options()).train()`, one process per GPU.

    python model_train.py --dataset synthetic --epoch 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 model_train.py ...
"""
import os
import sys

# Data-parallel runs: the captured step's streams (the capture stream, the pose network's side stream, the gradient exchange, RCCL's
# own, the prefetcher's uploads + image preparation) are mapped onto the HIP runtime's hardware queues, 4 per device by default,
# and which two share one decides what overlaps.  Measured with a process group of one rank (round 5, pose network beside the
# depth network, images/s resident | DataLoader-fed loop): 2 queues 742 | 722 (bf16 1475 | 1412), 4 queues 744 | 669 (1487 |
# 1248), 8 queues 514 | 723; an EAGER data-parallel step is the opposite (2: 686, 4: 625, 8: 733) and host-bound besides.  So a
# captured data-parallel run asks for 2.  The runtime reads the variable when it is loaded, i.e. before `import torch`.
# MDX_HW_QUEUES=0 leaves the runtime's default, MDX_HW_QUEUES=n asks for n; a value the user exported (GPU_MAX_HW_QUEUES) is never
# overridden.  (Round 3's sweep, before the side streams: LABNOTES.md.)
def _graph_requested(argv):
    for i, a in enumerate(argv):
        v = a.split("=", 1)[1] if a.startswith("--graph=") else (argv[i + 1] if a == "--graph" and i + 1 < len(argv) else None)
        if v is not None:
            return str(v).lower() in ("1", "true", "yes")
    return True


# (only when this file is the program: imported by another one -- bench.py -- the runtime is loaded already and the variable
# would merely leak into that program's child processes)
if os.path.basename(sys.argv[0] or "") == "model_train.py" and os.environ.get("MDX_HW_QUEUES", "") != "0" and (
        os.environ.get("MDX_HW_QUEUES") or (int(os.environ.get("WORLD_SIZE", "1")) > 1 and _graph_requested(sys.argv[1:]))):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("MDX_HW_QUEUES") or "2")

import numpy as np        # noqa: E402
import torch              # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from model_option import options      # noqa: E402
from model_tool import *              # noqa: E402,F401,F403


class device_prefetcher(object):
    """Iterates a DataLoader one batch ahead: the entries a step reads (compute._step_reads) of batch i+1 are copied
    host -> device on a SIDE stream while step i computes (the loader pins its batches), so the PCIe transfer
    (~0.13 GB per batch of 12 at 192x640) is off the critical path.  The reference copies every key at the start of
    each step on the compute stream (processor.py:34-35)."""

    def __init__(self, loader, device, wanted, prepare=None):
        self.loader, self.device, self.wanted, self.prepare = loader, device, wanted, prepare
        self.stream = torch.cuda.Stream(device) if str(device).startswith("cuda") else None

    def _upload(self, batch):
        if self.stream is None or batch is None:
            return batch
        with torch.cuda.stream(self.stream):
            batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) and self.wanted(k) else v)
                     for k, v in batch.items()}
            # decoded frames -> the step's entries (compute.prepare), also on the side stream
            return self.prepare(batch) if self.prepare is not None else batch

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._upload(next(it, None))
        while nxt is not None:
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for v in nxt.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream(self.device))
            cur, nxt = nxt, self._upload(next(it, None))
            yield cur


class graphed_step(object):
    """One training step (networks, the photometric kernels launched through the C-ABI, the gradient all-reduce of a
    data-parallel job, fused Adam) captured into ONE hipGraph and replayed: ~1600 kernel launches leave the host's
    critical path, so the step stays GPU-bound while the host decodes and collates the next batches (measured: with 12
    loader workers alive the eager step needs 24.5 ms of host time for 20.7 ms of GPU time).  Inputs live in static
    device buffers the batch is copied into; the outputs the loop reads afterwards (loss, depth of scale 0, auto-masks)
    are the graph's own tensors.  What Python does during a step and a replay would skip is re-applied per replay: the
    batch-norm step counters.  Under torch.distributed the exchange (model_tool/parallel.py: bucketed RCCL all-reduce
    issued from inside backward) is part of the graph.  Any change of the learning rate goes through set_lr (a device
    tensor the captured Adam reads).

    The warm-up steps capture needs (MIOpen picks its kernels, the allocator settles, RCCL opens its communicator) are
    side-effect free: weights, batch-norm statistics, Adam moments, step counters and the offset of the in-kernel noise
    generator are put back afterwards, so the first replay is step 1 of the run -- graph on and graph off follow the same
    trajectory and draw the same noise."""

    def __init__(self, tr, example, warmup=3):
        self.tr = tr
        dev = tr.device
        # several ranks without MDX_DP_GRAPH=1: the SPLIT form -- graph A = forward, backward, gradients gathered into the flat
        # buffer; the all-reduce issued eagerly between the replays (RCCL never inside a capture: a captured multi-rank exchange
        # has not run on hardware); graph B = Adam.  Two graph launches and one collective per step on the host instead of ~1900
        # kernel launches (an eager data-parallel step is host-bound since the pose network runs beside the depth network).
        from model_tool.parallel import dp_graph_allowed
        sync = tr.setting.sync
        self.split = sync is not None and (not dp_graph_allowed(sync.world) or os.environ.get("MDX_DP_SPLIT", "") == "1")
        opt = tr.setting.optim["optimizer"]
        self.lr = torch.tensor(float(opt.param_groups[0]["lr"]), device=dev)
        for g in opt.param_groups:
            g["capturable"] = True
            g["lr"] = self.lr
        # the velodyne ground truth is read by control.metric from the batch itself, never by the captured step
        wanted = lambda k: tr.compute._step_reads(k) and not (isinstance(k, tuple) and k[0] == "depth")  # noqa: E731
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) and wanted(k) else v) for k, v in example.items()}
        self.copied = {k for k, v in example.items() if torch.is_tensor(v) and wanted(k)}
        from model_layer.depth_encoder import BatchNorm2d
        nets = list(tr.setting.raw_model.values())
        self.bns = [m for net in nets for m in net.modules() if isinstance(m, BatchNorm2d)]
        # ---- snapshot of everything a training step changes ----
        tensors = [t for net in nets for t in list(net.parameters()) + list(net.buffers())]
        saved = [t.detach().clone() for t in tensors]
        saved_bn = [m._pending_batches for m in self.bns]
        # the in-kernel noise generator's {seed, offset}: every warm-up step advances the offset on the device
        rng = tr.compute.noise_rng(dev) if tr.compute.draws_in_kernel() else None
        saved_rng = rng.tensor.clone() if rng is not None else None
        had_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                     for p, st in opt.state.items()}
        # ONE side stream for the warm-up and the capture: the gradient-accumulation nodes autograd creates during
        # the captured forward then live on the stream that produces their gradients
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                tr._eager_step(dict(self.static))
            with torch.no_grad():
                for t, v in zip(tensors, saved):
                    t.copy_(v)
                for p, st in opt.state.items():
                    old = had_state.get(id(p))
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if old is not None and k in old:
                                v.copy_(old[k])
                            else:
                                v.zero_()          # Adam's initial state: zero moments, step 0
                if rng is not None:
                    rng.tensor.copy_(saved_rng)
            for m, n in zip(self.bns, saved_bn):
                m._pending_batches = n
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        del saved, had_state
        before = [m._pending_batches for m in self.bns]
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the DataLoader's pin-memory thread keeps allocating pinned host buffers for the next batches
        # (and RCCL's watchdog thread polls events) while this thread captures; in the default "global" mode such a
        # call from ANY thread invalidates the capture
        self.graph_apply = None
        if self.split:
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                outputs = tr._step_gradients(dict(self.static))
            self.graph_apply = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_apply, stream=self.stream, capture_error_mode="thread_local", pool=self.graph.pool()):
                tr.setting.optim["optimizer"].step()
        else:
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                outputs = tr._eager_step(dict(self.static))
        # the graph's output tensors, without the autograd graph behind them: a loss that kept its grad_fn would keep the
        # capture pass's gradient-accumulation nodes (created on the capture stream) alive into later eager steps
        def _plain(v):
            if torch.is_tensor(v):
                return v.detach()
            if isinstance(v, (list, tuple)):
                return type(v)(_plain(x) for x in v)
            return v
        self.outputs = {k: _plain(v) for k, v in outputs.items()}
        del outputs
        self.bn_incr = [m._pending_batches - b for m, b in zip(self.bns, before)]
        for m, b in zip(self.bns, before):         # the capture pass ran no kernel: it was not a step
            m._pending_batches = b
        torch.cuda.synchronize(dev)

    def set_lr(self, value):
        self.lr.fill_(float(value))

    def __call__(self, inputs):
        for k, v in inputs.items():
            if k in self.copied and torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        if self.graph_apply is not None:
            self.tr.setting.sync.exchange()
            self.graph_apply.replay()
        for m, inc in zip(self.bns, self.bn_incr):
            m._pending_batches += inc
        return self.outputs


class trainer(object):
    def __init__(self, opt):
        self.opt = opt
        self._graphed = None
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()     # a rehearsal may run several ranks on one GPU (gloo backend)
            self.device = "cuda:%d" % local
            torch.cuda.set_device(local)
        else:
            self.device = "cpu"
        if world > 1 and not torch.distributed.is_initialized():
            if torch.cuda.is_available():
                torch.distributed.init_process_group(os.environ.get("MDX_DIST_BACKEND", "nccl"),
                                                     device_id=torch.device(self.device))
            else:
                torch.distributed.init_process_group("gloo")
        self.rank = int(os.environ.get("RANK", "0"))
        from mdx.tuning import install_miopen_db
        install_miopen_db(self.rank)          # tuned conv / batch-norm solvers for the default shapes (immediate mode)
        torch.backends.cudnn.benchmark = bool(getattr(opt, "miopen_find", False))
        self.setting = setting(opt, self.device)
        self.compute = compute(opt, self.device)
        self.control = control(opt, self.device)

    _pose_stream = None

    def _pose_beside_depth(self):
        return (bool(getattr(self.opt, "overlap_pose", True)) and str(self.device).startswith("cuda")
                and self.opt.pose_type in ("separate", "posecnn") and torch.is_grad_enabled())

    def batches(self, loader):
        """The loader's batches, uploaded one step ahead on a side stream (GPU) or as they are (CPU)."""
        if str(self.device).startswith("cuda") and getattr(self.opt, "device_prefetch", True):
            return device_prefetcher(loader, self.device, self.compute._step_reads, self.compute.prepare)
        return loader

    def batch_process(self, inputs):
        # bf16 networks: every convolution weight cast ONCE, by one launch, before the two networks fork (and the weight gradients
        # back by one launch at the end of backward) instead of by autocast around each convolution (mdx/shadow.py)
        if (self.compute.amp == "bf16" and str(self.device).startswith("cuda") and torch.is_grad_enabled()
                and getattr(self.opt, "shadow_weights", True)):
            from mdx.shadow import bf16_weights
            with bf16_weights(self.setting.raw_model.values()):
                return self._batch_process(inputs)
        return self._batch_process(inputs)

    def _batch_process(self, inputs):
        outputs = {}
        if self._pose_beside_depth():
            # the separate pose network does not depend on the depth network: it runs on a side stream beside it (forward here,
            # backward likewise -- autograd runs a node's backward on its forward's stream); many of both networks' kernels are
            # too small to fill the GPU on their own
            inputs = self.compute.stage_inputs(inputs)
            cur = torch.cuda.current_stream(self.device)
            if self._pose_stream is None:
                self._pose_stream = torch.cuda.Stream(self.device)
                if self.setting.sync is not None:
                    self.setting.sync.backward_streams([self._pose_stream])
            self._pose_stream.wait_stream(cur)
            with torch.cuda.stream(self._pose_stream):
                inputs, outputs = self.compute.forward_pose(inputs, outputs, self.setting)
            inputs, outputs = self.compute.forward_depth(inputs, outputs, self.setting)
            self.compute.loss_prologue(inputs, outputs)       # what the loss needs of the batch and the disparities alone
            cur.wait_stream(self._pose_stream)
        else:
            inputs, outputs = self.compute.forward_depth(inputs, outputs, self.setting)
            inputs, outputs = self.compute.forward_pose(inputs, outputs, self.setting)
        inputs, outputs = self.compute.image2warping(inputs, outputs, self.setting)
        outputs = self.compute.compute_loss(inputs, outputs, self.setting)
        return outputs

    def _step_gradients(self, inputs):
        """The step up to its gradients, gathered into the flat buffer and NOT exchanged (graph A of graphed_step's split form)."""
        outputs = self.batch_process(inputs)
        sync = self.setting.sync
        sync.zero()
        with sync.gather_only():
            outputs["loss"].backward()
            sync.finish()
        return outputs

    def _eager_step(self, inputs):
        """forward, backward, (data parallel: gradient all-reduce, overlapped with backward), Adam."""
        outputs = self.batch_process(inputs)
        sync = self.setting.sync
        if sync is None:
            self.setting.optim["optimizer"].zero_grad(set_to_none=True)
        else:
            sync.zero()                    # the gradients are views into one flat buffer: one memset
        outputs["loss"].backward()
        if sync is not None:
            sync.finish()
        self.setting.optim["optimizer"].step()
        return outputs

    def can_graph(self):
        """The step is capturable when nothing in it runs on the host: not with the reference's host-side noise
        (--noise cpu: torch.randn on the CPU + a copy from pageable memory), not with a non-RCCL process group.  With more than
        one rank the capture takes the split form (graphed_step) unless MDX_DP_GRAPH=1 asks for the exchange inside the graph."""
        sync = self.setting.sync
        return (str(self.device).startswith("cuda") and self.compute.noise_mode != "cpu"
                and (sync is None or sync.backend == "nccl"))

    def train_step(self, inputs):
        """opt.graph (GPU): the step -- under torch.distributed including the gradient exchange -- is captured once
        into a hipGraph and replayed (graphed_step); otherwise eager."""
        use_graph = getattr(self.opt, "graph", False) and self.can_graph()
        if not use_graph:
            if getattr(self.opt, "graph", False) and not getattr(self, "_told_eager", False) and str(self.device).startswith("cuda"):
                self._told_eager = True       # once, rank 0: --graph was asked for and the step runs eager (other bucket sizes, queues)
                if self.rank == 0:
                    why = "--noise cpu draws on the host" if self.compute.noise_mode == "cpu" else "the process group is not RCCL"
                    print("model_train: --graph requested but the step is not captured (%s); running the eager step" % why, flush=True)
            return self._eager_step(inputs)
        inputs = self.compute.prepare(inputs)     # decoded frames -> step entries (a no-op after the prefetcher)
        if self._graphed is None:
            self._graphed = graphed_step(self, inputs)
        mon = getattr(self.control, "_side", None)
        if mon is not None:                       # the depth monitor of the previous step reads the graph's outputs
            torch.cuda.current_stream(self.device).wait_stream(mon)
        return self._graphed(inputs)

    def train(self):
        names = self.control.metric_name
        epoch_train = {k: [] for k in names}
        epoch_valid = {k: [] for k in names}
        start = 0
        if getattr(self.opt, "resume", 0):     # weights, optimiser, scheduler and the per-epoch logs so far
            start = self.control.resume(self.setting, self.opt.resume, epoch_train, epoch_valid, compute=self.compute)
        for epoch in range(start, self.opt.epoch):
            batch_train = {k: [] for k in names}
            batch_valid = {k: [] for k in names}
            self.setting.set_train()
            sampler = getattr(self.setting.train_dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            for step, train_inputs in enumerate(self.batches(self.setting.train_dataloader)):
                train_outputs = self.train_step(train_inputs)
                batch_train = self.control.metric(train_inputs, train_outputs, batch_train)
                if self.opt.max_steps and step + 1 >= self.opt.max_steps:
                    break
            self.setting.set_valid()
            for step, valid_inputs in enumerate(self.batches(self.setting.valid_dataloader)):
                with torch.no_grad():
                    valid_outputs = self.batch_process(valid_inputs)
                    batch_valid = self.control.metric(valid_inputs, valid_outputs, batch_valid)
                if self.opt.max_steps and step + 1 >= self.opt.max_steps:
                    break
            self.setting.optim["scheduler"].step()
            if self._graphed is not None:         # StepLR wrote a new Python value: hand it to the captured Adam's lr tensor
                lr = self.setting.optim["scheduler"].get_last_lr()[0]
                self._graphed.set_lr(lr)
                for g in self.setting.optim["optimizer"].param_groups:
                    g["lr"] = self._graphed.lr
            # epoch means over everything the JOB saw: one all-reduce of the 2 x 8 scalars (every rank takes part)
            mean_train, mean_valid = self.control.epoch_means(batch_train), self.control.epoch_means(batch_valid)
            for key in names:
                epoch_train[key].append(mean_train[key])
                epoch_valid[key].append(mean_valid[key])
            if self.rank == 0:
                self.control.print(epoch, mean_train, mean_valid)
            self.control.save(epoch, epoch_train, epoch_valid, self.setting, compute=self.compute)


if __name__ == "__main__":
    trainer(options()).train()
