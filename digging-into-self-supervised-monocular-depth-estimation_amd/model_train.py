"""Training entry (reference: model_train.py:24-101): `trainer(options()).train()`, one process per GPU.

    python model_train.py --dataset synthetic --epoch 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 model_train.py ...
"""
import os
import sys

import numpy as np
import torch

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
    """One training step (networks, the photometric kernels launched through the C-ABI, fused Adam) captured into ONE
    hipGraph and replayed: ~1600 kernel launches leave the host's critical path, so the step stays GPU-bound while the
    host decodes and collates the next batches (measured: with 12 loader workers alive the eager step needs 24.5 ms of
    host time for 20.7 ms of GPU time).  Inputs live in static device buffers the batch is copied into; the outputs the
    loop reads afterwards (loss, depth of scale 0, auto-masks) are the graph's own tensors.  What Python does during a
    step and a replay would skip is re-applied per replay: the batch-norm step counters.  Single-process training only
    (under DDP the eager step runs); any change of the learning rate goes through set_lr (a device tensor the captured
    Adam reads)."""

    def __init__(self, tr, example, warmup=3):
        self.tr = tr
        dev = tr.device
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
        self.bns = [m for net in tr.setting.raw_model.values() for m in net.modules() if isinstance(m, BatchNorm2d)]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                tr._eager_step(dict(self.static))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        before = [m._pending_batches for m in self.bns]
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the DataLoader's pin-memory thread keeps allocating pinned host buffers for the next batches
        # while this thread captures; in the default "global" mode such a call from ANY thread invalidates the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = tr._eager_step(dict(self.static))
        self.bn_incr = [m._pending_batches - b for m, b in zip(self.bns, before)]
        torch.cuda.synchronize(dev)

    def set_lr(self, value):
        self.lr.fill_(float(value))

    def __call__(self, inputs):
        for k, v in inputs.items():
            if k in self.copied and torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
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
            self.device = "cuda:%d" % local
            torch.cuda.set_device(local)
        else:
            self.device = "cpu"
        if world > 1 and not torch.distributed.is_initialized():
            torch.distributed.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
        self.rank = int(os.environ.get("RANK", "0"))
        from mdx.tuning import install_miopen_db
        install_miopen_db(self.rank)          # tuned conv / batch-norm solvers for the default shapes (immediate mode)
        torch.backends.cudnn.benchmark = bool(getattr(opt, "miopen_find", False))
        self.setting = setting(opt, self.device)
        self.compute = compute(opt, self.device)
        self.control = control(opt, self.device)

    def batches(self, loader):
        """The loader's batches, uploaded one step ahead on a side stream (GPU) or as they are (CPU)."""
        if str(self.device).startswith("cuda") and getattr(self.opt, "device_prefetch", True):
            return device_prefetcher(loader, self.device, self.compute._step_reads, self.compute.prepare)
        return loader

    def batch_process(self, inputs):
        outputs = {}
        inputs, outputs = self.compute.forward_depth(inputs, outputs, self.setting)
        inputs, outputs = self.compute.forward_pose(inputs, outputs, self.setting)
        inputs, outputs = self.compute.image2warping(inputs, outputs, self.setting)
        outputs = self.compute.compute_loss(inputs, outputs, self.setting)
        return outputs

    def _eager_step(self, inputs):
        outputs = self.batch_process(inputs)
        self.setting.optim["optimizer"].zero_grad(set_to_none=True)
        outputs["loss"].backward()
        self.setting.optim["optimizer"].step()
        return outputs

    def train_step(self, inputs):
        """opt.graph (single process, GPU): the step is captured once into a hipGraph and replayed (graphed_step);
        otherwise eager."""
        use_graph = (getattr(self.opt, "graph", False) and str(self.device).startswith("cuda")
                     and not self.setting.distributed)
        if not use_graph:
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
        start = self.control.resume(self.setting, self.opt.resume) if getattr(self.opt, "resume", 0) else 0
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
            for key in names:
                epoch_train[key].append(self.control._mean(batch_train[key]))
                epoch_valid[key].append(self.control._mean(batch_valid[key]))
            if self.rank == 0:
                self.control.print(epoch, batch_train, batch_valid)
            self.control.save(epoch, epoch_train, epoch_valid, self.setting)


if __name__ == "__main__":
    trainer(options()).train()
