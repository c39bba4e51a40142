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

    def __init__(self, loader, device, wanted):
        self.loader, self.device, self.wanted = loader, device, wanted
        self.stream = torch.cuda.Stream(device) if str(device).startswith("cuda") else None

    def _upload(self, batch):
        if self.stream is None or batch is None:
            return batch
        with torch.cuda.stream(self.stream):
            return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) and self.wanted(k) else v)
                    for k, v in batch.items()}

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


class trainer(object):
    def __init__(self, opt):
        self.opt = opt
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
            return device_prefetcher(loader, self.device, self.compute._step_reads)
        return loader

    def batch_process(self, inputs):
        outputs = {}
        inputs, outputs = self.compute.forward_depth(inputs, outputs, self.setting)
        inputs, outputs = self.compute.forward_pose(inputs, outputs, self.setting)
        inputs, outputs = self.compute.image2warping(inputs, outputs, self.setting)
        outputs = self.compute.compute_loss(inputs, outputs, self.setting)
        return outputs

    def train_step(self, inputs):
        outputs = self.batch_process(inputs)
        self.setting.optim["optimizer"].zero_grad(set_to_none=True)
        outputs["loss"].backward()
        self.setting.optim["optimizer"].step()
        return outputs

    def train(self):
        names = self.control.metric_name
        epoch_train = {k: [] for k in names}
        epoch_valid = {k: [] for k in names}
        for epoch in range(self.opt.epoch):
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
            for key in names:
                epoch_train[key].append(self.control._mean(batch_train[key]))
                epoch_valid[key].append(self.control._mean(batch_valid[key]))
            if self.rank == 0:
                self.control.print(epoch, batch_train, batch_valid)
            self.control.save(epoch, epoch_train, epoch_valid, self.setting)


if __name__ == "__main__":
    trainer(options()).train()
