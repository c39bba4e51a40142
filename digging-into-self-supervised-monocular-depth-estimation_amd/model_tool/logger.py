"""`control`: metrics, printing and checkpoints (reference: model_tool/logger.py:25-72).

Same metric names and file naming (`./model_save/<save>/<key><epoch>.pt`, loss/<metric>.npy).  Unlike the
reference, per-batch values stay on the device and are synchronised once per epoch (the reference issues
nine device->host copies per step: model_train.py:69, logger.py:31-35)."""
import os

import numpy as np
import torch

from model_loss import compute_depth_metric


class control(object):
    def __init__(self, opt, device):
        self.opt = opt
        self.device = device
        self.metric_name = ["loss", "abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"]
        # The train-time depth monitor (csrc/monitor.hip, ~0.16 ms of kernels) runs on the step's own stream.  Running it
        # on a side stream beside the next step's convolutions was measured and is WORSE: the cross-stream dependencies
        # (the monitor waits for the step, the next graph replay waits for the monitor) cost ~1 ms per step
        # (tools/loop_bisect.py: +1.10 ms on a side stream, +0.27 ms on the main stream).  opt.metric_side_stream keeps
        # the experiment reachable.
        self._side = None
        if str(device).startswith("cuda") and getattr(opt, "metric_side_stream", False):
            self._side = torch.cuda.Stream(device)

    def metric(self, inputs, outputs, metric_dict):
        metric_dict["loss"].append(outputs["loss"].detach().clone())     # a graph-replayed step returns the SAME tensor every step
        if ("depth", 0) in inputs and ("depth", 0, 0) in outputs:
            gt, depth = inputs[("depth", 0)], outputs[("depth", 0, 0)]
            if self._side is not None and gt.is_cuda and depth.is_cuda:
                main = torch.cuda.current_stream(gt.device)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    depth_errors = compute_depth_metric(inputs, outputs, "torch")
                for t in (gt, depth):
                    t.record_stream(self._side)          # their memory must not be reused before the monitor has read it
            else:
                depth_errors = compute_depth_metric(inputs, outputs, "torch")
            for index, metric in enumerate(self.metric_name[1:]):
                metric_dict[metric].append(depth_errors[index].detach())
        return metric_dict

    @staticmethod
    def _mean(values):
        if not len(values):
            return float("nan")
        if torch.is_tensor(values[0]):
            if values[0].is_cuda:
                torch.cuda.synchronize(values[0].device)    # the monitor's side stream included
            return float(torch.stack([v.float().reshape(()) for v in values]).mean().cpu())
        return float(np.mean(values))

    def epoch_means(self, log):
        """{metric: mean over the epoch} -- over every rank's batches when the run is data-parallel (the reference's
        logger prints the mean of everything the run saw, logger.py:30-48): ONE all-reduce of the 8 scalars, which every
        rank must enter."""
        from .parallel import mean_over_ranks
        local = [self._mean(log[key]) for key in self.metric_name]
        return dict(zip(self.metric_name, mean_over_ranks(local, self.device)))

    def print(self, epoch, train_log, valid_log):
        """logs: {metric: per-batch values} (reference signature, logger.py:37-48) or {metric: epoch mean}."""
        print("EPOCH   {0}".format(epoch + 1))
        for name, log in (("Train Log", train_log), ("Valid Log", valid_log)):
            print(name, end=" ")
            for key in self.metric_name:
                v = log[key]
                print("  {} {:0.3f}".format(key, self._mean(v) if isinstance(v, (list, tuple)) else float(v)), end=" ")
            print(" ")

    def save(self, epoch, train_log, valid_log, setting, compute=None):
        if int(os.environ.get("RANK", "0")) != 0:
            return
        save_directory = os.path.join("./model_save", self.opt.save)
        loss_directory = os.path.join(save_directory, "loss")
        os.makedirs(loss_directory, exist_ok=True)
        models = getattr(setting, "raw_model", setting.model)
        if (epoch + 1) % 2 == 0 or (epoch + 1) == self.opt.epoch:
            for key in models:
                torch.save(models[key].state_dict(), os.path.join(save_directory, key + str(epoch + 1) + ".pt"))
            # what a restart needs beyond the reference's files (it saves weights only, logger.py:51-68)
            torch.save({"epoch": epoch + 1, "optimizer": setting.optim["optimizer"].state_dict(),
                        "scheduler": setting.optim["scheduler"].state_dict(),
                        # steps served by the in-kernel noise generator (compute.noise_rng): a resumed run goes on in the
                        # stream instead of replaying the draws of epoch 0
                        "noise_offset": compute.noise_offset() if compute is not None else None,
                        "train_log": {k: list(v) for k, v in train_log.items()},
                        "valid_log": {k: list(v) for k, v in valid_log.items()}},
                       os.path.join(save_directory, "state" + str(epoch + 1) + ".pt"))
        if (epoch + 1) == self.opt.epoch:
            for key in self.metric_name:
                np.save(os.path.join(loss_directory, key + ".npy"), np.asarray(valid_log[key]))
                np.save(os.path.join(loss_directory, "train_" + key + ".npy"), np.asarray(train_log[key]))

    def resume(self, setting, epoch, train_log=None, valid_log=None, compute=None):
        """Restart from the files `save` wrote after `epoch` epochs (every rank loads them): network weights from the
        reference-named `<key><epoch>.pt`, optimiser, scheduler and the per-epoch logs so far (so that the loss/*.npy
        files of the finished run cover every epoch) from `state<epoch>.pt`.  `epoch` must be one `save` wrote a
        checkpoint for: the even ones and the last (the reference's schedule, logger.py:60).  -> the epoch to go on with."""
        save_directory = os.path.join("./model_save", self.opt.save)
        models = getattr(setting, "raw_model", setting.model)
        for key in models:
            path = os.path.join(save_directory, key + str(epoch) + ".pt")
            if not os.path.exists(path):
                raise FileNotFoundError("cannot resume from epoch %d: %s is missing (checkpoints are written after even "
                                        "epochs and after the last one)" % (epoch, path))
            models[key].load_state_dict(torch.load(path, map_location=self.device))
        state = torch.load(os.path.join(save_directory, "state" + str(epoch) + ".pt"), map_location=self.device)
        setting.optim["optimizer"].load_state_dict(state["optimizer"])
        setting.optim["scheduler"].load_state_dict(state["scheduler"])
        if compute is not None and state.get("noise_offset") is not None:
            compute.set_noise_offset(state["noise_offset"])
        for mine, key in ((train_log, "train_log"), (valid_log, "valid_log")):
            if mine is not None and key in state:
                for k in mine:
                    mine[k][:0] = list(state[key].get(k, []))
        return int(state["epoch"])
