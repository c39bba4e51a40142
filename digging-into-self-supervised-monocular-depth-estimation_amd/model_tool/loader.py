"""`setting`: builds loaders, networks, projection modules, losses and the optimiser
(reference: model_tool/loader.py:16-119), plus what the reference never had: one process per GPU, a rank-sharded
sampler and the gradient all-reduce over RCCL (torch.distributed backend "nccl" on ROCm; model_tool/parallel.py).
"""
import os

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from model_layer import *  # noqa: F401,F403
from model_loss import *   # noqa: F401,F403
from .synthetic import SyntheticKITTI


def _opt(opt, name, default):
    return getattr(opt, name, default)


def collate_step_keys(samples):
    """default_collate over the entries a step reads only (model_tool.processor.step_reads): the workers do not stack,
    and the pinning thread does not copy, what nothing consumes."""
    from torch.utils.data import default_collate
    from .processor import step_reads
    return default_collate([{k: v for k, v in s.items() if step_reads(k)} for s in samples])


def collate_raw_step_keys(samples):
    """gpu_image_prep: the decoded frames padded into one block per frame id (model_loader.kitti.collate_raw)."""
    from model_loader.kitti import collate_raw
    from .processor import step_reads
    return collate_raw(samples, step_reads)


def gpu_image_prep(opt, device):
    """opt.gpu_image_prep: "auto" = on when the step runs on a GPU (the kernels have no CPU counterpart)."""
    v = str(_opt(opt, "gpu_image_prep", "auto")).lower()
    return str(device).startswith("cuda") if v == "auto" else v in ("1", "true", "yes")


class setting(object):
    def __init__(self, opt, device):
        self.opt = opt
        self.device = device
        self.num_pose_frames = len(opt.frame_ids) if opt.pose_frames == "all" else 2
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        # a process group makes the run data-parallel -- also a group of ONE rank (the full exchange path on one GPU)
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        if self.distributed:
            self.world_size, self.rank = torch.distributed.get_world_size(), torch.distributed.get_rank()

        self.train_dataloader = self.set_loader("train", True, True)
        self.valid_dataloader = self.set_loader("val", False, False)
        self.model = {}
        self.sync = None
        self.parameters = []
        self.set_model()
        self.loss = {}
        self.set_loss()
        self.optim = {}
        self.set_optim()

    # reference: loader.py:50-66
    def set_loader(self, split, is_training, shuffle):
        opt = self.opt
        if opt.dataset == "synthetic":
            length = _opt(opt, "synthetic_length", 64 * opt.batch) if is_training else 4 * opt.batch
            dataset = SyntheticKITTI(length, opt.frame_ids, opt.height, opt.width, len(opt.scales),
                                     seed=0 if is_training else 1, pool=_opt(opt, "synthetic_pool", 0),
                                     uint8=_opt(opt, "uint8_loader", False),
                                     raw=gpu_image_prep(opt, self.device) and _opt(opt, "synthetic_raw", False),
                                     is_training=is_training, geometry=_opt(opt, "synthetic_geometry", False))
        else:
            from model_loader import KITTIMonoDataset_v2, KITTIMonoStereoDataset
            from model_utility import readlines
            names = readlines(os.path.join(opt.splits, opt.datatype, "{}_files.txt".format(split)))
            cls = KITTIMonoDataset_v2 if opt.dataset == "kitti_mono" else KITTIMonoStereoDataset
            dataset = cls(opt.datapath, names, is_training, opt.frame_ids, opt.height, opt.width, ".jpg", len(opt.scales))
            dataset.uint8 = _opt(opt, "uint8_loader", False)
            dataset.gpu_prep = gpu_image_prep(opt, self.device)
        sampler = None
        if self.distributed:
            sampler = DistributedSampler(dataset, self.world_size, self.rank, shuffle=shuffle, drop_last=True)
            shuffle = False
        workers = opt.num_workers
        if workers < 0:
            # -1 = sized for this rank: at most what the step consumes -- with the GPU image preparation 12 workers deliver ~700-860
            # samples/s, 16 ~1100, 24 ~1200 on a 16-core share (tools/loader_cost.py --workers N; the decoders wait on I/O, a few
            # more workers than cores pay); an fp32 step takes ~750 samples/s, a bf16 step ~1500 (round 5)
            share = max(2, len(os.sched_getaffinity(0)) // max(1, min(self.world_size, torch.cuda.device_count() or 1)))
            workers = max(2, min(24, share + share // 2) if str(_opt(opt, "amp", "none")) == "bf16" else min(16, share))
        return DataLoader(dataset, opt.batch, shuffle, sampler=sampler, num_workers=workers,
                          drop_last=True, pin_memory=str(self.device).startswith("cuda"),
                          collate_fn=(collate_raw_step_keys if getattr(dataset, "gpu_prep", False) or getattr(dataset, "raw", False)
                                      else collate_step_keys if _opt(opt, "collate_step_keys", False) else None),
                          persistent_workers=workers > 0, prefetch_factor=(_opt(opt, "prefetch_factor", 4) if workers > 0 else None))

    # reference: loader.py:70-96
    def set_model(self):
        opt = self.opt
        self.model["encoder"] = ResnetEncoder(num_layers=opt.num_layers, pretrained=opt.weight_init)
        self.model["decoder"] = DepthDecoder(num_ch_enc=self.model["encoder"].num_ch_enc, scales=opt.scales)
        if opt.pose_type == "posecnn":
            self.model["pose_decoder"] = PoseCNN(num_input_frames=self.num_pose_frames)
        elif opt.pose_type == "shared":
            self.model["pose_decoder"] = PoseDecoder(self.model["encoder"].num_ch_enc, self.num_pose_frames)
        elif opt.pose_type == "separate":
            self.model["pose_encoder"] = ResnetEncoder(opt.num_layers, opt.weight_init, self.num_pose_frames)
            self.model["pose_decoder"] = PoseDecoder(self.model["pose_encoder"].num_ch_enc, num_input_features=1,
                                                     num_frames_to_predict_for=2)
        self.inv_projection = {0: Depth2PointCloud(opt.batch, opt.height, opt.width).to(self.device)}
        self.for_projection = {0: PointCloud2Pixel(opt.batch, opt.height, opt.width).to(self.device)}
        for key in self.model:
            self.model[key] = self.model[key].to(self.device)
        # which stages keep channels-last maps (mdx.layout: "none", "all", "auto" or a list like "stem,layer1,decoder")
        from mdx.layout import apply_plan
        # ("auto" is a GPU plan: the CPU run of the reference-style device pick keeps the reference's planar modules)
        plan = _opt(opt, "channels_last", "auto")
        if plan == "auto" and not str(self.device).startswith("cuda"):
            plan = "none"
        self.channels_last_stages = apply_plan(self.model, plan)
        for key in self.model:
            m = self.model[key]
            # the ResNet classifier head never receives a gradient (the reference keeps it in the optimiser
            # list, loader.py:93-95, where Adam skips it); frozen here so DDP needs no unused-parameter scan
            for name, p in m.named_parameters():
                if name.startswith("encoder.fc."):
                    p.requires_grad_(False)
            self.model[key] = m
            self.parameters += [p for p in m.parameters() if p.requires_grad]
        self.raw_model = self.model
        if self.distributed:
            # one flat gradient buffer + bucketed all-reduce(mean) issued from inside backward (model_tool/parallel.py):
            # capturable together with the rest of the step; buffers (batch-norm statistics) stay per GPU as in the
            # single-device reference, parameters start from rank 0's
            from .parallel import grad_sync, broadcast_state, dp_graph_allowed
            broadcast_state(self.model.values())
            comm = {"fp32": None, "bf16": torch.bfloat16}[str(_opt(opt, "grad_comm", "fp32"))]
            # a captured step exchanges ONE bucket when backward ends; an eager one overlaps 32 MB buckets with backward
            # (measured: model_tool/parallel.py)
            captured = bool(_opt(opt, "graph", False)) and str(self.device).startswith("cuda") \
                and str(_opt(opt, "noise", "device")) != "cpu" and torch.distributed.get_backend() == "nccl"       # = trainer.can_graph()
            mb = int(_opt(opt, "bucket_mb", 0)) or ((1 << 20) if captured else 32)
            self.sync = grad_sync(self.parameters, bucket_mb=mb, comm_dtype=comm)

    # reference: loader.py:99-103
    def set_loss(self):
        self.loss["reprojection"] = ReprojectionLoss().to(self.device)
        self.loss["edge_aware"] = SmoothLoss().to(self.device)

    # reference: loader.py:106-109
    def set_optim(self):
        if str(self.device).startswith("cuda") and _opt(self.opt, "native_adam", True):
            from mdx.optim import Adam             # torch.optim.Adam(fused=True) with its step as one launch (csrc/adam.hip)
            self.optim["optimizer"] = Adam(self.parameters, float(self.opt.learning_rate))
        else:
            fused = str(self.device).startswith("cuda")
            self.optim["optimizer"] = torch.optim.Adam(self.parameters, float(self.opt.learning_rate), fused=fused)
        self.optim["scheduler"] = torch.optim.lr_scheduler.StepLR(self.optim["optimizer"], self.opt.scheduler_step)

    def set_train(self):
        for value in self.model.values():
            value.train()

    def set_valid(self):
        for value in self.model.values():
            value.eval()
