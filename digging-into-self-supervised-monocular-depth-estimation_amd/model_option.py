"""Command-line options (reference: model_option.py:5-89): same flag names and defaults; list-typed flags
are parsed properly, and the multi-GPU / precision flags the reference lacks are added."""
import argparse


def _ids(text):
    return [t if t == "s" else int(t) for t in str(text).replace(",", " ").split()]


def options(argv=None):
    p = argparse.ArgumentParser(description="MI355X self-supervised depth training")
    p.add_argument("--datapath", type=str, default="./dataset/kitti")
    p.add_argument("--splits", type=str, default="./splits")
    p.add_argument("--dataset", type=str, default="synthetic", choices=["kitti_mono", "kitti_stereo", "synthetic"])
    p.add_argument("--datatype", type=str, default="kitti_eigen_zhou")
    p.add_argument("--epoch", type=int, default=24)
    p.add_argument("--batch", type=int, default=12)
    p.add_argument("--prepetch", type=int, default=2)
    p.add_argument("--num_workers", type=int, default=None,
                   help="DataLoader workers per process.  Default: 12 (the reference's, model_option.py:32-34) for fp32 networks, 24 "
                        "with --amp bf16, whose step consumes ~1500 samples/s (12 workers deliver ~700-860 with --gpu_image_prep, 16 "
                        "~1100, 24 ~1200 on a 16-core share: tools/loader_cost.py --workers N); -1 = sized from this rank's share of "
                        "the host cores")
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--scheduler_step", type=int, default=15)
    p.add_argument("--disp_smoothness", type=float, default=1e-3)
    p.add_argument("--save", type=str, default="mono")
    p.add_argument("--height", type=int, default=192)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--scales", type=_ids, default=[0, 1, 2, 3])
    p.add_argument("--min_depth", type=float, default=0.1)
    p.add_argument("--max_depth", type=float, default=100.0)
    p.add_argument("--frame_ids", type=_ids, default=[0, -1, 1])
    p.add_argument("--pose_frames", type=str, default="pair", choices=["pair", "all"])
    p.add_argument("--num_layers", type=int, default=18, choices=[18, 34, 50, 101, 152])
    p.add_argument("--weight_init", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--pose_type", type=str, default="separate", choices=["posecnn", "shared", "separate"])
    p.add_argument("--use_automasking", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    # additions
    p.add_argument("--fused", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--fused_train", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True,
                   help="one launch for every scale's photometric term and its gradient (csrc/photo_train.hip)")
    p.add_argument("--uint8_loader", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True,
                   help="colours stay uint8 through the DataLoader; x/255 happens on the GPU (same values, 1/4 of the bytes)")
    p.add_argument("--collate_step_keys", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True,
                   help="DataLoader workers stack only the entries a step reads")
    p.add_argument("--graph", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True,
                   help="single-GPU training: capture the step into one hipGraph and replay it (host launch cost off the critical path)")
    p.add_argument("--device_prefetch", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True,
                   help="upload the next batch on a side stream while the current step computes")
    p.add_argument("--gpu_image_prep", type=str, default="auto", choices=["auto", "true", "false"],
                   help="KITTI loaders: workers only decode; flip, Lanczos pyramid, colour jitter and ToTensor run on the GPU "
                        "(bit-equal to Pillow; csrc/imgproc.hip).  auto = on when training on a GPU")
    p.add_argument("--synthetic_raw", action="store_true",
                   help="synthetic dataset hands over KITTI-sized decoded frames (1242x375 uint8), as the KITTI loaders do with gpu_image_prep")
    p.add_argument("--resume", type=int, default=0,
                   help="restart after this many finished epochs from ./model_save/<save>/ (weights <key><N>.pt + state<N>.pt)")
    p.add_argument("--native_adam", type=int, default=1,
                   help="GPU: torch.optim.Adam(fused=True) with its step as ONE launch (mdx/optim.py, csrc/adam.hip); 0: torch's own step")
    p.add_argument("--shadow_weights", type=int, default=1,
                   help="--amp bf16: cast every convolution weight once per step by one launch (mdx/shadow.py); 0: autocast's cast per convolution")
    p.add_argument("--fused_tail", type=int, default=1,
                   help="GPU: the loss as one autograd node, the pose head's output -> projections in one launch (0: the reference's small torch ops)")
    p.add_argument("--overlap_pose", type=int, default=1,
                   help="1: the separate pose network runs on a side stream beside the depth network, forward and backward (many of "
                        "their kernels are too small to fill the GPU alone: +13 %% fp32, +32 %% bf16 on one MI355X); 0: one after the other")
    p.add_argument("--synthetic_geometry", action="store_true",
                   help="synthetic dataset: frames rendered from one rigid textured scene at known poses, ground truth = its "
                        "depth (model_tool/synthetic.py: scene) -- for tests of what the step learns")
    p.add_argument("--synthetic_pool", type=int, default=0, help="synthetic dataset: number of distinct samples kept (0 = all)")
    p.add_argument("--noise", type=str, default="device", choices=["device", "cpu"])
    p.add_argument("--grad_comm", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="data parallel: dtype of the gradient all-reduce (bf16 halves the bytes over xGMI; gradients rounded once)")
    p.add_argument("--bucket_mb", type=int, default=0,
                   help="data parallel: size of an all-reduce bucket (0 = one bucket for a captured step, 32 MB eager)")
    p.add_argument("--amp", type=str, default="none", choices=["none", "bf16"])
    p.add_argument("--channels_last", type=str, nargs="?", const="all", default="auto",
                   help='stages whose maps are channels-last (NHWC): "none", "all", "auto" (= mdx.layout.DEFAULT_PLAN, the '
                        'measured fastest) or a list of stem,layer1..layer4,decoder,pose (mdx/layout.py)')
    p.add_argument("--synthetic_length", type=int, default=768)
    p.add_argument("--max_steps", type=int, default=0, help="stop an epoch early (0 = full epoch)")
    p.add_argument("--miopen_find", action="store_true",
                   help="cudnn.benchmark: MIOpen find mode (a search of minutes for shapes that are not in the "
                        "shipped find-db; the default, immediate mode, already uses the db)")
    o = p.parse_args(argv)
    if o.num_workers is None:
        o.num_workers = 24 if o.amp == "bf16" else 12
    return o
