"""Eigen-split evaluation (reference: model_test.py:29-159), cv2-free.

    load_weights        model_test.py:29-43      load_ground_truth   model_test.py:47-57
    inference           model_test.py:61-119     evaluate_image      model_test.py:89-112 (the per-image block)

Protocol, step for step as the reference: depth network on ("color", 0, 0); scaled disparity =
disparity2depth(disp, 1e-3, 80)[0] (HIP kernel); per image: resize the SCALED DISPARITY to the ground truth's own
size with cv2.resize's default bilinear rule (resize_bilinear below), THEN depth = 1 / disparity; ground truth =
point2depth(calibration, scan, cam 2, vel_depth=True) at native size; mask gt in (1e-3, 80) and the Garg crop
rows 153:371, cols 44:1197 (eigen splits); per-image median scaling; clamp to [1e-3, 80]; seven metrics; mean over images.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from model_layer import ResnetEncoder, DepthDecoder, disparity2depth   # noqa: E402
from model_loss import compute_depth_error                             # noqa: E402
from model_utility import readlines, point2depth                       # noqa: E402

METRICS = ["abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"]
MIN_DEPTH, MAX_DEPTH = 1e-3, 80.0
GARG_CROP = (153, 371, 44, 1197)       # model_test.py:101, fixed numbers whatever the ground-truth size


def load_weights(encoder, decoder, encoder_path, decoder_path, device):
    """reference model_test.py:29-43: extra keys of the checkpoint (e.g. height/width) are filtered out."""
    enc_state = torch.load(encoder_path, map_location=device)
    encoder.load_state_dict({k: v for k, v in enc_state.items() if k in encoder.state_dict()})
    decoder.load_state_dict(torch.load(decoder_path, map_location=device))
    return encoder, decoder


def load_ground_truth(datapath, lines):
    """reference model_test.py:47-57: velodyne depth of camera 2 at the image's native size, one map per test line."""
    out = []
    for line in lines:
        folder, frame_id, _ = line.split()
        calib = os.path.join(datapath, folder.split("/")[0])
        velo = os.path.join(datapath, folder, "velodyne_points/data", "{:010d}.bin".format(int(frame_id)))
        out.append(point2depth(calib, velo, 2, True).astype(np.float32))
    return out


def resize_bilinear(img, width, height):
    """cv2.resize(img, (width, height)) for a float32 single-channel image (INTER_LINEAR, the default the reference
    uses at model_test.py:95): source coordinate (dst + 0.5) * scale - 0.5, taps clamped to the image (replicated
    border), float32 arithmetic."""
    img = np.asarray(img, np.float32)
    h, w = img.shape

    def taps(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        low, high = i0 < 0, i0 >= n_in - 1
        frac[low | high] = 0
        i0 = np.clip(i0, 0, n_in - 1)
        return i0, np.minimum(i0 + 1, n_in - 1), frac
    y0, y1, fy = taps(height, h)
    x0, x1, fx = taps(width, w)
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return (top * (1 - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


def evaluate_image(pred_disparity, ground_truth, eigen=True):
    """reference model_test.py:89-112 for ONE image.  pred_disparity [h,w]: scaled disparity; ground_truth [H,W]
    (0 = no return).  Returns the seven metrics, or None when the mask is empty."""
    height, width = ground_truth.shape
    pred_depth = 1 / resize_bilinear(pred_disparity, width, height)
    if eigen:
        mask = np.logical_and(ground_truth > MIN_DEPTH, ground_truth < MAX_DEPTH)
        crop_mask = np.zeros(mask.shape)
        crop_mask[GARG_CROP[0]:GARG_CROP[1], GARG_CROP[2]:GARG_CROP[3]] = 1
        mask = np.logical_and(mask, crop_mask)
    else:
        mask = ground_truth > 0.
    if not mask.any():
        return None
    pred_depth, gt = pred_depth[mask], ground_truth[mask]
    pred_depth = pred_depth * (np.median(gt) / np.median(pred_depth))
    pred_depth[pred_depth < MIN_DEPTH] = MIN_DEPTH
    pred_depth[pred_depth > MAX_DEPTH] = MAX_DEPTH
    return compute_depth_error(gt, pred_depth, "numpy")


def inference(opt, dataset=None, lines=None, ground_truth=None, encoder=None, decoder=None, weights=None, device=None):
    """Mean metrics over the test split.  `dataset` yields ("color", 0, 0); `ground_truth`: list of native-size maps
    (built with load_ground_truth from `lines` when absent)."""
    if device is None:
        device = "cuda:%d" % torch.cuda.current_device()
    if lines is None:
        lines = readlines(os.path.join(opt.splits, opt.datatype, "test_files.txt"))
    if dataset is None:
        from model_loader import KITTIMonoDataset_v2
        dataset = KITTIMonoDataset_v2(opt.datapath, lines, False, [0], opt.height, opt.width, ".jpg", 4)
        dataset.load_depth = False          # ground truth comes from load_ground_truth, as in the reference
    if ground_truth is None:
        ground_truth = load_ground_truth(opt.datapath, lines)
    encoder = encoder or ResnetEncoder(opt.num_layers, False)
    decoder = decoder or DepthDecoder(encoder.num_ch_enc)
    if weights:
        load_weights(encoder, decoder, weights[0], weights[1], "cpu")
    with torch.cuda.device(device):
        encoder, decoder = encoder.to(device).eval(), decoder.to(device).eval()
        loader = torch.utils.data.DataLoader(dataset, batch_size=getattr(opt, "batch", 16), shuffle=False, num_workers=0)
        disparities = []
        with torch.no_grad():
            for batch in loader:
                disp = decoder(encoder(batch[("color", 0, 0)].to(device)))[("disp", 0)]
                scaled, _ = disparity2depth(disp, MIN_DEPTH, MAX_DEPTH)        # model_test.py:81
                disparities.append(scaled.cpu()[:, 0].numpy())
    disparities = np.concatenate(disparities)
    eigen = getattr(opt, "datatype", "kitti_eigen_zhou") in ("kitti_eigen_zhou", "kitti_eigen_full")
    rows = [r for r in (evaluate_image(disparities[i], ground_truth[i], eigen) for i in range(len(disparities)))
            if r is not None]
    mean = np.mean(np.array(rows, dtype=np.float64), axis=0)
    return dict(zip(METRICS, mean.tolist()))


if __name__ == "__main__":
    from model_option import options
    o = options()
    save = os.path.join("./model_save", o.save)
    w = (os.path.join(save, "encoder%d.pt" % o.epoch), os.path.join(save, "decoder%d.pt" % o.epoch))
    res = inference(o, weights=w if os.path.isfile(w[0]) else None)
    print(("{:>9}" * 7).format(*METRICS))
    print(("{:9.3f}" * 7).format(*[res[k] for k in METRICS]))
