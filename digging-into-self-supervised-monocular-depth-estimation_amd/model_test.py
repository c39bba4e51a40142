"""Eigen-split evaluation (reference: model_test.py:29-159), cv2-free.

    load_weights     model_test.py:29-43     inference   model_test.py:61-119
Protocol: depth network on ("color", 0, 0), disparity2depth(disp, 1e-3, 80) (HIP kernel), bilinear resize to
the ground-truth size, Garg crop, per-image median scaling, clamp to [1e-3, 80], seven metrics, mean over images.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from model_layer import ResnetEncoder, DepthDecoder, disparity2depth   # noqa: E402
from model_loss import compute_depth_error                             # noqa: E402
from model_utility import readlines                                    # noqa: E402

METRICS = ["abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"]


def load_weights(encoder, decoder, encoder_path, decoder_path, device):
    """reference model_test.py:29-43: extra keys of the checkpoint (e.g. height/width) are filtered out."""
    enc_state = torch.load(encoder_path, map_location=device)
    encoder.load_state_dict({k: v for k, v in enc_state.items() if k in encoder.state_dict()})
    decoder.load_state_dict(torch.load(decoder_path, map_location=device))
    return encoder, decoder


def garg_crop_mask(gt):
    """reference model_test.py:100-103: rows 0.408..0.992, cols 0.036..0.964 of the ground-truth image."""
    gh, gw = gt.shape[-2:]
    crop = torch.zeros_like(gt, dtype=torch.bool)
    crop[..., int(0.40810811 * gh):int(0.99189189 * gh), int(0.03594771 * gw):int(0.96405229 * gw)] = True
    return crop


def evaluate_batch(disp, gt, min_depth=1e-3, max_depth=80.0):
    """disp [B,1,h,w] (sigmoid output), gt [B,1,H,W] (0 = no return) -> list of per-image metric tuples."""
    _, depth = disparity2depth(disp, min_depth, max_depth)
    depth = TF.interpolate(depth, gt.shape[-2:], mode="bilinear", align_corners=False)   # cv2.resize INTER_LINEAR
    out = []
    for b in range(gt.shape[0]):
        mask = (gt[b] > min_depth) & (gt[b] < max_depth) & garg_crop_mask(gt[b])
        g, p = gt[b][mask], depth[b][mask]
        if g.numel() == 0:
            continue
        p = p * (torch.median(g) / torch.median(p))
        p = torch.clamp(p, min_depth, max_depth)
        out.append(tuple(float(v) for v in compute_depth_error(g, p, "torch")))
    return out


def inference(opt, dataset=None, encoder=None, decoder=None, weights=None):
    device = "cuda:0"
    if dataset is None:
        from model_loader import KITTIMonoDataset_v2
        names = readlines(os.path.join(opt.splits, opt.datatype, "test_files.txt"))
        dataset = KITTIMonoDataset_v2(opt.datapath, names, False, [0], opt.height, opt.width, ".jpg", 4)
    encoder = encoder or ResnetEncoder(opt.num_layers, False)
    decoder = decoder or DepthDecoder(encoder.num_ch_enc)
    if weights:
        load_weights(encoder, decoder, weights[0], weights[1], "cpu")
    encoder, decoder = encoder.to(device).eval(), decoder.to(device).eval()
    loader = torch.utils.data.DataLoader(dataset, batch_size=getattr(opt, "batch", 16), shuffle=False, num_workers=0)
    rows = []
    with torch.no_grad():
        for batch in loader:
            disp = decoder(encoder(batch[("color", 0, 0)].to(device)))[("disp", 0)]
            rows += evaluate_batch(disp, batch[("depth", 0)].to(device))
    mean = np.mean(np.array(rows), axis=0)
    return dict(zip(METRICS, mean.tolist()))


if __name__ == "__main__":
    from model_option import options
    o = options()
    save = os.path.join("./model_save", o.save)
    w = (os.path.join(save, "encoder%d.pt" % o.epoch), os.path.join(save, "decoder%d.pt" % o.epoch))
    res = inference(o, weights=w if os.path.isfile(w[0]) else None)
    print(("{:>9}" * 7).format(*METRICS))
    print(("{:9.3f}" * 7).format(*[res[k] for k in METRICS]))
