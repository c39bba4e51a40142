"""Depth metrics (reference: model_loss/model_metric.py:19-105), cv2-free restatement."""
import numpy as np
import torch
import torch.nn.functional as TF


def compute_depth_error(ground_truth, prediction, lib="numpy"):
    """reference: model_metric.py:19-66 -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    if lib == "numpy":
        thresh = np.maximum((ground_truth / prediction), (prediction / ground_truth))
        a1 = (thresh < 1.25).mean()
        a2 = (thresh < 1.25 ** 2).mean()
        a3 = (thresh < 1.25 ** 3).mean()
        rmse = np.sqrt(((ground_truth - prediction) ** 2).mean())
        rmse_log = np.sqrt(((np.log(ground_truth) - np.log(prediction)) ** 2).mean())
        abs_rel = np.mean(np.abs(ground_truth - prediction) / ground_truth)
        sq_rel = np.mean(((ground_truth - prediction) ** 2) / ground_truth)
        return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
    if lib == "torch":
        threshold = torch.maximum((ground_truth / prediction), (prediction / ground_truth))
        a1 = (threshold < 1.25).float().mean()
        a2 = (threshold < 1.25 ** 2).float().mean()
        a3 = (threshold < 1.25 ** 3).float().mean()
        rmse = torch.sqrt(((ground_truth - prediction) ** 2).mean())
        rmse_log = torch.sqrt(((torch.log(ground_truth) - torch.log(prediction)) ** 2).mean())
        abs_rel = torch.mean(torch.abs(ground_truth - prediction) / ground_truth)
        sq_rel = torch.mean((ground_truth - prediction) ** 2 / ground_truth)
        return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
    raise ValueError("lib arg is 'numpy' or 'torch'")


def compute_depth_metric(inputs, outputs, lib="torch"):
    """reference: model_metric.py:70-105.  Train-time monitor: bilinear to the ground-truth size,
    Garg crop (the reference hard-codes 375x1242 -> rows 153:371, cols 44:1197), batch-level median
    scaling, clamp to [1e-3, 80].  Stays on the device; no host sync here."""
    gt = inputs[("depth", 0)]
    gh, gw = gt.shape[-2:]
    pred = outputs[("depth", 0, 0)].detach()
    pred = torch.clamp(TF.interpolate(pred, [gh, gw], mode="bilinear", align_corners=False), 1e-3, 80)
    mask = gt > 0
    crop = torch.zeros_like(mask)
    crop[:, :, int(0.40810811 * gh):int(0.99189189 * gh), int(0.03594771 * gw):int(0.96405229 * gw)] = 1
    mask = mask * crop
    gt_m, pred_m = gt[mask], pred[mask]
    pred_m = pred_m * (torch.median(gt_m) / torch.median(pred_m))
    pred_m = torch.clamp(pred_m, min=1e-3, max=80)
    return compute_depth_error(ground_truth=gt_m, prediction=pred_m, lib=lib)
