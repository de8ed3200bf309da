"""Depth / disparity error statistics printed by the DSGN detect-under-attack scripts
(attack/DSGN/predict_and_save_pgd.py:202-247 and :304-329; the same functions appear in predict_and_save_patch.py).

They run once per image after the detector pass (not on the perturbation hot path), on whatever device the
prediction lives on.  Semantics and float32 summation order follow the reference; tests/golden pins them with the
reference's own functions executed on seeded inputs."""
import numpy as np
import torch


def _abs_err(pred, gt, valid):
    return (pred - gt).abs()[valid]


def error_estimating(pred_disp, ground_truth, maxdisp=192):
    """Share of pixels whose disparity is off by more than 3 px AND more than 5 % (predict_and_save_pgd.py:202-214),
    per image over 0 < gt < maxdisp, summed over the images.  -> (sum, number of images)"""
    total = 0.
    for pred, gt in zip(pred_disp, ground_truth):
        valid = (gt > 0) & (gt < maxdisp)
        err = _abs_err(pred, gt, valid)
        outliers = (err > 3.) & (err / gt[valid] > 0.05)
        total = total + outliers.sum().float() / valid.sum()
    return float(total), len(pred_disp)


def depth_error_estimating(pred_disp, ground_truth, max_depth=None, depth_disp=False, calib_batch=None, calib_R_batch=None):
    """Mean and median absolute depth error per image, summed over the images (predict_and_save_pgd.py:217-247).
    depth_disp=True: the prediction already is a depth, valid where 0 < gt < max_depth (the reference reads the global
    cfg.max_depth).  depth_disp=False: it is a disparity, converted with f_u * baseline / disp from the calibration
    pair, valid where 0 < gt < 60.  An image without valid pixels contributes 0.  -> (sum of means, n, sum of medians)"""
    mean_sum, median_sum = 0., 0.
    for i, (pred, gt) in enumerate(zip(pred_disp, ground_truth)):
        if depth_disp:
            valid = (gt > 0) & (gt < max_depth)
        else:
            left, right = calib_batch[i], calib_R_batch[i]
            baseline = (left.P[0, 3] - right.P[0, 3]) / left.P[0, 0]
            pred = (left.f_u * baseline) / pred
            valid = (gt > 0) & (gt < 60)
        if valid.sum() > 0:
            err = _abs_err(pred, gt, valid)
            mean_sum = mean_sum + err.mean()
            median_sum = median_sum + err.median()
    return float(mean_sum), len(pred_disp), float(median_sum)


def _depth_of(disp, f_u, baseline, depth_disp):
    disp[disp < 0] = 0                      # in place, as the reference clips its argument
    hit = disp > 0
    return (disp if depth_disp else f_u * baseline / (disp + 1. - hit)), hit


def project_disp_to_depth_map(f_u, disp, baseline=0.54, depth_disp=False):
    """The ``--save_depth_map`` array (predict_and_save_pgd.py:304-311): numpy, negative predictions clipped to 0."""
    return _depth_of(disp, f_u, baseline, depth_disp)[0]


def project_disp_to_points(f_u, disp, baseline=0.54, depth_disp=False):
    """Image-space half of ``project_disp_to_depth`` (:314-329): one (u, v, depth) row per pixel with a positive
    prediction, row-major.  The reference hands these to the upstream ``calib.project_image_to_velo`` and keeps the
    points with x >= 0 and z < max_high (``--save_lidar``)."""
    depth, hit = _depth_of(disp, f_u, baseline, depth_disp)
    v, u = np.mgrid[0:depth.shape[0], 0:depth.shape[1]]
    pts = np.stack([u.ravel(), v.ravel(), depth.ravel()], axis=1)
    return pts[hit.ravel()]
