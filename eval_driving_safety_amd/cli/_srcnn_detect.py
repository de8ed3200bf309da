"""The loop shared by the two Stereo R-CNN detect-under-attack counterparts
(attack/Stereo-RCNN/predict_and_save_pgd.py:72-427, predict_and_save_patch.py:133-539).

The network, the box / keypoint decoders, the 3D box estimator and the KITTI result writer are upstream code and are
called under the names the scripts import them by.  This package supplies: the patch paste (HIP), the per-class NMS
(``adv_nms_f32`` - deterministic, bit-exact indices), the dense photometric alignment (``adv_dense_align_f32``), the
``im_info`` scale override for pre-scaled adversarial PNGs, and the result-folder layout."""
import importlib
import math as m
import os
import types

import numpy as np
import torch

from .. import patchgeom, pixelio
from ..attacks import DetectUnderAttack


def load_patch_for_detection(patch_dir, ratio, epochs):
    """``init_patch`` of predict_and_save_patch.py:59-79 (the trained patch must exist)"""
    d = "{0}/stereo_rcnn_patch_ratio_{1}/epoch{2}".format(patch_dir, ratio, epochs)
    patch_dim, radius = patchgeom.init_patch_dims(patchgeom.SRCNN_SHAPE[0], ratio)
    if not os.path.isdir(d):
        raise Exception("Patch directory NOT found.")
    patch, _ = pixelio.load_or_init_patch(d, patch_dim, allow_resize=True)
    return patch_dim, radius, patch


def _upstream():
    imp = importlib.import_module
    bt = imp("model.rpn.bbox_transform")
    return types.SimpleNamespace(
        bbox_transform_inv=bt.bbox_transform_inv, kpts_transform_inv=bt.kpts_transform_inv, border_transform_inv=bt.border_transform_inv,
        clip_boxes=bt.clip_boxes, kitti_utils=imp("model.utils.kitti_utils"), box_estimator=imp("model.utils.box_estimator"))


def split_box_deltas(bbox_pred, n_classes):
    """the six regression outputs per class -> left (dx, dy, dw, dh) and right (dx', dy, dw', dh) deltas
    (predict_and_save_pgd.py:168-184; y and height are shared by the two eyes)"""
    p = bbox_pred[0]
    left = torch.stack([p[:, 0::6], p[:, 1::6], p[:, 2::6], p[:, 3::6]], dim=2).reshape(p.shape[0], 4 * n_classes)
    right = torch.stack([p[:, 4::6], p[:, 1::6], p[:, 5::6], p[:, 3::6]], dim=2).reshape(p.shape[0], 4 * n_classes)
    return left.reshape(-1, 4), right.reshape(-1, 4)


def run(args, rt, mode, dev, ops):
    cfg, imdb, roidb = rt.cfg, rt.imdb, rt.roidb
    up = _upstream()
    n_cls = len(imdb._classes)
    if mode == "patch":
        result_dir = os.path.join(args.out_root, pixelio.srcnn_result_dir(ratio=args.ratio, epochs=args.epochs))      # :137
        _, _, host = load_patch_for_detection(args.patch_dir, args.ratio, args.epochs)
        patch = torch.from_numpy(host).to(dev)
    else:
        result_dir = os.path.join(args.out_root, pixelio.srcnn_result_dir(iter_num=args.iter, alpha=args.alpha))        # :75
        patch = None
    os.makedirs(result_dir, exist_ok=True)
    det = DetectUnderAttack("srcnn", mode, result_dir, patch=patch, atk_mode=getattr(args, "atk_mode", "random"), seed=args.pos_seed,
                            device=dev)
    eval_thresh = 0.05                                                          # :117-118
    stds = torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_STDS, dtype=torch.float32, device=dev)
    means = torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_MEANS, dtype=torch.float32, device=dev)
    dstds = torch.tensor(cfg.TRAIN.DIM_NORMALIZE_STDS, dtype=torch.float32, device=dev)
    dmeans = torch.tensor(cfg.TRAIN.DIM_NORMALIZE_MEANS, dtype=torch.float32, device=dev)
    written = 0
    for i, batch in enumerate(rt.batches()):
        if mode == "patch" and getattr(args, "debug", False) and args.debugnum is not None and i > args.debugnum:      # patch :205-206
            break
        extra = batch.extra
        if mode == "pgd":   # the attacked PNGs are already at network scale: boxes map back with 600/375 (pgd :134-136, quirk Q14)
            extra.im_info = torch.from_numpy(pixelio.srcnn_im_info_prescaled(extra.im_info.cpu().numpy(), cfg.TRAIN.SCALES[0], 375)).to(dev)
        x = det.prepare(batch)
        if x is None:
            continue
        im_left, im_right = x[:1], x[1:]
        with torch.no_grad():
            out = rt.model(im_left, im_right, extra.im_info, extra.gt_boxes_left, extra.gt_boxes_right, extra.gt_boxes_merge,
                           extra.gt_dim_orien, extra.gt_kpts, extra.num_boxes)
        rois_left, rois_right, cls_prob, bbox_pred, bbox_pred_dim, kpts_prob, left_prob, right_prob = out[:8]
        scores = cls_prob.data
        boxes_left, boxes_right = rois_left.data[:, :, 1:5], rois_right.data[:, :, 1:5]
        box_delta_left, box_delta_right = split_box_deltas(bbox_pred.data, n_cls)
        dim_orien = bbox_pred_dim.data.view(-1, 5)
        max_prob, kpts_delta = torch.max(kpts_prob.data.view(-1, 4 * cfg.KPTS_GRID), 1)                                 # :190-200
        _, left_delta = torch.max(left_prob.data.view(-1, cfg.KPTS_GRID), 1)
        _, right_delta = torch.max(right_prob.data.view(-1, cfg.KPTS_GRID), 1)
        box_delta_left = (box_delta_left * stds + means).view(1, -1, 4 * n_cls)                                         # :202-219
        box_delta_right = (box_delta_right * stds + means).view(1, -1, 4 * n_cls)
        dim_orien = (dim_orien * dstds + dmeans).view(1, -1, 5 * n_cls)
        kpts_delta, left_delta, right_delta = kpts_delta.view(1, -1, 1), left_delta.view(1, -1, 1), right_delta.view(1, -1, 1)
        max_prob = max_prob.view(1, -1, 1)
        pred_boxes_left = up.bbox_transform_inv(boxes_left, box_delta_left, 1)                                         # :221-231
        pred_boxes_right = up.bbox_transform_inv(boxes_right, box_delta_right, 1)
        pred_kpts, kpts_type = up.kpts_transform_inv(boxes_left, kpts_delta, cfg.KPTS_GRID)
        pred_left = up.border_transform_inv(boxes_left, left_delta, cfg.KPTS_GRID)
        pred_right = up.border_transform_inv(boxes_left, right_delta, cfg.KPTS_GRID)
        pred_boxes_left = up.clip_boxes(pred_boxes_left, extra.im_info.data, 1)
        pred_boxes_right = up.clip_boxes(pred_boxes_right, extra.im_info.data, 1)
        scale = extra.im_info[0, 2].data
        pred_boxes_left, pred_boxes_right = pred_boxes_left / scale, pred_boxes_right / scale                         # :236-240
        pred_kpts, pred_left, pred_right = pred_kpts / scale, pred_left / scale, pred_right / scale
        scores = scores.squeeze()
        pred_boxes_left, pred_boxes_right = pred_boxes_left.squeeze(), pred_boxes_right.squeeze()
        pred_kpts = torch.cat((pred_kpts, kpts_type, max_prob, pred_left, pred_right), 2).squeeze()
        dim_orien = dim_orien.squeeze()

        img_path = roidb[i]["img_left"]                                                                                 # :251-259
        image_number = img_path.split("/")[-1].split(".")[0]
        calib_path = img_path.replace("image_2", "calib").replace("png", "txt")
        calib = up.kitti_utils.read_obj_calibration(calib_path)
        im_shape = (int(round(float(extra.im_info[0, 0]) / float(scale))), int(round(float(extra.im_info[0, 1]) / float(scale))), 3)

        for j in range(1, imdb.num_classes):                                                                            # :277-
            inds = torch.nonzero(scores[:, j] > eval_thresh).view(-1)
            if inds.numel() == 0:
                continue
            cls_scores = scores[:, j][inds]
            _, order = torch.sort(cls_scores, 0, True)
            cls_boxes_left = pred_boxes_left[inds][:, j * 4:(j + 1) * 4]
            cls_boxes_right = pred_boxes_right[inds][:, j * 4:(j + 1) * 4]
            cls_dim_orien = dim_orien[inds][:, j * 5:(j + 1) * 5]
            cls_kpts = pred_kpts[inds]
            cls_dets_left = torch.cat((cls_boxes_left, cls_scores.unsqueeze(1)), 1)[order]
            cls_dets_right = torch.cat((cls_boxes_right, cls_scores.unsqueeze(1)), 1)[order]
            cls_dim_orien, cls_kpts = cls_dim_orien[order], cls_kpts[order]
            keep = ops.nms(cls_boxes_left[order, :].contiguous(), cls_scores[order], cfg.TEST.NMS).view(-1).long()       # :300 (HIP, deterministic)
            cls_dets_left, cls_dets_right = cls_dets_left[keep], cls_dets_right[keep]
            cls_dim_orien, cls_kpts = cls_dim_orien[keep], cls_kpts[keep]
            infered_kpts = torch.from_numpy(up.kitti_utils.infer_boundary(im_shape, cls_dets_left.cpu().numpy())).type_as(cls_dets_left)
            for d in range(cls_dets_left.size()[0]):                                                                    # :316-321
                if (cls_kpts[d, 4] - cls_kpts[d, 3]) < 0.5 * (infered_kpts[d, 1] - infered_kpts[d, 0]):
                    cls_kpts[d, 3:5] = infered_kpts[d]
            boxes_all, kpts_all, poses_all = cls_dets_left.new(0, 5), cls_dets_left.new(0, 5), cls_dets_left.new(0, 8)
            for d in range(cls_dets_left.size()[0]):                                                                    # :340-378
                if cls_dets_left[d, -1] > eval_thresh:
                    box_left = cls_dets_left[d, 0:4].cpu().numpy()
                    box_right = cls_dets_right[d, 0:4].cpu().numpy()
                    dim = cls_dim_orien[d, 0:3].cpu().numpy()
                    alpha = m.atan2(cls_dim_orien[d, 3], cls_dim_orien[d, 4])
                    status, state = up.box_estimator.solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, box_left, box_right,
                                                                                cls_kpts[d].cpu().numpy())
                    if status > 0:
                        poses = im_left.new_zeros(8)
                        poses[0], poses[1], poses[2] = float(state[0]), float(state[1]), float(state[2])
                        poses[3], poses[4], poses[5] = float(dim[0]), float(dim[1]), float(dim[2])
                        poses[6], poses[7] = float(state[3]), alpha
                        boxes_all = torch.cat((boxes_all, cls_dets_left[d, 0:5].unsqueeze(0)), 0)
                        kpts_all = torch.cat((kpts_all, cls_kpts[d].unsqueeze(0)), 0)
                        poses_all = torch.cat((poses_all, poses.unsqueeze(0)), 0)
            if boxes_all.shape[0] == 0:
                continue
            # disparity by dense photometric alignment on the network-scale pair (:381; HIP: ops.dense_align)
            succ, dis_final = ops.dense_align(calib, scale, im_left, im_right, boxes_all[:, 0:4], kpts_all, poses_all[:, 0:7])
            for s in range(succ.size(0)):                                                                               # :387-416
                if succ[s] > 0:
                    box_left = boxes_all[s, 0:4].cpu().numpy()
                    score = boxes_all[s, 4].cpu().numpy()
                    dim = poses_all[s, 3:6].cpu().numpy()
                    state_rect, z = up.box_estimator.solve_x_y_theta_from_kpt(im_shape, calib, poses_all[s, 7].cpu().numpy(), dim, box_left,
                                                                              dis_final[s].cpu().numpy(), kpts_all[s].cpu().numpy())
                    xyz = np.array([state_rect[0], state_rect[1], z])
                    up.kitti_utils.write_detection_results(result_dir, image_number, calib, box_left, xyz, dim, state_rect[2], score)
                    written += 1
    return written, result_dir
