"""The loop shared by the two DSGN detect-under-attack counterparts
(attack/DSGN/predict_and_save_pgd.py:332-455, predict_and_save_patch.py:394-554): detector forward without gradients on
attacked pairs, one KITTI label file per image, the depth-error statistics, the optional depth-map / pseudo-lidar /
feature-map dumps, the KITTI evaluation hand-off and the ``result_<checkpoint>.txt`` lines.

The detector, its post-processor, ``get_dimensions`` and the evaluation script are upstream code (``rt`` is an
``upstream.DsgnRuntime``); pasting (HIP), label text, statistics and the file layout are this package's."""
import importlib
import os
import shutil
import subprocess
import time

import numpy as np
import torch

from .. import depthstats, patchgeom, pixelio
from ..attacks import DetectUnderAttack


def find_get_dimensions():
    """``get_dimensions`` arrives in the scripts through ``from dsgn.utils.* import *`` (predict_and_save_pgd.py:27-29)"""
    for mod in ("dsgn.utils.numpy_utils", "dsgn.utils.numba_utils", "dsgn.utils.torch_utils"):
        try:
            m = importlib.import_module(mod)
        except Exception:
            continue
        if hasattr(m, "get_dimensions"):
            return m.get_dimensions
    raise ImportError("get_dimensions not found in dsgn.utils.{numpy,numba,torch}_utils (upstream DSGN)")


def load_patch_for_detection(patch_dir, ratio, epochs):
    """``init_patch`` of predict_and_save_patch.py:339-358: the trained patch must exist; a patch of another size (one
    trained on Stereo R-CNN) is resized bilinearly to the DSGN diameter."""
    d = "{0}/dsgn_patch_ratio_{1}/epoch{2}".format(patch_dir, ratio, epochs)
    patch_dim, radius = patchgeom.init_patch_dims(patchgeom.DSGN_SHAPE[0], ratio)
    if not os.path.isdir(d):
        raise Exception("Patch directory NOT found.")
    patch, _ = pixelio.load_or_init_patch(d, patch_dim, allow_resize=True)
    return patch_dim, radius, patch


def feature_hooks(model, sink):
    """``--save_feat_map``: forward hooks on the children of the upstream feature extractor (predict_and_save_patch.py:
    157-167); outputs are appended to ``sink``"""
    inner = model.module if hasattr(model, "module") else model
    handles = []
    for child in inner.children():
        if type(child).__name__ == "feature_extraction":
            for sub in child.children():
                handles.append(sub.register_forward_hook(lambda m, i, o: sink.append(o)))
            break
    return handles


def kitti_eval(output_path, loadmodel, tag, valid_classes):
    """the shell-out of predict_and_save_pgd.py:287-301 to the upstream kitti-object-eval-python checkout"""
    eval_dir = "./dsgn/eval/kitti-object-eval-python"
    ckpt = loadmodel.split("/")[-1].split(".")[0] if loadmodel is not None else "default"
    result = os.path.join(output_path, "result_kitti_{}{}.txt".format(ckpt, tag))
    if not os.path.isdir(eval_dir):
        print("kitti-object-eval-python not found under ./dsgn/eval - AP evaluation skipped (labels are in {})".format(
            os.path.join(output_path, "kitti_output" + tag)))
        return None
    for i, cls in enumerate(valid_classes):
        with open(result, "w" if i == 0 else "a") as f:
            subprocess.run(["bash", "eval.sh", os.path.abspath(os.path.join(output_path, "kitti_output" + tag)),
                            str(0 if cls == 2 else (1 if cls == 1 else 2))], cwd=eval_dir, stdout=f, stderr=subprocess.STDOUT)
    with open(result) as f:
        print(f.read())
    return result


def run(args, rt, mode, dev):
    cfg = rt.cfg
    get_dimensions = find_get_dimensions() if cfg.RPN3D_ENABLE else None
    if not os.path.isdir(args.save_path):
        os.makedirs(args.save_path)
    output_path = os.path.dirname(args.loadmodel)                              # :334-335 (relative to the checkout root)
    label_dir = os.path.join(output_path, "kitti_output" + args.tag)
    if os.path.exists(label_dir):                                              # :336-341
        shutil.rmtree(label_dir)
    os.makedirs(label_dir)
    patch = None
    if mode == "patch":
        _, _, host = load_patch_for_detection(args.patch_dir, args.ratio, args.epochs)
        patch = torch.from_numpy(host).to(dev)
    det = DetectUnderAttack("dsgn", mode, label_dir, patch=patch, atk_mode=getattr(args, "atk_mode", "random"),
                            seed=args.pos_seed, device=dev)
    feats = []
    if args.save_feat_map:
        feature_hooks(rt.model, feats)
    all_err, all_err_med, written = 0., 0., 0
    for batch_idx, batch in enumerate(rt.detect_batches()):
        if getattr(cfg, "debug", False) and batch_idx * len(batch) > args.debugnum:      # :349-351
            break
        x = det.prepare(batch)                                                 # patch mode: wrong-shape pairs skipped, rest pasted
        if x is None:
            continue
        extra = batch.extra
        start_time = time.time()
        cfg.time = time.time()
        del feats[:]
        output = rt.predict(x, extra)
        if args.save_feat_map:                                                 # :370-389
            feat_out_dir = "{0}/{1}".format(args.save_feat_path, batch.names[0])
            os.makedirs(feat_out_dir, exist_ok=True)
            flat = [t for f in feats for t in ([f] if isinstance(f, torch.Tensor) else list(f))]
            for k, t in enumerate(flat):
                np.save("{}/feat_out{}.npy".format(feat_out_dir, k), t.detach().cpu().numpy())
        if cfg.RPN3D_ENABLE:                                                   # :391-395
            pred_disp, box_pred = output
            pixelio.kitti_output(box_pred[0], extra.image_indexes, label_dir, get_dimensions, getattr(cfg, "learn_viewpoint", False))
            written += len(extra.image_indexes)
        else:
            pred_disp, = output
        print("time = %.2f" % (time.time() - start_time))
        if getattr(cfg, "PlaneSweepVolume", True) and getattr(cfg, "loss_disp", True) and len(pred_disp) > 0:   # :397-410
            gt_disp = extra.gt_disp.to(pred_disp[0].device) if hasattr(pred_disp[0], "device") else extra.gt_disp
            if cfg.eval_depth:
                err, n, err_med = depthstats.depth_error_estimating(pred_disp, gt_disp, max_depth=cfg.max_depth, depth_disp=True)
                print("Mean depth error(m): {} Median(m): {} (batch {})".format(err / n, err_med / n, n))
                all_err += err
                all_err_med += err_med
            else:
                err, n = depthstats.error_estimating(pred_disp, gt_disp)
                print(">3px error: {} (batch {})".format(err / n, n))
                all_err += err
        if args.save_depth_map or args.save_lidar:                             # :412-436
            for i, idx in enumerate(extra.image_indexes):
                c, c_r = extra.calib[i], extra.calib_R[i]
                hh, ww = extra.image_sizes[i][0], extra.image_sizes[i][1]
                base = (c.P[0, 3] - c_r.P[0, 3]) / c.P[0, 0]
                if args.save_depth_map:
                    os.makedirs("{}/depth_maps/".format(args.save_path), exist_ok=True)
                    dm = depthstats.project_disp_to_depth_map(c.f_u, pred_disp[i].cpu().numpy()[:hh, :ww], baseline=base, depth_disp=True)
                    np.save("{}/depth_maps/{:06d}.npy".format(args.save_path, idx), dm)
                if args.save_lidar:
                    pts = depthstats.project_disp_to_points(c.f_u, pred_disp[i].cpu().numpy()[:hh, :ww], baseline=base, depth_disp=True)
                    cloud = c.project_image_to_velo(pts)                        # upstream calibration object
                    cloud = cloud[(cloud[:, 0] >= 0) & (cloud[:, 2] < 1.)]
                    np.concatenate([cloud, np.ones((cloud.shape[0], 1))], 1).astype(np.float32).tofile(
                        "{}/{:06d}.bin".format(args.save_path, idx))
    if cfg.RPN3D_ENABLE:
        kitti_eval(output_path, args.loadmodel, args.tag, cfg.valid_classes)   # :440-441
    print(args.loadmodel)
    n_all = max(1, len(rt.dataset))
    all_err /= n_all
    all_err_med /= n_all
    print("Mean Error", all_err)                                               # :447-455
    result_txt = "{}/result_{}.txt".format(os.path.dirname(args.loadmodel), args.loadmodel.split("/")[-1].split(".")[0])
    with open(result_txt, "a") as f:
        f.write("Mean Error: {}\n".format(all_err))
        if cfg.eval_depth:
            print("Median Error", all_err_med)
            f.write("Median Error: {}\n".format(all_err_med))
    return written, label_dir


def run_shaped(args, mode, dev):
    """``--model shaped``: the same file surface around the DSGN-shaped surrogate (adapters.PsvStereoAdapter.detect) - a PNG
    folder in, ``<out_root>/kitti_output{tag}/NNNNNN.txt`` out - without an upstream checkout (no depth ground truth, so no
    depth statistics)."""
    from .. import adapters, data
    from . import _common
    if mode == "pgd" and args.alpha and args.iter:
        args.tag += "_iter{0}_alpha{1}".format(str(args.iter), str(args.alpha))
    if mode == "patch":
        args.tag += "_ratio{0}_epochs{1}".format(args.ratio, args.epochs)
    label_dir = os.path.join(args.out_root, "kitti_output" + args.tag)
    if os.path.exists(label_dir):
        shutil.rmtree(label_dir)
    os.makedirs(label_dir)
    patch = None
    if mode == "patch":
        _, _, host = load_patch_for_detection(args.patch_dir, args.ratio, args.epochs)
        patch = torch.from_numpy(host).to(dev)
    batch = args.btest if args.btest else 1
    workers = args.loader_workers if args.loader_workers is not None else (0 if args.debug else 12)
    loader = data.SyntheticStereo(args.synthetic, "dsgn", batch, seed=args.seed) if args.synthetic \
        else data.KittiFolder(args.data_path, args.split_file, batch, workers=workers)
    det = DetectUnderAttack("dsgn", mode, label_dir, patch=patch, atk_mode=getattr(args, "atk_mode", "random"), seed=args.pos_seed, device=dev)
    net = _common.layerlist_dsgn(dev, args) if args.model == "layerlist" else adapters.PsvStereoAdapter(dev, seed=args.seed, hourglass=True, dsgn_head=True)
    n = det.run(loader, net, debugnum=args.debugnum if args.debug else None)
    print("wrote %d label files to %s" % (n, label_dir))
    return n, label_dir
