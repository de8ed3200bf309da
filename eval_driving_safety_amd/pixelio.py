"""Filesystem surface of the attacks: PNG folders, patch.npy, KITTI label text.  This is the contract
the reference's predict_and_save_* and evaluation/ scripts consume; pixel values come
out of the HIP export kernel, this module only crops, encodes and names files."""
import os
import queue
import threading

import numpy as np


def iter_dir(prefix, k, eye):
    """``dsgn_pgd_iters_{k}/image_{2,3}`` (attack/DSGN/pgd_attack.py:279-281,357-359);
    ``stereo_rcnn_pgd_iters_{k}/image_{2,3}`` (attack/Stereo-RCNN/pgd_attack.py:126-128,220-222)."""
    return os.path.join("%s_pgd_iters_%d" % (prefix, k), "image_2" if eye == 0 else "image_3")


def _encode_png(path, hwc_u8, bgr):
    from PIL import Image
    a = hwc_u8[:, :, ::-1] if bgr else hwc_u8      # cv2.imwrite stores a BGR array as an RGB file
    Image.fromarray(np.ascontiguousarray(a)).save(path)


class PngWriter:
    """Background PNG encoder: the attack loop hands over host uint8 arrays and continues; ``close()``
    waits.  (The reference encodes synchronously inside the loop, attack/DSGN/pgd_attack.py:357-374:
    40 PNGs per pair for N=20.)"""

    def __init__(self, workers=4, bgr=False):
        self.q = queue.Queue(maxsize=256)
        self.bgr = bgr
        self.err = None
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                path, arr = item
                os.makedirs(os.path.dirname(path), exist_ok=True)
                _encode_png(path, arr, self.bgr)
            except Exception as e:      # surfaced by close()
                self.err = e

    def put(self, path, hwc_u8, crop_w=None, crop_h=None):
        """crop (0,0,w,h) as save_img does (attack/DSGN/pgd_attack.py:192)"""
        a = hwc_u8
        if crop_h is not None or crop_w is not None:
            a = a[:crop_h, :crop_w]
        self.q.put((path, a))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err


def patch_dir(prefix, ratio, epoch, root="."):
    """``{dsgn,stereo_rcnn}_patch_ratio_{ratio}/epoch{E}`` (attack/DSGN/patch_attack.py:286,438)."""
    return os.path.join(root, "%s_patch_ratio_%s" % (prefix, ratio), "epoch%s" % epoch)


def save_patch(path_dir, patch):
    """patch.npy float32 [1,3,D,D] (attack/DSGN/patch_attack.py:229-232,438-443)."""
    os.makedirs(path_dir, exist_ok=True)
    a = np.asarray(patch, dtype=np.float32)
    if a.ndim == 3:
        a = a[None]
    np.save(os.path.join(path_dir, "patch.npy"), a)


def load_or_init_patch(path_dir, patch_dim):
    """init_patch (attack/Stereo-RCNN/patch_attack.py:67-76): resume from ``epoch0/patch.npy`` if the
    directory exists, else zeros (saved).  A patch of another size (the DSGN script's cross-model
    transfer, attack/DSGN/patch_attack.py:220-227, resizes with cv2) is rejected here - see DESIGN.md."""
    f = os.path.join(path_dir, "patch.npy")
    if os.path.isdir(path_dir) and os.path.exists(f):
        patch = np.load(f).astype(np.float32)
        if patch.shape != (1, 3, patch_dim, patch_dim):
            raise ValueError("existing patch %s is %s, expected %s (resize it offline)"
                             % (f, patch.shape, (1, 3, patch_dim, patch_dim)))
        return patch, True
    patch = np.zeros((1, 3, patch_dim, patch_dim), dtype=np.float32)
    save_patch(path_dir, patch)
    return patch, False


KITTI_FMT = ("{} -1 -1 {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.8f}\n")


def kitti_label_line(cls, bbox, score, center, dims):
    """One detection as attack/DSGN/predict_and_save_pgd.py:273-283 writes it.  ``center`` = 3D box
    centre (x,y,z), ``dims`` = (h, w, l, ry); y is shifted to the box bottom (y + h/2), alpha is the
    observation angle -atan2(x, z) + ry.  evaluation/convert_scenarios.py:74-93 splits on ' '."""
    h, w, l, ry = dims
    name = "Pedestrian" if cls == 1 else "Car" if cls == 2 else "Cyclist"
    alpha = -np.arctan2(np.float32(center[0]), np.float32(center[2])) + ry
    y = np.float32(center[1]) + np.float32(h / 2.0)
    return KITTI_FMT.format(name, alpha, bbox[0], bbox[1], bbox[2], bbox[3], h, w, l, center[0], y, center[2], ry, score)


def write_kitti_labels(output_path, image_index, detections):
    """``{:06d}.txt`` per image (predict_and_save_pgd.py:252); detections = iterable of
    (cls, bbox[4], score, center[3], (h,w,l,ry))."""
    os.makedirs(output_path, exist_ok=True)
    with open(os.path.join(output_path, "{:06d}.txt".format(image_index)), "w") as f:
        for det in detections:
            f.write(kitti_label_line(*det))
