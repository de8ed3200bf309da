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


def _png_chunk(tag, data):
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)


def _encode_png(path, hwc_u8, bgr, compress_level=1):
    """An 8-bit RGB PNG written directly: signature, IHDR, one IDAT = zlib(scanlines, each behind a filter byte), IEND.
    The reference writes through PIL (attack/DSGN/pgd_attack.py:193); any PNG reader (PIL, cv2, skimage - what the reference's
    predict_and_save_* scripts use) decodes the SAME pixels from this file - only size and encode time depend on the filter and the
    zlib level.  Why not PIL here: its encoder keeps the interpreter lock while it deflates, so a pool of writer threads does not
    scale (96 threads on a 256-thread host: 5.8 pairs/s against 6.1 with 16, profiles/r03_folder_attack.jsonl); numpy's
    subtraction and zlib.compress / crc32 release it.  Filter 1 ("Sub": each byte minus the byte one pixel to its left, modulo 256)
    keeps smooth image content compressible at level 1."""
    import struct
    import zlib
    a = hwc_u8[:, :, ::-1] if bgr else hwc_u8      # cv2.imwrite stores a BGR array as an RGB file
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    flat = a.reshape(h, 3 * w)
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 1                                   # filter type Sub
    raw[:, 1:4] = flat[:, :3]
    np.subtract(flat[:, 3:], flat[:, :-3], out=raw[:, 4:])          # uint8 arithmetic wraps modulo 256, as the filter is defined
    data = zlib.compress(raw.tobytes(), int(compress_level))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + _png_chunk(b"IDAT", data) +
                _png_chunk(b"IEND", b""))


class PngWriter:
    """Background PNG encoder: the attack loop hands over host uint8 arrays and continues; ``close()``
    waits.  (The reference encodes synchronously inside the loop, attack/DSGN/pgd_attack.py:357-374:
    40 PNGs per pair for N=20.)"""

    def __init__(self, workers=4, bgr=False, compress_level=1):
        self.q = queue.Queue(maxsize=max(256, 8 * workers))
        self.bgr, self.compress_level = bgr, int(compress_level)
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
                _encode_png(path, arr, self.bgr, self.compress_level)
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


class AsyncExporter:
    """Moves the 8-bit iterates device -> pinned host -> PNG encoder threads without stalling the attack
    stream (the reference blocks on ``.cpu()`` + numpy + zlib twice per PGD step, pgd_attack.py:357-374).

    Two device export buffers alternate: the PGD kernel of step k writes buffer k % 2 while the copy engine
    drains buffer (k-1) % 2 on a side stream into a pinned slot; a drain thread waits for that copy's event,
    takes a private copy and hands per-image views to the PngWriter.  For CPU tensors (the oracle-backed
    test shim) it degrades to a synchronous copy.
    """

    SLOTS = 2

    def __init__(self, writer, alloc, device):
        import torch
        self.torch = torch
        self.writer = writer
        self.dev_bufs = [alloc() for _ in range(self.SLOTS)]
        self.cuda = self.dev_bufs[0].is_cuda
        self.turn = 0
        self.err = None
        if self.cuda:
            self.stream = torch.cuda.Stream(device=device)
            self.pinned = [torch.empty(b.shape, dtype=b.dtype, pin_memory=True) for b in self.dev_bufs]
            self.copied = [None] * self.SLOTS                 # event: D2H of the slot finished
            self.free = [threading.Semaphore(1) for _ in range(self.SLOTS)]
            self.jobs = queue.Queue()
            self.thread = threading.Thread(target=self._drain, daemon=True)
            self.thread.start()

    def next_buffer(self):
        """device buffer the next kernel may write (its previous copy-out has been ordered before reuse)"""
        slot = self.turn % self.SLOTS
        if self.cuda and self.copied[slot] is not None:
            self.torch.cuda.current_stream().wait_event(self.copied[slot])
        return self.dev_bufs[slot]

    def submit(self, fan_out):
        """the kernel writing next_buffer() has been enqueued on the current stream; ``fan_out(host_array)``
        is called later with the [n, rows, W, 3] uint8 host copy and must enqueue the PNG jobs"""
        slot = self.turn % self.SLOTS
        self.turn += 1
        buf = self.dev_bufs[slot]
        if not self.cuda:
            fan_out(buf.detach().to("cpu", copy=True).numpy())
            return
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record()
        self.free[slot].acquire()                             # the drain thread is done with this pinned slot
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            self.pinned[slot].copy_(buf, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        self.copied[slot] = done
        self.jobs.put((slot, done, fan_out))

    def _drain(self):
        while True:
            item = self.jobs.get()
            if item is None:
                return
            slot, done, fan_out = item
            try:
                done.synchronize()
                host = self.pinned[slot].numpy().copy()
                self.free[slot].release()
                fan_out(host)
            except Exception as e:
                self.err = e
                self.free[slot].release()

    def close(self):
        if self.cuda:
            self.jobs.put(None)
            self.thread.join()
        if self.err is not None:
            raise self.err


def resize_patch_bilinear(patch, new_dim):
    """``cv2.resize(patch_hwc, (new_dim, new_dim), interpolation=cv2.INTER_LINEAR)`` of init_patch
    (attack/DSGN/patch_attack.py:224-227: a patch trained on the other detector is resized on resume).
    UNPINNED: cv2 is neither in the reference tree nor in this image; this follows OpenCV's documented
    float32 algorithm - pixel centres aligned, fx = (dx+0.5)*scale-0.5, edge clamp, separable float32
    interpolation, horizontal pass first.  patch [1,3,D,D] -> [1,3,new_dim,new_dim]."""
    a = np.asarray(patch, dtype=np.float32)[0]                 # [3, D, D]
    src = a.shape[1]
    if src == new_dim:
        return a[None].copy()
    scale = np.float64(src) / np.float64(new_dim)

    def taps(n_dst):
        idx = np.empty(n_dst, np.int64)
        w1 = np.empty(n_dst, np.float32)
        for d in range(n_dst):
            f = (d + 0.5) * scale - 0.5
            s = int(np.floor(f))
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= src - 1:
                s, f = src - 1, np.float32(0)
            idx[d], w1[d] = s, f
        return idx, w1

    ix, wx = taps(new_dim)
    ix1 = np.minimum(ix + 1, src - 1)
    rows = a[:, :, ix] * (np.float32(1) - wx) + a[:, :, ix1] * wx          # horizontal pass
    iy, wy = ix, wx                                                          # square patch: same taps
    iy1 = np.minimum(iy + 1, src - 1)
    out = rows[:, iy, :] * (np.float32(1) - wy)[None, :, None] + rows[:, iy1, :] * wy[None, :, None]
    return out[None].astype(np.float32)


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


def load_or_init_patch(path_dir, patch_dim, allow_resize=False):
    """init_patch: resume from ``epoch0/patch.npy`` if the directory exists, else zeros (saved).
    Stereo R-CNN loads the file as it is (attack/Stereo-RCNN/patch_attack.py:67-69); the DSGN script
    resizes a patch of another size bilinearly (cross-model transfer, attack/DSGN/patch_attack.py:220-227):
    ``allow_resize=True``."""
    f = os.path.join(path_dir, "patch.npy")
    if os.path.isdir(path_dir) and os.path.exists(f):
        patch = np.load(f).astype(np.float32)
        if patch.shape != (1, 3, patch_dim, patch_dim):
            if allow_resize and patch.ndim == 4 and patch.shape[:2] == (1, 3) and patch.shape[2] == patch.shape[3]:
                return resize_patch_bilinear(patch, patch_dim), True
            raise ValueError("existing patch %s is %s, expected %s" % (f, patch.shape, (1, 3, patch_dim, patch_dim)))
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


def boxlist_detections(prediction, get_dimensions, learn_viewpoint=False):
    """The (cls, bbox, score, center, dims) tuples of one upstream ``BoxList`` prediction, as ``kitti_output`` derives
    them (attack/DSGN/predict_and_save_pgd.py:253-271): centre = mean of the 8 box corners, (h, w, l, ry) from the
    upstream ``get_dimensions`` of the centred corners, optional viewpoint correction; a prediction without 3D corners
    gives zeros."""
    labels = prediction.get_field("labels").cpu()
    boxes = prediction.bbox.cpu()
    scores = prediction.get_field("scores").cpu()
    corners = prediction.get_field("box_corner3d").cpu() if prediction.has_field("box_corner3d") else None
    out = []
    for i in range(len(labels)):
        if corners is not None:
            assert labels[i] != 0
            c = corners[i].reshape(8, 3)
            center = c.mean(dim=0)
            h, w, l, ry = get_dimensions((c - center.view(1, 3)).transpose(0, 1))
            if learn_viewpoint:
                ry = ry - np.arctan2(center[2], center[0]) + np.pi / 2
        else:
            h, w, l, ry, center = 0., 0., 0., 0., [0., 0., 0.]
        out.append((labels[i], boxes[i], scores[i], center, (h, w, l, ry)))
    return out


def kitti_output(box_pred_left, image_indexes, output_path, get_dimensions, learn_viewpoint=False, log=print):
    """``kitti_output`` of the DSGN detect scripts (predict_and_save_pgd.py:250-284): one label file per image"""
    for prediction, image_index in zip(box_pred_left, image_indexes):
        write_kitti_labels(output_path, image_index, boxlist_detections(prediction, get_dimensions, learn_viewpoint))
        log("Wrote {}".format(image_index))


# --- result folders of the detect-under-attack scripts -------------------------------------------------
def dsgn_tag(tag="", iter_num=None, alpha=None, ratio=None, epochs=None, debugnum=None, train=False):
    """args.tag as the DSGN scripts extend it (attack/DSGN/predict_and_save_pgd.py:84-94,
    predict_and_save_patch.py:88-98): debug{n}, _train, _iter{i}_alpha{a} / _ratio{r}_epochs{E}."""
    if debugnum is not None:
        tag += "debug{}".format(debugnum)
    if train:
        tag += "_train"
    if alpha and iter_num:
        tag += "_iter{0}_alpha{1}".format(str(iter_num), str(alpha))
    if ratio is not None and epochs is not None:
        tag += "_ratio{0}_epochs{1}".format(str(ratio), str(epochs))
    return tag


def dsgn_label_dir(loadmodel, tag):
    """``<dirname(loadmodel)>/kitti_output{tag}`` (predict_and_save_pgd.py:334-341)"""
    return os.path.join(os.path.dirname(loadmodel), "kitti_output" + tag)


def srcnn_result_dir(iter_num=None, alpha=None, ratio=None, epochs=None):
    """``result_stereo_rcnn_pgd_{iter}_{alpha}`` (attack/Stereo-RCNN/predict_and_save_pgd.py:75) or
    ``result_stereo_rcnn_ratio_{r}/epoch{E}`` (predict_and_save_patch.py:137)"""
    if ratio is not None:
        return "result_stereo_rcnn_ratio_{0}/epoch{1}".format(ratio, epochs)
    return "result_stereo_rcnn_pgd_{0}_{1}".format(iter_num, alpha)


def srcnn_im_info_prescaled(im_info, scale_target=600, native_height=375):
    """The Stereo R-CNN attack saves its adversarial PNGs at network scale; when they are read back the
    eval script forces ``im_info[0][2] = 600 / 375`` so boxes map to KITTI coordinates
    (attack/Stereo-RCNN/predict_and_save_pgd.py:134-136, quirk Q14).  Returns a modified copy."""
    info = np.array(im_info, dtype=np.float32, copy=True)
    info[0][2] = float(scale_target) / float(native_height)
    return info


# --- the consumer's view of a label file (validation of what this engine writes) ---------------------------
def load_label(label_path):
    """How ``evaluation/convert_scenarios.py:52-95`` reads a label file: split on single spaces, fields
    0 type, 1 truncated, 2 occluded, 3 alpha, 4-7 bbox, 8-10 (h, w, l), 11-13 (x, y, z), 14 rotation_y.
    Returns a list of [type, truncated, occluded, alpha, [bbox], [h, w, l], [x, y, z], ry]."""
    label = []
    with open(label_path, "r") as f:
        for line in f:
            e = line.strip().split(" ")
            label.append([e[0], float(e[1]), float(e[2]), float(e[3]), [float(v) for v in e[4:8]],
                          [float(v) for v in e[8:11]], [float(v) for v in e[11:14]], float(e[14])])
    return label


SCENARIO_TYPES = ("Car", "Van", "Truck", "Misc")     # evaluation/convert_scenarios.py:117


def scenario_obstacles(label):
    """The static obstacles ``convert_scenario`` (evaluation/convert_scenarios.py:116-133) builds from a label:
    only Car/Van/Truck/Misc, rotation wrapped into [-pi, pi], rectangle width = w, length = l, position
    (z, -x) in the scenario frame, orientation -(ry - pi/2).  For checking label files without CommonRoad."""
    out = []
    for item in label:
        if item[0] not in SCENARIO_TYPES:
            continue
        orient = item[7]
        while orient < -np.pi:
            orient += 2 * np.pi
        while orient > np.pi:
            orient -= 2 * np.pi
        out.append(dict(width=item[5][1], length=item[5][2], position=[item[6][2], -item[6][0]],
                        orientation=-(orient - 0.5 * np.pi)))
    return out
