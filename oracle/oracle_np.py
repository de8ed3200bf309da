"""ORACLE - numpy restatement of the reference's perturbation inner loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``eval_driving_safety_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker.  The shipped path is the HIP library behind
``include/advengine.h``.

Parity status: PINNED for every function below except where a docstring says
"unpinned".  ``tests/golden/make_golden.py`` executes the reference's own statements
(lifted out of its scripts by AST line range) under torch-CPU on seeded inputs;
``tests/test_oracle_golden.py`` demands bit-for-bit agreement of this file with those
outputs.  All citations are relative to /root/reference.

Arithmetic conventions that make the restatement bit-exact (each was checked against
torch 2.10 CPU):
  * a float32 tensor combined with a Python float computes in float32 with the scalar
    rounded to float32 first (``x * 0.229`` == ``x * float32(0.229)``);
  * ``torch.clamp(min=a, max=b)`` rounds a, b to float32, propagates NaN, and is
    ``min(max(x, a), b)``;
  * ``torch.sign`` is ``(x > 0) - (x < 0)``: sign(nan) = sign(-0.0) = +0.0;
  * every intermediate is rounded to float32 separately (no fused multiply-add).
"""
import numpy as np

F32 = np.float32

# attack/DSGN/pgd_attack.py:153-154
DSGN_MEAN = (0.485, 0.456, 0.406)
DSGN_STD = (0.229, 0.224, 0.225)
# attack/Stereo-RCNN/pgd_attack.py:190-203 (same constants in patch_attack.py:272-277)
SRCNN_PIXEL_MEANS = (102.9801, 115.9465, 122.7717)
SRCNN_LO = tuple(F32(0 - m) for m in SRCNN_PIXEL_MEANS)
SRCNN_HI = tuple(F32(255 - m) for m in SRCNN_PIXEL_MEANS)


def _f32(a):
    a = np.asarray(a)
    assert a.dtype == np.float32, a.dtype
    return a


def torch_sign(g):
    """torch.sign for float32: (g > 0) - (g < 0); nan and +-0 give +0."""
    g = _f32(g)
    return (g > 0).astype(F32) - (g < 0).astype(F32)


def torch_clamp(x, lo, hi):
    """torch.clamp(x, min=lo, max=hi) with float32 bounds.  NaN propagates, and on ties between zeros of
    different sign the INPUT wins (ATen: ``min_ps(max, max_ps(min, x))`` returns its second operand on
    equality; the scalar tail ``std::min(std::max(x, lo), hi)`` agrees).  np.minimum / np.maximum are not used
    because their SIMD paths resolve +-0 ties the other way (pinned by the golden case srcnn_pgd_zero_eps)."""
    lo, hi = F32(lo), F32(hi)
    with np.errstate(invalid="ignore"):
        return np.where(x < lo, lo, np.where(x > hi, hi, x)).astype(np.float32)


# ----------------------------------------------------------------------------- a1 / a2
def denormalize(im):
    """attack/DSGN/pgd_attack.py:196-200 - ``im[c] = im[c] * std[c] + mean[c]``.

    The reference touches only batch element 0 (quirk Q1); its callers always have
    batch 1.  Here every batch element is processed the same way.  Returns a new array.
    """
    im = _f32(im)
    out = np.empty_like(im)
    for c in range(3):
        out[:, c] = im[:, c] * F32(DSGN_STD[c]) + F32(DSGN_MEAN[c])
    return out


def normalize(im):
    """attack/DSGN/pgd_attack.py:203-207 - ``(im[c] - mean[c]) / std[c]`` (true division)."""
    im = _f32(im)
    out = np.empty_like(im)
    with np.errstate(invalid="ignore"):
        for c in range(3):
            out[:, c] = (im[:, c] - F32(DSGN_MEAN[c])) / F32(DSGN_STD[c])
    return out


# ----------------------------------------------------------------------------- a3
def pgd_step_norm01(x, grad, clean, alpha, eps):
    """One DSGN PGD step, attack/DSGN/pgd_attack.py:339-354.

    x      normalised image [N,3,H,W] (the tensor the detector saw)
    grad   d loss / d x, same shape
    clean  DENORMALISED clean image (pgd_attack.py:297-298; ``ori`` at :254-255 is the
           same tensor, quirk Q2)
    returns the next normalised iterate.
    """
    x, grad, clean = _f32(x), _f32(grad), _f32(clean)
    with np.errstate(invalid="ignore", over="ignore"):
        d = denormalize(x)                                         # :339-340
        adv = d + F32(alpha) * torch_sign(grad)                    # :343-344
        eta = torch_clamp(adv - clean, -eps, eps)                  # :346-347
        y = torch_clamp(clean + eta, 0, 1)                         # :349-350
        return normalize(y)                                        # :353-354


# ----------------------------------------------------------------------------- a13
def pgd_step_meansub255(x, grad, clean, alpha, eps255):
    """One Stereo R-CNN PGD step, attack/Stereo-RCNN/pgd_attack.py:177-217.

    Images are BGR minus PIXEL_MEANS on the 0..255 scale; ``eps255`` is already
    ``255 * args.eps`` (pgd_attack.py:57); ``clean`` is the un-attacked input (:122-123).
    """
    x, grad, clean = _f32(x), _f32(grad), _f32(clean)
    with np.errstate(invalid="ignore", over="ignore"):
        adv = x + F32(alpha) * torch_sign(grad)                    # :177-179
        eta = torch_clamp(adv - clean, -eps255, eps255)            # :181-184
        holder = clean + eta                                       # :186-187
        out = np.empty_like(holder)
        for c in range(3):                                         # :189-207
            out[:, c] = torch_clamp(holder[:, c], SRCNN_LO[c], SRCNN_HI[c])
        return out                                                 # :209-217


# ----------------------------------------------------------------------------- a5
def tensor2im_u8(x_norm, crop_h, crop_w):
    """attack/DSGN/pgd_attack.py:157-193 (tensor2im + save_img's crop) for one image.

    x_norm [3,H,W] normalised float32 -> uint8 [crop_h, crop_w, 3].  Denormalise in
    float32, times 255 in float32, then ``astype(uint8)`` which TRUNCATES toward zero
    (and wraps modulo 256 for out-of-range values, as the C conversion numpy performs on
    x86-64 does: float -> int32 -> low byte).  PIL's ``crop((0, 0, w, h))`` keeps the
    top-left w x h window.
    """
    x = _f32(x_norm)
    assert x.ndim == 3 and x.shape[0] == 3
    im = np.empty_like(x)
    for c in range(3):                                             # :173-174
        im[c] = x[c] * F32(DSGN_STD[c]) + F32(DSGN_MEAN[c])
    im = im * F32(255)                                             # :175
    im = np.transpose(im, (1, 2, 0))                               # :176
    with np.errstate(invalid="ignore"):
        u8 = im.astype(np.int32).astype(np.uint8)                  # :179
    return np.ascontiguousarray(u8[:crop_h, :crop_w])              # :192


def srcnn_hwc_plus_means(x):
    """attack/Stereo-RCNN/pgd_attack.py:233-236 - CHW -> HWC, ``+= cfg.PIXEL_MEANS``.

    PIXEL_MEANS is a float64 array in the upstream config, so numpy's in-place add
    computes in float64 and rounds once to float32.
    """
    x = _f32(x)
    hwc = np.ascontiguousarray(np.transpose(x, (1, 2, 0)))
    means = np.array([[list(SRCNN_PIXEL_MEANS)]], dtype=np.float64)
    return (hwc.astype(np.float64) + means).astype(F32)


def srcnn_export_u8(x):
    """What ``cv2.imwrite`` (pgd_attack.py:237) stores for the float32 HWC image of
    ``srcnn_hwc_plus_means``: OpenCV converts to 8-bit with saturate_cast<uchar>(float) =
    cvRound (round half to even; on x86-64 a value that does not fit int32, or NaN, becomes
    INT_MIN) then clip to [0, 255].  UNPINNED: cv2 is neither in the reference tree nor in this
    image; this follows OpenCV's documented conversion."""
    f = srcnn_hwc_plus_means(x)
    with np.errstate(invalid="ignore"):
        ok = np.abs(f) < F32(2147483648.0)
        r = np.rint(np.where(ok, f, F32(0)).astype(np.float64))
    r = np.where(ok, r, -1.0)
    return np.clip(r, 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------- a6 / a7
def init_patch_dims(short_side, ratio):
    """attack/DSGN/patch_attack.py:213-218 (384) / attack/Stereo-RCNN/patch_attack.py:60-65 (600)."""
    patch_dim = int(short_side * ratio)
    if patch_dim % 2 == 0:
        patch_dim += 1
    return patch_dim, int(patch_dim / 2)


# column bands of generate_round_mask, as fractions of the image width:
# attack/DSGN/patch_attack.py:240 ('random'); attack/DSGN/predict_and_save_patch.py:366-373
ATK_MODE_BANDS = {"random": (0.2, 0.8), "sp_left": (0.2, 0.4),
                  "sp_straight": (0.4, 0.6), "sp_right": (0.6, 0.8)}


def round_mask_centers(rng, h, w, radius, atk_mode="random"):
    """Centres drawn by generate_round_mask (attack/DSGN/patch_attack.py:239-243;
    attack/Stereo-RCNN/patch_attack.py:81-84): two ``randint`` calls on the Python
    ``random`` stream, row first; right-eye column is ``int(cx - 40 * 1.6)``.
    ``rng`` is the ``random`` module or a ``random.Random``."""
    lo, hi = ATK_MODE_BANDS[atk_mode]
    cy = rng.randint(int(h * 0.4), int(h - radius - 1))
    cx = rng.randint(int(w * lo), int(w * hi))
    return [cy, cx], [cy, int(cx - (40 * 1.6))]


def disc_mask(h, w, cy, cx, radius):
    """attack/DSGN/patch_attack.py:245-248 - float32 [h,w], 1 inside the closed disc."""
    Y, X = np.ogrid[:h, :w]
    dist = np.sqrt((Y - cy) ** 2 + (X - cx) ** 2)
    return (dist <= radius).astype("float32")


# ----------------------------------------------------------------------------- a8
def patch_paste(img, patch, cy, cx, radius):
    """attack/DSGN/patch_attack.py:326-333,369-376: zero-pad the patch to the image size
    with its centre at (cy, cx), then ``(1 - M) * img + M * P`` over the WHOLE image.
    img [1,3,H,W], patch [1,3,D,D] with D = 2*radius+1.  Returns a new array."""
    img, patch = _f32(img), _f32(patch)
    _, _, h, w = img.shape
    d = 2 * radius + 1
    assert patch.shape[-2:] == (d, d)
    assert cy - radius >= 0 and cx - radius >= 0 and cy + radius < h and cx + radius < w
    P = np.zeros_like(img)
    P[:, :, cy - radius:cy + radius + 1, cx - radius:cx + radius + 1] = patch
    M = disc_mask(h, w, cy, cx, radius)[None, None]
    with np.errstate(invalid="ignore"):
        return (F32(1) - M) * img + M * P


# ----------------------------------------------------------------------------- a11 / a12
def patch_delta(grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha=1e3):
    """attack/DSGN/patch_attack.py:416-430 up to the clamp: the bounding-SQUARE windows
    of the two gradients, ``clamp(0.5 * alpha * (gL + gR), -eps, eps)``.  ``0.5 * alpha``
    is a Python float product (500.0) applied as one float32 scalar."""
    gl, gr = _f32(grad_l), _f32(grad_r)
    wl = gl[:, :, cy - radius:cy + radius + 1, cx_l - radius:cx_l + radius + 1]
    wr = gr[:, :, cy - radius:cy + radius + 1, cx_r - radius:cx_r + radius + 1]
    with np.errstate(invalid="ignore", over="ignore"):
        return torch_clamp(F32(0.5 * alpha) * (wl + wr), -eps, eps)


def patch_update(patch, grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha=1e3, lo=None, hi=None):
    """``patch -= delta`` (attack/DSGN/patch_attack.py:427-430); with ``lo``/``hi`` the
    Stereo R-CNN per-channel range clamp follows (attack/Stereo-RCNN/patch_attack.py:268-281)."""
    out = _f32(patch) - patch_delta(grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha)
    if lo is not None:
        for c in range(3):
            out[:, c] = torch_clamp(out[:, c], lo[c], hi[c])
    return out


def patch_apply_delta(patch, delta, lo=None, hi=None):
    """``patch - delta`` then the optional per-channel clamp: the second half of
    patch_update, split off so that a summed (multi-image / all-reduced) delta can be applied."""
    out = _f32(patch) - _f32(delta)
    if lo is not None:
        for c in range(3):
            out[:, c] = torch_clamp(out[:, c], lo[c], hi[c])
    return out


# ----------------------------------------------------------------------------- a16
def kitti_label_line(cls, bbox, score, corners, dims):
    """One line of attack/DSGN/predict_and_save_pgd.py:250-284.

    cls 1/2/other -> Pedestrian/Car/Cyclist (:273); corners [8,3] float32 give the box
    centre (float32 mean, :262); ``dims`` = (h, w, l, ry) is what upstream DSGN's
    ``get_dimensions`` returns for the centred corners (UNPINNED - not in the reference
    tree, supplied by the caller); alpha = -atan2(x, z) + ry; y is shifted by h/2 (:280).
    """
    corners = np.asarray(corners, dtype=np.float32).reshape(8, 3)
    # torch-CPU's float32 mean over dim 0 of an [8,3] tensor keeps four running sums over rows
    # j, j+4 and folds them left to right (found by search, pinned by the golden label text)
    x = corners
    center = ((((x[0] + x[4]) + (x[1] + x[5])) + (x[2] + x[6])) + (x[3] + x[7])) / F32(8)
    h, w, l, ry = dims
    name = "Pedestrian" if cls == 1 else "Car" if cls == 2 else "Cyclist"
    alpha = -np.arctan2(center[0], center[2]) + ry
    y = center[1] + F32(h / 2.0)
    return ("{} -1 -1 {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.8f}\n"
            .format(name, alpha, bbox[0], bbox[1], bbox[2], bbox[3], h, w, l,
                    center[0], y, center[2], ry, score))


# ----------------------------------------------------------------------------- K7
def psv_build(left, right, shift):
    """Plane-sweep concatenation cost volume of a DSGN-style detector (reached through the model call,
    attack/DSGN/pgd_attack.py:308).  UNPINNED against upstream DSGN (its CUDA op is not in the reference
    tree); this restates the published construction: per depth plane d one integer disparity shift s,
    cost[b, :C, d, :, s:] = left[..., s:], cost[b, C:, d, :, s:] = right[..., :W-s], zero elsewhere."""
    left, right = _f32(left), _f32(right)
    b, c, h, w = left.shape
    d = shift.shape[1]
    cost = np.zeros((b, 2 * c, d, h, w), np.float32)
    for bi in range(b):
        for di in range(d):
            s = max(0, min(int(shift[bi, di]), w))
            cost[bi, :c, di, :, s:] = left[bi, :, :, s:]
            cost[bi, c:, di, :, s:] = right[bi, :, :, :w - s]
    return cost


def psv_build_bwd(grad_cost, shift):
    """Exact adjoint of psv_build; the sum over planes runs d = 0, 1, ... in float32."""
    g = _f32(grad_cost)
    b, c2, d, h, w = g.shape
    c = c2 // 2
    gl = np.zeros((b, c, h, w), np.float32)
    gr = np.zeros((b, c, h, w), np.float32)
    for bi in range(b):
        for di in range(d):
            s = max(0, min(int(shift[bi, di]), w))
            tl = np.zeros((c, h, w), np.float32)
            tr = np.zeros((c, h, w), np.float32)
            tl[:, :, s:] = g[bi, :c, di, :, s:]
            tr[:, :, :w - s] = g[bi, c:, di, :, s:]
            if di == 0:
                gl[bi], gr[bi] = tl, tr
            else:
                gl[bi] += tl
                gr[bi] += tr
    return gl, gr


def _lerp_plane(sf, w):
    sf = F32(sf)
    sf = F32(0) if not (sf >= 0) else sf
    sf = F32(w) if sf > F32(w) else sf
    s0 = int(np.floor(sf))
    w1 = F32(sf - F32(s0))
    return s0, (s0 + 1 if w1 > 0 else s0), F32(F32(1) - w1), w1


def psv_build_lerp(left, right, shift):
    """Interpolating form of psv_build (csrc/psv.hip, LERP): float shift sf per plane; for x >= ceil(sf)
    cost_r[x] = w0 * right[x - s0] + w1 * right[x - s0 - 1] (zero-extended), s0 = floor(sf), w1 = sf - s0.  UNPINNED."""
    left, right = _f32(left), _f32(right)
    b, c, h, w = left.shape
    d = shift.shape[1]
    cost = np.zeros((b, 2 * c, d, h, w), np.float32)
    for bi in range(b):
        rz = np.concatenate([np.zeros((c, h, w + 1), np.float32), right[bi]], axis=2)       # rz[..., w + 1 + i] = right[i]
        for di in range(d):
            s0, sc, w0, w1 = _lerp_plane(shift[bi, di], w)
            x = np.arange(sc, w)
            if x.size == 0:
                continue
            cost[bi, :c, di, :, sc:] = left[bi, :, :, sc:]
            a = rz[:, :, w + 1 + x - s0]
            bb = rz[:, :, w + 1 + x - s0 - 1]
            cost[bi, c:, di, :, sc:] = F32(w0) * a + F32(w1) * bb
    return cost


def psv_build_lerp_bwd(grad_cost, shift):
    """exact adjoint of psv_build_lerp: per plane w0 * g[x + s0] + w1 * g[x + s0 + 1] over the unmasked outputs, planes
    summed d = 0, 1, ... in float32"""
    g = _f32(grad_cost)
    b, c2, d, h, w = g.shape
    c = c2 // 2
    gl = np.zeros((b, c, h, w), np.float32)
    gr = np.zeros((b, c, h, w), np.float32)
    for bi in range(b):
        for di in range(d):
            s0, sc, w0, w1 = _lerp_plane(shift[bi, di], w)
            tl = np.zeros((c, h, w), np.float32)
            tl[:, :, sc:] = g[bi, :c, di, :, sc:]
            gm = np.zeros((c, h, 2 * w + 2), np.float32)                                    # masked, zero-extended gradient row
            gm[:, :, sc:w] = g[bi, c:, di, :, sc:]
            x = np.arange(w)
            tr = F32(w0) * gm[:, :, x + s0] + F32(w1) * gm[:, :, x + s0 + 1]
            if di == 0:
                gl[bi], gr[bi] = tl, tr
            else:
                gl[bi] += tl
                gr[bi] += tr
    return gl, gr


# ----------------------------------------------------------------------------- RoI path (8f row 3)
def _roi_taps(height, width, y, x):
    """bilinear_interpolate pre-computation of maskrcnn-benchmark's ROIAlign (float32, same op order as the
    kernel).  UNPINNED against the upstream Stereo R-CNN extension (not in the reference tree)."""
    valid = not (y < F32(-1.0) or y > F32(height) or x < F32(-1.0) or x > F32(width))
    y = F32(0) if y <= 0 else y
    x = F32(0) if x <= 0 else x
    y_low, x_low = int(y), int(x)
    if y_low >= height - 1:
        y_high = y_low = height - 1
        y = F32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= width - 1:
        x_high = x_low = width - 1
        x = F32(x_low)
    else:
        x_high = x_low + 1
    ly, lx = F32(y - F32(y_low)), F32(x - F32(x_low))
    hy, hx = F32(F32(1) - ly), F32(F32(1) - lx)
    return valid, y_low, x_low, y_high, x_high, F32(hy * hx), F32(hy * lx), F32(ly * hx), F32(ly * lx)


def _roi_bins(roi, scale, ph, pw, sampling_ratio):
    scale = F32(scale)
    sw, sh = F32(roi[1] * scale), F32(roi[2] * scale)
    ew, eh = F32(roi[3] * scale), F32(roi[4] * scale)
    rw, rh = max(F32(ew - sw), F32(1)), max(F32(eh - sh), F32(1))
    bh, bw = F32(rh / F32(ph)), F32(rw / F32(pw))
    gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(F32(rh / F32(ph))))
    gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(F32(rw / F32(pw))))
    return int(roi[0]), sh, sw, bh, bw, gh, gw


def _roi_samples(roi, scale, ph, pw, sampling_ratio, height, width):
    """yield (ph_i, pw_i, count, taps) for every sample of every bin"""
    b, sh, sw, bh, bw, gh, gw = _roi_bins(roi, scale, ph, pw, sampling_ratio)
    for i in range(ph):
        for j in range(pw):
            for iy in range(gh):
                y = F32(F32(sh + F32(F32(i) * bh)) + F32(F32(F32(F32(iy) + F32(0.5)) * bh) / F32(gh)))
                for ix in range(gw):
                    x = F32(F32(sw + F32(F32(j) * bw)) + F32(F32(F32(F32(ix) + F32(0.5)) * bw) / F32(gw)))
                    yield i, j, F32(gh * gw), _roi_taps(height, width, y, x)
    return b


def roi_align(feat, rois, pooled, spatial_scale, sampling_ratio=0):
    feat, rois = _f32(feat), _f32(rois)
    _, c, h, w = feat.shape
    ph, pw = (pooled, pooled) if isinstance(pooled, int) else pooled
    out = np.zeros((rois.shape[0], c, ph, pw), np.float32)
    for r, roi in enumerate(rois):
        if roi[0] < 0:                 # a negative batch index: the roi is skipped (the kernel leaves its rows untouched; zeros here)
            continue
        plane = feat[int(roi[0])]
        count = None
        for i, j, count, (valid, yl, xl, yh, xh, w1, w2, w3, w4) in _roi_samples(roi, spatial_scale, ph, pw, sampling_ratio, h, w):
            if valid:
                v = ((w1 * plane[:, yl, xl] + w2 * plane[:, yl, xh]) + w3 * plane[:, yh, xl]) + w4 * plane[:, yh, xh]
            else:
                v = np.zeros(c, np.float32)
            out[r, :, i, j] = out[r, :, i, j] + v
        if count is not None:
            out[r] = out[r] / count
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale, sampling_ratio=0):
    g, rois = _f32(grad_out), _f32(rois)
    b, c, h, w = feat_shape
    ph, pw = g.shape[2:]
    gf = np.zeros((b, c, h, w), np.float64)          # float64 accumulation: the kernel's atomic order is not fixed
    for r, roi in enumerate(rois):
        bi = int(roi[0])
        for i, j, count, (valid, yl, xl, yh, xh, w1, w2, w3, w4) in _roi_samples(roi, spatial_scale, ph, pw, sampling_ratio, h, w):
            if valid:
                gv = g[r, :, i, j]
                gf[bi, :, yl, xl] += F32(gv * w1) / count
                gf[bi, :, yl, xh] += F32(gv * w2) / count
                gf[bi, :, yh, xl] += F32(gv * w3) / count
                gf[bi, :, yh, xh] += F32(gv * w4) / count
    return gf.astype(np.float32)


def roi_align_bwd_ordered(grad_out, rois, feat_shape, spatial_scale, sampling_ratio=0, segments=1):
    """The backward as csrc/roi.hip's gather kernel sums it: float32, contributions added per feature element in the order
    roi index, sample row (ph, iy), sample column (pw, ix), tap 1..4 - each contribution rounded as (g * w) / count.
    ``segments`` = G > 1 (the kernel's rule for more than 1024 rois, include/advengine.h: G = min(8, ceil(R / 512))): the roi indices are cut
    into G runs of ceil(R / G) consecutive indices, each run summed as above into its own map, the maps added in run order."""
    g, rois = _f32(grad_out), _f32(rois)
    if segments > 1:
        n = len(rois)
        step = (n + segments - 1) // segments
        total = np.zeros(feat_shape, np.float32)
        for s in range(segments):
            lo, hi = s * step, min(n, (s + 1) * step)
            if lo < hi:
                total = total + roi_align_bwd_ordered(g[lo:hi], rois[lo:hi], feat_shape, spatial_scale, sampling_ratio)
        return total
    b, c, h, w = feat_shape
    ph, pw = g.shape[2:]
    gf = np.zeros((b, c, h, w), np.float32)
    for r, roi in enumerate(rois):
        if roi[0] < 0:                 # skipped roi (negative batch index): belongs to no image's tile lists
            continue
        bi, sh, sw, bh, bw, gh, gw = _roi_bins(roi, spatial_scale, ph, pw, sampling_ratio)
        count = F32(gh * gw)
        for i in range(ph):
            for iy in range(gh):
                y = F32(F32(sh + F32(F32(i) * bh)) + F32(F32(F32(F32(iy) + F32(0.5)) * bh) / F32(gh)))
                for j in range(pw):
                    gv = g[r, :, i, j]
                    for ix in range(gw):
                        x = F32(F32(sw + F32(F32(j) * bw)) + F32(F32(F32(F32(ix) + F32(0.5)) * bw) / F32(gw)))
                        valid, yl, xl, yh, xh, w1, w2, w3, w4 = _roi_taps(h, w, y, x)
                        if not valid:
                            continue
                        for (yy, xx, wt) in ((yl, xl, w1), (yl, xh, w2), (yh, xl, w3), (yh, xh, w4)):
                            gf[bi, :, yy, xx] = gf[bi, :, yy, xx] + F32(gv * wt) / count
    return gf


def nms(boxes, thresh):
    """Greedy NMS, boxes [N,4] pre-sorted by descending score, legacy +1 areas, float32 IoU in the kernel's
    op order; returns kept indices."""
    bx = _f32(boxes)
    n = bx.shape[0]
    removed = np.zeros(n, bool)
    keep = []
    area = (bx[:, 2] - bx[:, 0] + F32(1)) * (bx[:, 3] - bx[:, 1] + F32(1))
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            rest = bx[i + 1:]
            left, right = np.maximum(bx[i, 0], rest[:, 0]), np.minimum(bx[i, 2], rest[:, 2])
            top, bottom = np.maximum(bx[i, 1], rest[:, 1]), np.minimum(bx[i, 3], rest[:, 3])
            ww = np.maximum(right - left + F32(1), F32(0))
            hh = np.maximum(bottom - top + F32(1), F32(0))
            inter = ww * hh
            iou = inter / (area[i] + area[i + 1:] - inter)
            removed[i + 1:] |= iou > F32(thresh)
    return np.array(keep, dtype=np.int64)


# ---------------------------------------------------------------------------------------------------------------
# csrc/volume.hip: fused depth regression, grid_sample on 5-D volumes, sigmoid focal loss.  The ops are upstream DSGN /
# maskrcnn-benchmark code (reached at attack/DSGN/pgd_attack.py:308,324), NOT in the reference tree: these restate the
# KERNELS' arithmetic in float32 and are themselves checked against torch's F.interpolate/softmax, F.grid_sample and an
# autograd formulation of the focal loss in tests/test_volume.py (CPU).
def bev_fold(v, pool):
    """csrc/volume.hip bev_fold_fwd: [B,C,Z,Y,X] -> [B, C * (Y // pool), Z, X], the float32 sum over ``pool`` consecutive rows in ascending
    order divided by pool - the values of F.avg_pool3d(v, (1, pool, 1)).permute(0, 1, 3, 2, 4).reshape(...)"""
    v = _f32(v)
    b, c, z, y, x = v.shape
    yp = y // pool
    out = np.empty((b, c, yp, z, x), np.float32)
    for yy in range(yp):
        acc = v[:, :, :, pool * yy, :].copy()
        for k in range(1, pool):
            acc = acc + v[:, :, :, pool * yy + k, :]
        out[:, :, yy] = acc / F32(pool)
    return out.reshape(b, c * yp, z, x)


def bev_fold_bwd(grad_out, shape, pool, mask=None):
    if mask is not None:
        return np.where(_f32(mask) > 0, bev_fold_bwd(grad_out, shape, pool), F32(0)).astype(np.float32)
    g = _f32(grad_out)
    b, c, z, y, x = shape
    yp = y // pool
    gv = np.zeros(shape, np.float32)
    gg = g.reshape(b, c, yp, z, x) / F32(pool)
    for yy in range(yp):
        for k in range(pool):
            gv[:, :, :, pool * yy + k, :] = gg[:, :, yy]
    return gv


def _up_taps(n_in, n_out):
    """csrc/resize.hip source_of: per output index (i0, i1, l0, l1) - torch's align_corners=False source coordinate in float32"""
    sc = np.float32(n_in) / np.float32(n_out)
    o = np.arange(n_out, dtype=np.float32)
    # fmaf(scale, o + 0.5, -0.5): the product of two float32 is exact in float64 and so is the sum here (< 2^12, bits down to 2^-37): one rounding
    src = np.maximum((sc.astype(np.float64) * (o + F32(0.5)).astype(np.float64) - 0.5).astype(np.float32), F32(0)).astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (F32(1) - l1).astype(np.float32), l1


def bilinear_up(x, size):
    """csrc/resize.hip bilinear_up_fwd (F.interpolate(x, size, mode="bilinear", align_corners=False); the FPN's _upsample_add,
    attack/Stereo-RCNN/stereo_rcnn.py:92-108): x [..., h, w] -> [..., ho, wo], every product and sum rounded on its own.  PINNED by
    tests/golden/upsample_add.npz (the reference's method executed, with its backward: tests/test_resize.py)"""
    x = _f32(x)
    h, w = x.shape[-2:]
    y0, y1, ly0, ly1 = _up_taps(h, size[0])
    x0, x1, lx0, lx1 = _up_taps(w, size[1])
    top = (lx0 * x[..., y0, :][..., x0]).astype(np.float32) + (lx1 * x[..., y0, :][..., x1]).astype(np.float32)
    bot = (lx0 * x[..., y1, :][..., x0]).astype(np.float32) + (lx1 * x[..., y1, :][..., x1]).astype(np.float32)
    return ((ly0[:, None] * top).astype(np.float32) + (ly1[:, None] * bot).astype(np.float32)).astype(np.float32)


def bilinear_up_bwd(grad_out, in_hw):
    """csrc/resize.hip bilinear_up_bwd, the adjoint as a gather: grad_in[iy][ix] = the sum over output rows ascending, columns ascending of
    (wy * wx) * g with w(o, i) = (i0(o) == i ? l0 : 0) + (i1(o) == i ? l1 : 0), zero weights skipped.  Walking the outputs in raster
    order and adding each to the pixels it reads visits every input pixel's terms in exactly that order."""
    g = _f32(grad_out)
    h, w = in_hw
    ho, wo = g.shape[-2:]
    lead = g.shape[:-2]
    gin = np.zeros(lead + (h, w), np.float32)

    def weights(n_in, n_out):
        i0, i1, l0, l1 = _up_taps(n_in, n_out)
        out = []
        for o in range(n_out):
            if i0[o] == i1[o]:
                out.append([(int(i0[o]), np.float32(l0[o] + l1[o]))])
            else:
                out.append([(int(i0[o]), l0[o]), (int(i1[o]), l1[o])])
        return out
    wys, wxs = weights(h, ho), weights(w, wo)
    for oy in range(ho):
        for ox in range(wo):
            for iy, wy in wys[oy]:
                if wy == 0:
                    continue
                for ix, wx in wxs[ox]:
                    if wx == 0:
                        continue
                    gin[..., iy, ix] = gin[..., iy, ix] + (np.float32(wy * wx) * g[..., oy, ox]).astype(np.float32)
    return gin


def _lin_scale(n_in, n_out, align):
    if align:
        return np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    return np.float32(n_in) / np.float32(n_out)


def _lin_taps(n_in, n_out, align):
    """torch's linear source index per output index: (i0, i1, l0, l1) as arrays, float32 arithmetic"""
    sc = _lin_scale(n_in, n_out, align)
    dst = np.arange(n_out, dtype=np.float32)
    src = sc * dst if align else np.maximum(sc * (dst + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    src = src.astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
    return i0, i1, (np.float32(1) - l1).astype(np.float32), l1


def trilinear_upsample(cost, out_size, align=False):
    """[B,D,h,w] -> [B,Do,H,W] in the kernel's nesting (w innermost, then h, then d), float32"""
    cost = np.asarray(cost, np.float32)
    do, ho, wo = out_size
    x0, x1, a0, a1 = _lin_taps(cost.shape[3], wo, align)
    v = a0 * cost[..., x0] + a1 * cost[..., x1]
    y0, y1, b0, b1 = _lin_taps(cost.shape[2], ho, align)
    v = b0[:, None] * v[:, :, y0, :] + b1[:, None] * v[:, :, y1, :]
    d0, d1, c0, c1 = _lin_taps(cost.shape[1], do, align)
    return (c0[:, None, None] * v[:, d0] + c1[:, None, None] * v[:, d1]).astype(np.float32)


def depth_regress(cost, depth_values, out_size, align=False):
    """-> depth [B,H,W], stats [B,2,H,W] (softmax max and denominator); plane-ordered float32 sums as the kernel"""
    up = trilinear_upsample(cost, out_size, align)
    zv = np.asarray(depth_values, np.float32)
    m = up.max(axis=1)
    s = np.zeros_like(m)
    e = np.zeros_like(m)
    for k in range(up.shape[1]):
        p = np.exp(up[:, k] - m).astype(np.float32)
        s = (s + p).astype(np.float32)
        e = (e + p * zv[k]).astype(np.float32)
    return (e / s).astype(np.float32), np.stack([m, s], 1)


def depth_regress_bwd(cost, depth_values, grad_depth, out_size, align=False):
    """float64 gradient of depth_regress w.r.t. cost (the kernel's float32 result is compared with a tolerance)"""
    cost = np.asarray(cost, np.float64)
    do, ho, wo = out_size
    taps = [_lin_taps(cost.shape[1], do, align), _lin_taps(cost.shape[2], ho, align), _lin_taps(cost.shape[3], wo, align)]
    mats = []
    for (i0, i1, l0, l1), n_in in zip(taps, cost.shape[1:]):
        m = np.zeros((len(i0), n_in))
        m[np.arange(len(i0)), i0] += l0
        m[np.arange(len(i0)), i1] += l1
        mats.append(m)
    up = np.einsum("kd,yh,xw,bdhw->bkyx", mats[0], mats[1], mats[2], cost)
    p = np.exp(up - up.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    zv = np.asarray(depth_values, np.float64)[None, :, None, None]
    depth = (p * zv).sum(1, keepdims=True)
    gup = np.asarray(grad_depth, np.float64)[:, None] * p * (zv - depth)
    return np.einsum("kd,yh,xw,bkyx->bdhw", mats[0], mats[1], mats[2], gup)


def _gs_corners(grid, dims, align):
    """aten grid_sampler_3d: per output voxel the eight (cell offset or -1, weight), float32"""
    d, h, w = dims
    g = np.asarray(grid, np.float32)

    def unnorm(c, size):
        if align:
            return ((c + np.float32(1)) / np.float32(2)) * np.float32(size - 1)
        return ((c + np.float32(1)) * np.float32(size) - np.float32(1)) / np.float32(2)

    ix, iy, iz = unnorm(g[..., 0], w), unnorm(g[..., 1], h), unnorm(g[..., 2], d)
    fx, fy, fz = np.floor(ix), np.floor(iy), np.floor(iz)
    ax, bx = fx + 1 - ix, ix - fx
    ay, by = fy + 1 - iy, iy - fy
    az, bz = fz + 1 - iz, iz - fz
    offs, wgts = [], []
    for k in range(8):
        xx = fx + (k & 1)
        yy = fy + ((k >> 1) & 1)
        zz = fz + ((k >> 2) & 1)
        wk = ((bx if k & 1 else ax) * (by if k & 2 else ay)).astype(np.float32) * (bz if k & 4 else az)
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1) & (zz >= 0) & (zz <= d - 1)
        off = np.where(ok, (np.where(ok, zz, 0) * h + np.where(ok, yy, 0)) * w + np.where(ok, xx, 0), -1).astype(np.int64)
        offs.append(off)
        wgts.append(wk.astype(np.float32))
    return offs, wgts


def grid_sample3d(vol, grid, align=False):
    """vol [B,C,D,H,W], grid [B,Z,Y,X,3] -> [B,C,Z,Y,X]; corners added in aten's order, float32"""
    vol = np.asarray(vol, np.float32)
    b, c = vol.shape[:2]
    offs, wgts = _gs_corners(grid, vol.shape[2:], align)
    flat = vol.reshape(b, c, -1)
    out = np.zeros((b, c) + offs[0].shape[1:], np.float32)
    for k in range(8):
        for i in range(b):
            ok = offs[k][i] >= 0
            val = flat[i][:, np.where(ok, offs[k][i], 0)] * wgts[k][i][None]
            out[i] = np.where(ok[None], (out[i] + val).astype(np.float32), out[i])
    return out


def grid_sample3d_bwd(grad_out, grid, vol_dims, align=False):
    """gradient w.r.t. vol in the gather kernel's order: per cell, contributions by ascending output voxel, float32"""
    go = np.asarray(grad_out, np.float32)
    b, c = go.shape[:2]
    d, h, w = vol_dims
    offs, wgts = _gs_corners(grid, vol_dims, align)
    gv = np.zeros((b, c, d * h * w), np.float32)
    gof = go.reshape(b, c, -1)
    for i in range(b):
        cell = np.stack([o[i].reshape(-1) for o in offs], 1)            # [N,8]
        wk = np.stack([x[i].reshape(-1) for x in wgts], 1)
        n = cell.shape[0]
        vox = np.repeat(np.arange(n), 8)
        cell, wk = cell.reshape(-1), wk.reshape(-1)
        keep = cell >= 0
        cell, wk, vox = cell[keep], wk[keep], vox[keep]
        order = np.lexsort((vox, cell))                                 # by cell, then by output voxel
        cell, wk, vox = cell[order], wk[order], vox[order]
        rank = np.arange(len(cell)) - np.searchsorted(cell, cell, side="left")
        for r in range(int(rank.max()) + 1 if len(rank) else 0):        # r-th entry of every list: one vectorised float32 add
            sel = rank == r
            gv[i][:, cell[sel]] = (gv[i][:, cell[sel]] + gof[i][:, vox[sel]] * wk[sel][None]).astype(np.float32)
    return gv.reshape(b, c, d, h, w)


def sigmoid_focal_loss(logits, targets, gamma=2.0, alpha=0.25):
    """-> (loss [N,K], dloss/dlogit [N,K]), float64 (the kernel's float32 is compared with a tolerance)"""
    x = np.asarray(logits, np.float64)
    t = np.asarray(targets).reshape(-1, 1)
    cls = np.arange(1, x.shape[1] + 1)[None]
    p = 1 / (1 + np.exp(-x))
    lp = np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))
    lq = np.minimum(-x, 0) - np.log1p(np.exp(-np.abs(x)))
    pos, neg = t == cls, (t >= 0) & (t != cls)
    loss = np.where(pos, -alpha * (1 - p) ** gamma * lp, 0) + np.where(neg, -(1 - alpha) * p ** gamma * lq, 0)
    grad = np.where(pos, alpha * (1 - p) ** gamma * (gamma * p * lp - (1 - p)), 0) + \
        np.where(neg, (1 - alpha) * p ** gamma * (p - gamma * (1 - p) * lq), 0)
    return loss, grad


def stem_pool(t, bias=None):
    """y = maxpool(3x3, stride 2, padding 1)(relu(t + bias)) and the code bytes of csrc/volume.hip:stem_pool_fwd - the tail of the ResNet
    stem (upstream: ``self.relu`` + ``self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)`` behind conv1 / bn1 of the
    fpn.pytorch-family resnet.py the Stereo R-CNN backbone is built from [UPSTREAM-UNVERIFIED path]).  Window scanned row-major, a later
    element wins only if greater: torch's argmax; code 15 where the maximum is <= 0 (relu passes no gradient).  Test infrastructure."""
    t = np.asarray(t, np.float32)
    b, c, h, w = t.shape
    v = t if bias is None else (t + np.asarray(bias, np.float32)[None, :, None, None]).astype(np.float32)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    pad = np.full((b, c, 2 * oh + 1, 2 * ow + 1), -np.inf, np.float32)
    pad[:, :, 1:h + 1, 1:w + 1] = v
    stack = np.stack([pad[:, :, ky:ky + 2 * oh:2, kx:kx + 2 * ow:2] for ky in range(3) for kx in range(3)], 0)
    arg = np.argmax(stack, 0)                     # the first maximum in row-major window order
    m = np.take_along_axis(stack, arg[None], 0)[0]
    keep = m > 0
    return np.where(keep, m, np.float32(0)).astype(np.float32), np.where(keep, arg, 15).astype(np.uint8)


def stem_pool_bwd(grad_y, code, in_hw):
    """the gradient w.r.t. t: every pixel sums, in (oy, ox) order, the outputs whose code points at it"""
    g = np.asarray(grad_y, np.float32)
    b, c, oh, ow = g.shape
    h, w = in_hw
    out = np.zeros((b, c, h + 2, w + 3), np.float32)          # padded by one on the low side (window offset -1), room on the high side
    for oy in range(oh):                                       # ascending (oy, ox): float32 sums in the kernel's order
        for ox in range(ow):
            k = code[:, :, oy, ox]
            live = k < 9
            ky, kx = k // 3, k % 3
            bi, ci = np.nonzero(live)
            out[bi, ci, 2 * oy + ky[live], 2 * ox + kx[live]] += g[bi, ci, oy, ox]
    return out[:, :, 1:h + 1, 1:w + 1].copy()


def box_iou(a, b):
    """[N,4] x [M,4] -> [N,M] with the legacy +1 widths (upstream lib/model/rpn/bbox_transform.py:bbox_overlaps [UPSTREAM-UNVERIFIED path];
    this package: surrogates._iou, csrc/boxes.hip:iou_of), float32 operation by operation.  Test infrastructure."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    one, zero = np.float32(1), np.float32(0)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.maximum(rb - lt + one, zero)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0] + one) * (a[:, 3] - a[:, 1] + one)
    area_b = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    return (inter / (area_a[:, None] + area_b[None, :] - inter)).astype(np.float32)


def box_encode(src, dst):
    """(dx, dy, log dw, log dh) that move ``src`` onto ``dst`` (bbox_transform; surrogates._encode).  np.log is within an ulp of the device's."""
    s, d = np.asarray(src, np.float32), np.asarray(dst, np.float32)
    one, half = np.float32(1), np.float32(0.5)
    sw, sh = s[:, 2] - s[:, 0] + one, s[:, 3] - s[:, 1] + one
    sx, sy = s[:, 0] + half * sw, s[:, 1] + half * sh
    dw, dh = d[:, 2] - d[:, 0] + one, d[:, 3] - d[:, 1] + one
    dx, dy = d[:, 0] + half * dw, d[:, 1] + half * dh
    return np.stack([(dx - sx) / sw, (dy - sy) / sh, np.log(dw / sw), np.log(dh / sh)], 1).astype(np.float32)


def box_decode_clip(src, d, width, height):
    """bbox_transform_inv (log-sizes clamped at 4) + clip_boxes (surrogates._decode and the clamps of its caller)"""
    s, d = np.asarray(src, np.float32), np.asarray(d, np.float32)
    one, half = np.float32(1), np.float32(0.5)
    sw, sh = s[:, 2] - s[:, 0] + one, s[:, 3] - s[:, 1] + one
    sx, sy = s[:, 0] + half * sw, s[:, 1] + half * sh
    cx, cy = d[:, 0] * sw + sx, d[:, 1] * sh + sy
    w, h = np.exp(np.minimum(d[:, 2], np.float32(4))) * sw, np.exp(np.minimum(d[:, 3], np.float32(4))) * sh
    out = np.stack([cx - half * w, cy - half * h, cx + half * w - one, cy + half * h - one], 1).astype(np.float32)
    out[:, 0::2] = np.clip(out[:, 0::2], 0, np.float32(width - 1))
    out[:, 1::2] = np.clip(out[:, 1::2], 0, np.float32(height - 1))
    return out
