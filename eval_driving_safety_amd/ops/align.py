"""dense photometric box alignment (csrc/align.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)

# --------------------------------------------------------------------------------------------
# dense photometric box alignment (attack/Stereo-RCNN/predict_and_save_pgd.py:381; upstream op, published algorithm)
def dense_align_cost(left, right, roi, dz, z_center, fb, step, k, out=None):
    """cost [n,k] of k candidate depths around z_center per object (adv_dense_align_cost_f32).  left/right [3,H,W],
    roi int32 [n,4] = (u0, v0, u1, v1), dz float32 [n,S] per-column depth offsets, z_center float32 [n] - all on the device."""
    l, r = _feat(left, "left"), _feat(right, "right")
    if l.dim() != 3 or l.shape[0] != 3 or l.shape != r.shape:
        raise ValueError("left/right must be [3,H,W] of equal shape")
    n = roi.shape[0]
    if not (roi.is_cuda and roi.dtype == torch.int32 and roi.is_contiguous() and tuple(roi.shape) == (n, 4)):
        raise TypeError("roi must be a contiguous int32 CUDA tensor [n,4]")
    dzt, zc = _feat(dz, "dz"), _feat(z_center, "z_center")
    if dzt.dim() != 2 or dzt.shape[0] != n or tuple(zc.shape) != (n,):
        raise ValueError("dz must be [n,S] and z_center [n]")
    if int((roi[:, 2] - roi[:, 0]).max()) > dzt.shape[1]:
        raise ValueError("a region is wider than the dz rows")
    out = torch.empty((n, k), dtype=torch.float32, device=l.device) if out is None else out
    with _on(l):
        _lib.call("adv_dense_align_cost_f32", _ptr(l), _ptr(r), l.shape[1], l.shape[2], n, _ptr(roi), _ptr(dzt), dzt.shape[1], _ptr(zc),
                  float(fb), float(step), int(k), _ptr(out), _stream(l))
    return out


def dense_align_argmin(cost, z_center, step):
    c, zc = _feat(cost, "cost"), _feat(z_center, "z_center")
    n, k = c.shape
    z = torch.empty((n,), dtype=torch.float32, device=c.device)
    cmin = torch.empty((n,), dtype=torch.float32, device=c.device)
    with _on(c):
        _lib.call("adv_dense_align_argmin_f32", _ptr(c), n, k, _ptr(zc), float(step), _ptr(z), _ptr(cmin), _stream(c))
    return z, cmin


def dense_align_search(left, right, roi, dz, z0, fb, coarse=(50, 0.5), fine=(20, 0.05)):
    """coarse-to-fine enumeration of the published algorithm: 50 depths 0.5 m apart around z0, then 20 depths 0.05 m apart
    around the best of those.  Four launches, no host round trip.  -> (z [n], cost [n])"""
    c0 = dense_align_cost(left, right, roi, dz, z0, fb, coarse[1], coarse[0])
    z1, _ = dense_align_argmin(c0, z0, coarse[1])
    c1 = dense_align_cost(left, right, roi, dz, z1, fb, fine[1], fine[0])
    return dense_align_argmin(c1, z1, fine[1])


def box_depth_offsets(u_cols, f, cx, x, z, width, length, theta):
    """depth of the visible surface of an upright box (footprint: centre (x, z), ``length`` along the heading ``theta``,
    ``width`` across) along the camera rays through the image columns ``u_cols`` (original-image pixels), minus the centre
    depth: the dz(u) of the alignment.  Columns whose ray misses the footprint get 0.  numpy, host side."""
    import numpy as np
    hx, hz = np.cos(theta), -np.sin(theta)                       # heading in the (x, z) ground plane (KITTI rotation_y)
    px, pz = -hz, hx
    corners = [(x + sl * hx * length / 2 + sw * px * width / 2, z + sl * hz * length / 2 + sw * pz * width / 2)
               for sl, sw in ((1, 1), (1, -1), (-1, -1), (-1, 1))]
    d = (np.asarray(u_cols, np.float64) - cx) / f                # ray: (t * d, t)
    best = np.full(d.shape, np.inf)
    for i in range(4):
        (ax, az), (bx, bz) = corners[i], corners[(i + 1) % 4]
        ex, ez = bx - ax, bz - az
        den = d * ez - ex                                         # solve t*d = ax + s*ex, t = az + s*ez
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (ax - d * az) / den
            t = az + s * ez
        ok = (np.abs(den) > 1e-12) & (s >= 0) & (s <= 1) & (t > 0)
        best = np.where(ok & (t < best), t, best)
    return np.where(np.isfinite(best), best - z, 0.0).astype(np.float32)


def dense_align(calib, scale, im_left, im_right, boxes, kpts, poses):
    """Same call as the upstream ``dense_align.align_parallel`` (predict_and_save_pgd.py:381-384): boxes [n,4] left boxes
    and kpts [n,>=5] (columns 3, 4 = the object's left / right border) in ORIGINAL-image pixels, poses [n,7] =
    (x, y, z, dim0, dim1, dim2, theta), the image pair [1,3,H,W] at network scale (= original * scale).
    Valid region: the lower half of the left box between the two borders.  -> (succ [n], disparity [n] in original pixels).
    The region / dz model and the search follow the published algorithm; UNPINNED against the upstream module."""
    import numpy as np
    n = boxes.shape[0]
    dev = im_left.device
    scale = float(scale)
    p2, p3 = np.asarray(calib.p2, np.float64), np.asarray(calib.p3, np.float64)
    f, cx = p2[0, 0], p2[0, 2]
    bl = (p2[0, 3] - p3[0, 3]) / f
    left = im_left[0] if im_left.dim() == 4 else im_left
    right = im_right[0] if im_right.dim() == 4 else im_right
    H, W = left.shape[1], left.shape[2]
    bx, kp, ps = boxes.detach().cpu().numpy(), kpts.detach().cpu().numpy(), poses.detach().cpu().numpy()
    rois, rows = [], []
    for i in range(n):
        x1, y1, x2, y2 = bx[i, :4]
        lo, hi = max(x1, min(kp[i, 3], kp[i, 4])), min(x2, max(kp[i, 3], kp[i, 4]))
        if not hi > lo:
            lo, hi = x1, x2
        u0, u1 = int(np.floor(lo * scale)), int(np.ceil(hi * scale))
        v0, v1 = int(np.floor(0.5 * (y1 + y2) * scale)), int(np.ceil(y2 * scale))
        u0, u1, v0, v1 = max(u0, 0), min(u1, W), max(v0, 0), min(v1, H)
        rois.append([u0, v0, max(u1, u0), max(v1, v0)])
        cols = (np.arange(u0, max(u1, u0)) + 0.5) / scale
        rows.append(box_depth_offsets(cols, f, cx, ps[i, 0], ps[i, 2], ps[i, 4], ps[i, 5], ps[i, 6]))
    stride = max(1, max(len(r) for r in rows))
    dz = np.zeros((n, stride), np.float32)
    for i, r in enumerate(rows):
        dz[i, :len(r)] = r
    roi_t = torch.tensor(rois, dtype=torch.int32, device=dev)
    z0 = torch.tensor(ps[:, 2], dtype=torch.float32, device=dev).contiguous()
    z, cost = dense_align_search(left.contiguous(), right.contiguous(), roi_t, torch.from_numpy(dz).to(dev), z0, f * bl * scale)
    succ = torch.isfinite(cost) & (z > 0)
    disp = torch.where(succ, (f * bl) / z, torch.zeros_like(z))
    return succ.to(torch.int32), disp


__all__ = [n for n in dir() if not n.startswith("__")]
