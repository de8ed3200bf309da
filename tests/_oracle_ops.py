"""An ``ops``-shaped object backed by the numpy ORACLE, for CPU tests of the host logic only
(attack drivers, sharding, all-reduce rule).  It lives under tests/ on purpose: product code never
imports the oracle, and the drivers default to the HIP library (which has no CPU path)."""
import numpy as np
import torch

from oracle import oracle_np as O


class Space:
    def __init__(self, name):
        self.name = name
        self.affine = name == "dsgn"
        self.lo = (0.0, 0.0, 0.0) if self.affine else tuple(float(v) for v in O.SRCNN_LO)
        self.hi = (1.0, 1.0, 1.0) if self.affine else tuple(float(v) for v in O.SRCNN_HI)

    @staticmethod
    def dsgn():
        return Space("dsgn")

    @staticmethod
    def srcnn():
        return Space("srcnn")


def _np(t):
    return t.detach().cpu().numpy()


def _put(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view_as(dst) if dst.dim() != arr.ndim else torch.from_numpy(np.ascontiguousarray(arr)))
    return dst


def alloc_u8(n, crop_h, w, device):
    return torch.zeros((n, crop_h, w, 3), dtype=torch.uint8, device=device)


def denormalize(x, space, out=None):
    out = torch.empty_like(x) if out is None else out
    return _put(out, O.denormalize(_np(x)))


def normalize(x, space, out=None):
    out = torch.empty_like(x) if out is None else out
    return _put(out, O.normalize(_np(x)))


def _export(x_np, space, rows, cols, out):
    for i in range(x_np.shape[0]):
        if space.affine:
            full = O.tensor2im_u8(x_np[i], rows, x_np.shape[3])
        else:
            full = O.srcnn_export_u8(x_np[i])[:rows]
        out[i, :rows, :full.shape[1]] = torch.from_numpy(full)
    return out


def export_u8(x, space, crop=None, out=None):
    n, _, h, w = x.shape
    crop = (h, w) if crop is None else crop
    out = alloc_u8(n, crop[0], w, x.device) if out is None else out
    return _export(_np(x), space, crop[0], crop[1], out)


def pgd_step(x, grad, clean, space, alpha, eps, out=None, u8_out=None, crop=None):
    f = O.pgd_step_norm01 if space.affine else O.pgd_step_meansub255
    res = f(_np(x), _np(grad), _np(clean), alpha, eps)
    out = torch.empty_like(x) if out is None else out
    _put(out, res)
    if u8_out is not None:
        rows, cols = crop if crop is not None else x.shape[2:]
        _export(res, space, rows, cols, u8_out)
    return out


def patch_paste(img, patch, cy, cx, radius):
    return _put(img, O.patch_paste(_np(img).reshape((1,) + tuple(img.shape[-3:])), _np(patch).reshape((1,) + tuple(patch.shape[-3:])), cy, cx, radius))


def patch_paste_batch(img, patch, centers, radius):
    c = _np(centers)
    for i in range(img.shape[0]):
        patch_paste(img[i:i + 1], patch, int(c[i, 0]), int(c[i, 1]), radius)
    return img


def patch_update(patch, grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha=1e3, lo=None, hi=None, delta_out=None):
    p4 = _np(patch).reshape((1,) + tuple(patch.shape[-3:]))
    res = O.patch_update(p4, _np(grad_l), _np(grad_r), cy, cx_l, cx_r, radius, eps, alpha,
                         None if lo is None else [np.float32(v) for v in lo], None if hi is None else [np.float32(v) for v in hi])
    return _put(patch, res.reshape(tuple(patch.shape)))


def patch_delta_batch(grad_l, grad_r, centers, radius, eps, alpha=1e3, out=None):
    c = _np(centers)
    gl, gr = _np(grad_l), _np(grad_r)
    acc = None
    for i in range(gl.shape[0]):
        d = O.patch_delta(gl[i:i + 1], gr[i:i + 1], int(c[i, 0]), int(c[i, 1]), int(c[i, 2]), radius, eps, alpha)
        acc = d if acc is None else acc + d
    res = torch.from_numpy(np.ascontiguousarray(acc[0]))
    if out is not None:
        out.copy_(res)
        return out
    return res


def patch_apply(patch, delta, lo=None, hi=None):
    p4 = _np(patch).reshape((1,) + tuple(patch.shape[-3:]))
    d4 = _np(delta).reshape(p4.shape)
    res = O.patch_apply_delta(p4, d4, None if lo is None else [np.float32(v) for v in lo], None if hi is None else [np.float32(v) for v in hi])
    return _put(patch, res.reshape(tuple(patch.shape)))
