"""a1-a13: pixel-space maps, the PGD step, 8-bit import / export, disc mask, patch paste / update (csrc/advengine.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)

# --------------------------------------------------------------------------------------------
def denormalize(x, space, out=None):
    """attack/DSGN/pgd_attack.py:196-200 for every image of the batch; ``out=x`` is in place."""
    xi = _img(x, "x")
    out = torch.empty_like(x) if out is None else out
    oi = _img(out, "out")
    _same(xi, oi, "x", "out")
    n, _, h, w = xi.shape
    with _on(x):
        _lib.call("adv_denormalize_f32", _ptr(xi), _ptr(oi), n, h, w, space.ref(), _stream(x))
    return out


def normalize(x, space, out=None):
    """attack/DSGN/pgd_attack.py:203-207."""
    xi = _img(x, "x")
    out = torch.empty_like(x) if out is None else out
    oi = _img(out, "out")
    _same(xi, oi, "x", "out")
    n, _, h, w = xi.shape
    with _on(x):
        _lib.call("adv_normalize_f32", _ptr(xi), _ptr(oi), n, h, w, space.ref(), _stream(x))
    return out


class CleanIndex:
    """The clean image of an attack held as one byte per element (adv_clean_index_t): ``index`` uint8 [N,3,H,W],
    ``ok`` int32 [N] (non-zero = every element of that image verified on the device; read by the kernels, never by the
    host on the attack path), ``lut`` float32 [2,3,256] (two candidate tables, see include/advengine.h), ``valid`` = (valid_h, valid_w) for the whole batch or an
    int32 device tensor [N,2] with one such pair per image: outside that corner the loader's zero padding is
    expected (clean == shift_c)."""

    def __init__(self, index, ok, lut, valid):
        self.index, self.ok, self.lut, self.valid = index, ok, lut, valid
        c = _lib.AdvCleanIndex()
        c.index, c.ok, c.lut = index.data_ptr(), ok.data_ptr(), lut.data_ptr()
        if isinstance(valid, torch.Tensor):
            c.valid_hw, c.valid_h, c.valid_w = valid.data_ptr(), 0, 0
        else:
            c.valid_hw, c.valid_h, c.valid_w = None, int(valid[0]), int(valid[1])
        self.c = c

    def ref(self):
        return ctypes.byref(self.c)

    def verified(self):
        """host copy of the per-image flags (synchronises: tests / reporting only)"""
        return [bool(v) for v in self.ok.cpu().tolist()]


def can_index_clean(x, space):
    """affine spaces (DSGN): rows of whole pixel groups; identity spaces (Stereo R-CNN): planes of whole pixel groups"""
    xi = x if x.dim() == 4 else x.unsqueeze(0)
    if x.data_ptr() % 16 != 0:
        return False
    return xi.shape[3] % 4 == 0 if space.affine else (xi.shape[2] * xi.shape[3]) % 4 == 0


def _valid_arg(valid, n, h, w, device):
    """None -> the whole frame; (vh, vw) -> every image; a list of n (vh, vw) -> int32 device tensor [n,2]"""
    if valid is None:
        return (h, w)
    if isinstance(valid, torch.Tensor):
        if not (valid.is_cuda and valid.dtype == torch.int32 and valid.is_contiguous() and tuple(valid.shape) == (n, 2)):
            raise TypeError("valid must be a contiguous int32 CUDA tensor [%d,2]" % n)
        return valid
    valid = list(valid)
    if len(valid) == 2 and not isinstance(valid[0], (tuple, list)):
        vh, vw = int(valid[0]), int(valid[1])
        if not (0 <= vh <= h and 0 <= vw <= w):
            raise ValueError("valid corner %s exceeds the frame %s" % ((vh, vw), (h, w)))
        return (vh, vw)
    if len(valid) != n:
        raise ValueError("valid must hold one (valid_h, valid_w) per image (%d), got %d" % (n, len(valid)))
    if len(set(tuple(v) for v in valid)) == 1:
        return _valid_arg(tuple(valid[0]), n, h, w, device)
    for vh, vw in valid:
        if not (0 <= vh <= h and 0 <= vw <= w):
            raise ValueError("valid corner %s exceeds the frame %s" % ((vh, vw), (h, w)))
    return torch.tensor([[int(a), int(b)] for a, b in valid], dtype=torch.int32, device=device)


def denormalize_indexed(x, space, out=None, reuse=None, valid=None, u8_out=None, crop=None):
    """``denormalize`` (identity spaces: a copy of ``x``, the clean pair of attack/Stereo-RCNN/pgd_attack.py:122-123) plus the
    8-bit index of the result: returns (clean, CleanIndex).  Pass the CleanIndex to
    ``pgd_step(..., clean_index=...)``: every step then reads 1 byte instead of 4 for the clean image of every image
    whose device-side check succeeded (images that came from 8-bit pixels via ToTensor + Normalize, zero-padded
    beyond ``valid`` = (valid_h, valid_w) or one such pair per image), with identical results.
    ``u8_out``/``crop``: also write the 8-bit export of ``x`` (iterate 0), as ``export_u8`` would."""
    xi = _img(x, "x")
    out = torch.empty_like(x) if out is None else out
    oi = _img(out, "out")
    _same(xi, oi, "x", "out")
    n, _, h, w = xi.shape
    valid = _valid_arg(valid, n, h, w, x.device)
    if reuse is not None and tuple(reuse.index.shape) == (n, 3, h, w) and reuse.index.device == x.device:
        ci = CleanIndex(reuse.index, reuse.ok, reuse.lut, valid)      # same shape: overwrite its buffers
    else:
        ci = CleanIndex(torch.empty((n, 3, h, w), dtype=torch.uint8, device=x.device),
                        torch.empty((n,), dtype=torch.int32, device=x.device),
                        torch.empty((2, 3, 256), dtype=torch.float32, device=x.device), valid)
    u8p, crop_h, crop_w, rs, is_ = _u8_args(u8_out, n, h, w, crop)
    with _on(x):
        _lib.call("adv_clean_index_build_f32", _ptr(xi), _ptr(oi), ci.ref(), u8p, n, h, w, space.ref(), crop_h, crop_w, rs, is_,
                  _stream(x))
    return out, ci


def import_u8(u8, space, pad_to, valid=None, want_clean=True, want_index=True):
    """The loader's transform on the device: ``u8`` uint8 [N,h,w,3] RGB pixels (CUDA) -> (x [N,3,H,W] normalised and zero-padded to
    ``pad_to`` = (H, W) as data.dsgn_transform does on the host - the same bits -, clean or None, CleanIndex or None).  ``valid``: the
    (h_i, w_i) of every image inside the uint8 buffer (default: the whole buffer).  The clean image and its 8-bit index come for free,
    by construction (no verification pass): pass them to ``pgd_step(..., clean_index=...)``."""
    if not (isinstance(u8, torch.Tensor) and u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[3] == 3 and u8.is_contiguous()):
        raise TypeError("u8 must be a contiguous CUDA uint8 tensor [N,h,w,3]")
    if not space.affine:
        raise ValueError("import_u8 implements the DSGN loader (affine pixel space)")
    n, hb, wb, _ = u8.shape
    H, W = int(pad_to[0]), int(pad_to[1])
    if valid is None:
        valid = (hb, wb)
    if not isinstance(valid, torch.Tensor):      # checked on the host before anything is uploaded (a device tensor is trusted)
        sizes = [tuple(valid)] if not isinstance(valid[0], (tuple, list)) else [tuple(v) for v in valid]
        if any(v[0] > hb or v[1] > wb for v in sizes):
            raise ValueError("a valid corner exceeds the uint8 buffer %s" % ((hb, wb),))
    valid = _valid_arg(valid, n, H, W, u8.device)
    x = torch.empty((n, 3, H, W), dtype=torch.float32, device=u8.device)
    clean = torch.empty_like(x) if want_clean else None
    ci = None
    if want_index:
        ci = CleanIndex(torch.empty((n, 3, H, W), dtype=torch.uint8, device=u8.device), torch.empty((n,), dtype=torch.int32, device=u8.device),
                        torch.empty((2, 3, 256), dtype=torch.float32, device=u8.device), valid)
    vh, vw = (0, 0) if isinstance(valid, torch.Tensor) else valid
    if ci is None and isinstance(valid, torch.Tensor):
        raise ValueError("per-image sizes need want_index=True (they travel in the CleanIndex)")
    with _on(u8):
        _lib.call("adv_import_u8_f32", _ptr(u8), u8.stride(1), u8.stride(0), _ptr(x), None if clean is None else _ptr(clean),
                  None if ci is None else ci.ref(), int(vh), int(vw), n, H, W, space.ref(), _stream(u8))
    return x, clean, ci


def alloc_u8(n, crop_h, w, device):
    """Export buffer with whole (uncropped-width) rows: [n, crop_h, w, 3] uint8."""
    return torch.empty((n, crop_h, w, 3), dtype=torch.uint8, device=device)


def _u8_args(u8, n, h, w, crop):
    if u8 is None:
        return ctypes.c_void_p(0), h, w, 0, 0
    crop_h, crop_w = crop if crop is not None else (h, w)
    if u8.dtype != torch.uint8 or not u8.is_cuda or u8.dim() != 4 or u8.shape[0] != n or u8.shape[3] != 3:
        raise ValueError("u8_out must be a CUDA uint8 tensor [n, rows, cols, 3]")
    if u8.stride(3) != 1 or u8.stride(2) != 3:
        raise ValueError("u8_out pixels must be packed HWC")
    if u8.shape[1] < crop_h or u8.shape[2] < crop_w:
        raise ValueError("u8_out %s is smaller than the crop %s" % (tuple(u8.shape), (crop_h, crop_w)))
    # rows of >= w pixels: the library stores whole rows with aligned 12-byte stores; shorter rows: only the
    # crop_w columns, byte by byte (include/advengine.h, adv_pgd_step_f32)
    return _ptr(u8), crop_h, crop_w, u8.stride(1), u8.stride(0)


def pgd_step(x, grad, clean, space, alpha, eps, out=None, u8_out=None, crop=None, clean_index=None):
    """One PGD/FGSM step for a batch of images in one pass over memory.

    DSGN (affine space):  attack/DSGN/pgd_attack.py:339-354 - x is the normalised image the
    detector saw, grad = x.grad, clean the DENORMALISED clean image (:297-298).
    Stereo R-CNN (identity space): attack/Stereo-RCNN/pgd_attack.py:177-217 - eps is already
    ``255 * args.eps`` (:57).

    ``clean_index`` (from ``denormalize_indexed``) lets the kernel read the clean image as bytes - same results.
    ``out`` (default: new tensor; pass ``x`` for in place) receives the next iterate;
    ``u8_out`` (optional, ``alloc_u8``) the 8-bit HWC image the reference would write to PNG
    for it (DSGN tensor2im, pgd_attack.py:157-179; Stereo R-CNN :233-237), rows/cols beyond
    ``crop=(h, w)`` being padding.
    """
    xi, gi, ci = _img(x, "x"), _img(grad, "grad"), _img(clean, "clean")
    _same(xi, gi, "x", "grad")
    _same(xi, ci, "x", "clean")
    out = torch.empty_like(x) if out is None else out
    oi = _img(out, "out")
    _same(xi, oi, "x", "out")
    n, _, h, w = xi.shape
    u8p, crop_h, crop_w, rs, is_ = _u8_args(u8_out, n, h, w, crop)
    with _on(x):
        if clean_index is not None:
            if tuple(clean_index.index.shape) != tuple(xi.shape) or clean_index.index.dtype != torch.uint8:
                raise ValueError("clean_index does not belong to this batch")
            _lib.call("adv_pgd_step_indexed_f32", _ptr(xi), _ptr(gi), _ptr(ci), clean_index.ref(), _ptr(oi), u8p,
                      n, h, w, space.ref(), float(alpha), float(eps), crop_h, crop_w, rs, is_, _stream(x))
        else:
            _lib.call("adv_pgd_step_f32", _ptr(xi), _ptr(gi), _ptr(ci), _ptr(oi), u8p, n, h, w, space.ref(),
                      float(alpha), float(eps), crop_h, crop_w, rs, is_, _stream(x))
    return out


def export_u8(x, space, crop=None, out=None):
    """The 8-bit HWC image of ``x`` as the reference saves it (iterate 0 = the clean pair,
    attack/DSGN/pgd_attack.py:279-294).  Returns [n, crop_h, W, 3]; columns >= crop_w are padding."""
    xi = _img(x, "x")
    n, _, h, w = xi.shape
    crop = (h, w) if crop is None else crop
    out = alloc_u8(n, crop[0], w, x.device) if out is None else out
    u8p, crop_h, crop_w, rs, is_ = _u8_args(out, n, h, w, crop)
    with _on(x):
        _lib.call("adv_export_u8_f32", _ptr(xi), u8p, n, h, w, space.ref(), crop_h, crop_w, rs, is_, _stream(x))
    return out


def disc_mask(h, w, cy, cx, radius, device):
    """generate_round_mask's mask (attack/DSGN/patch_attack.py:245-248) as float32 [h,w]."""
    out = torch.empty((h, w), dtype=torch.float32, device=device)
    with _on(out):
        _lib.call("adv_disc_mask_f32", _ptr(out), h, w, int(cy), int(cx), int(radius), _stream(out))
    return out


def _patch(p, name="patch"):
    if not (isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
        raise TypeError("%s must be a contiguous float32 CUDA tensor" % name)
    if p.dim() == 4 and p.shape[0] == 1:
        p = p[0]
    if p.dim() != 3 or p.shape[0] != 3 or p.shape[1] != p.shape[2] or p.shape[1] % 2 != 1:
        raise ValueError("%s must be [1,3,D,D] or [3,D,D] with D odd, got %s" % (name, tuple(p.shape)))
    return p


def patch_paste(img, patch, cy, cx, radius):
    """In-place paste of the round patch into ONE image,
    attack/DSGN/patch_attack.py:326-333,369-376 (Stereo R-CNN :178-185,221-230)."""
    ii, pp = _img(img, "img"), _patch(patch)
    if ii.shape[0] != 1:
        raise ValueError("patch_paste takes one image; use patch_paste_batch")
    _, _, h, w = ii.shape
    with _on(img):
        _lib.call("adv_patch_paste_f32", _ptr(ii), _ptr(pp), h, w, pp.shape[1], int(cy), int(cx), int(radius), _stream(img))
    return img


def _centers(c, n, k):
    if not (isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.int32 and c.is_contiguous()
            and tuple(c.shape) == (n, k)):
        raise TypeError("centers must be a contiguous int32 CUDA tensor [%d,%d]" % (n, k))
    return c


def patch_paste_batch(img, patch, centers, radius):
    """Paste into every image of [N,3,H,W]; ``centers`` int32 [N,2] = (cy, cx) on the device."""
    ii, pp = _img(img, "img"), _patch(patch)
    n, _, h, w = ii.shape
    cc = _centers(centers, n, 2)
    with _on(img):
        _lib.call("adv_patch_paste_batch_f32", _ptr(ii), _ptr(pp), n, h, w, pp.shape[1], _ptr(cc), int(radius), _stream(img))
    return img


def patch_update(patch, grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha=1e3, lo=None, hi=None, delta_out=None):
    """In-place per-image patch update, attack/DSGN/patch_attack.py:416-430; with lo/hi the
    Stereo R-CNN per-channel clamp follows (attack/Stereo-RCNN/patch_attack.py:257-281)."""
    pp = _patch(patch)
    gl, gr = _img(grad_l, "grad_l"), _img(grad_r, "grad_r")
    _same(gl, gr, "grad_l", "grad_r")
    if gl.shape[0] != 1:
        raise ValueError("patch_update takes one image pair; use patch_delta_batch + patch_apply")
    _, _, h, w = gl.shape
    dp = ctypes.c_void_p(0) if delta_out is None else _ptr(_patch(delta_out, "delta_out"))
    with _on(patch):
        _lib.call("adv_patch_update_f32", _ptr(pp), _ptr(gl), _ptr(gr), h, w, pp.shape[1], int(cy), int(cx_l), int(cx_r),
                  int(radius), float(0.5 * alpha), float(eps), _f3(lo), _f3(hi), dp, _stream(patch))
    return patch


def patch_delta_batch(grad_l, grad_r, centers, radius, eps, alpha=1e3, out=None):
    """Sum over N image pairs of clamp(0.5*alpha*(gL_win + gR_win), +-eps), evaluated against one
    patch snapshot (data-parallel form of patch_attack.py:416-430).  centers int32 [N,3] =
    (cy, cxL, cxR).  Returns [3,D,D] - the buffer the multi-GPU all-reduce carries."""
    gl, gr = _img(grad_l, "grad_l"), _img(grad_r, "grad_r")
    _same(gl, gr, "grad_l", "grad_r")
    n, _, h, w = gl.shape
    cc = _centers(centers, n, 3)
    d = 2 * int(radius) + 1
    out = torch.empty((3, d, d), dtype=torch.float32, device=gl.device) if out is None else out
    oo = _patch(out, "out")
    with _on(gl):
        _lib.call("adv_patch_delta_batch_f32", _ptr(gl), _ptr(gr), n, h, w, d, _ptr(cc), int(radius), float(0.5 * alpha),
                  float(eps), _ptr(oo), _stream(gl))
    return out


def patch_apply(patch, delta, lo=None, hi=None):
    """patch -= delta, then the optional per-channel clamp (second half of patch_update)."""
    pp, dd = _patch(patch), _patch(delta, "delta")
    if pp.shape != dd.shape:
        raise ValueError("patch and delta shapes differ")
    with _on(patch):
        _lib.call("adv_patch_apply_f32", _ptr(pp), _ptr(dd), pp.shape[1], _f3(lo), _f3(hi), _stream(patch))
    return patch


__all__ = [n for n in dir() if not n.startswith("__")]
