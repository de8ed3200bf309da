"""Host-side operators over torch ROCm tensors -> libadvengine.so.

Each function is the counterpart of one inline block of the reference's attack scripts (cited
per function, paths relative to the reference root) and keeps its argument meaning.  Tensors
stay where they are: the kernels are enqueued on the caller's current torch stream, nothing
is copied, allocated (except a result tensor when ``out`` is not given) or synchronised.

torch is plumbing here (device memory + streams); all arithmetic is in the HIP library.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib, routes
from ._lib import AdvSpace

_lib.load()  # fail at import time if the HIP library is not built


class Space:
    """Per-channel pixel space (adv_space_t)."""

    def __init__(self, c_space, name):
        self.c = c_space
        self.name = name

    @staticmethod
    def dsgn(reference_on_gpu=False):
        """ImageNet-normalised RGB in [0,1]: attack/DSGN/pgd_attack.py:153-154,196-207,349-350.
        ``reference_on_gpu``: re-normalise with the float32 reciprocal, as torch's GPU kernels evaluate ``tensor / std[c]`` -
        bit-identical to a GPU run of the reference script; the default is bit-identical to a CPU run."""
        s = AdvSpace()
        if reference_on_gpu:
            _lib.load().adv_space_dsgn_gpu_reference(ctypes.byref(s))
            return Space(s, "dsgn_norm01_gpu_reference")
        _lib.load().adv_space_dsgn(ctypes.byref(s))
        return Space(s, "dsgn_norm01")

    @staticmethod
    def srcnn():
        """BGR minus PIXEL_MEANS on 0..255: attack/Stereo-RCNN/pgd_attack.py:189-207."""
        s = AdvSpace()
        _lib.load().adv_space_srcnn(ctypes.byref(s))
        return Space(s, "srcnn_meansub255")

    @property
    def affine(self):
        return self.c.kind in (_lib.ADV_SPACE_AFFINE, _lib.ADV_SPACE_AFFINE_RCP)

    @property
    def lo(self):
        return tuple(self.c.lo)

    @property
    def hi(self):
        return tuple(self.c.hi)

    def ref(self):
        return ctypes.byref(self.c)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _img(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("%s must be a CUDA/ROCm tensor (there is no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[1] != 3:
        raise ValueError("%s must be [N,3,H,W] or [3,H,W], got %s" % (name, tuple(t.shape)))
    return t


def _same(a, b, na, nb):
    if a.shape != b.shape or a.device != b.device:
        raise ValueError("%s %s/%s and %s %s/%s differ" % (na, tuple(a.shape), a.device, nb, tuple(b.shape), b.device))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _f3(v):
    return None if v is None else (ctypes.c_float * 3)(*[float(x) for x in v])


class _on:
    """make the tensor's device current for the launch (no-op in the one-process-per-GPU case)"""

    def __init__(self, t):
        self.dev = t.device
        self.ctx = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


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


# --------------------------------------------------------------------------------------------
# K7: plane-sweep cost volume (reached through the detector call, attack/DSGN/pgd_attack.py:308)
def _feat(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float32 CUDA tensor" % name)
    return t


def _shift(shift, b):
    if not (isinstance(shift, torch.Tensor) and shift.is_cuda and shift.dtype == torch.int32 and shift.is_contiguous()
            and shift.dim() == 2 and shift.shape[0] == b):
        raise TypeError("shift must be a contiguous int32 CUDA tensor [B,D]")
    return shift


def psv_build(left, right, shift, out=None):
    """cost[b, :C, d] = left, cost[b, C:, d] = right shifted by shift[b,d] pixels, zero where x < shift.
    left/right [B,C,H,W], shift int32 [B,D] -> [B,2C,D,H,W]."""
    l, r = _feat(left, "left"), _feat(right, "right")
    if l.dim() != 4 or l.shape != r.shape:
        raise ValueError("left/right must be [B,C,H,W] of equal shape")
    b, c, h, w = l.shape
    sh = _shift(shift, b)
    d = sh.shape[1]
    out = torch.empty((b, 2 * c, d, h, w), dtype=torch.float32, device=l.device) if out is None else _feat(out, "out")
    if tuple(out.shape) != (b, 2 * c, d, h, w):
        raise ValueError("out must be [B,2C,D,H,W]")
    with _on(l):
        _lib.call("adv_psv_build_f32", _ptr(l), _ptr(r), _ptr(sh), _ptr(out), b, c, d, h, w, _stream(l))
    return out


def psv_build_bwd(grad_cost, shift):
    """adjoint of psv_build: grad_cost [B,2C,D,H,W] -> (grad_left, grad_right) [B,C,H,W], summed over the
    depth planes in plane order (reproducible float32)."""
    g = _feat(grad_cost, "grad_cost")
    if g.dim() != 5 or g.shape[1] % 2:
        raise ValueError("grad_cost must be [B,2C,D,H,W]")
    b, c2, d, h, w = g.shape
    sh = _shift(shift, b)
    if sh.shape[1] != d:
        raise ValueError("shift has %d planes, grad_cost %d" % (sh.shape[1], d))
    gl = torch.empty((b, c2 // 2, h, w), dtype=torch.float32, device=g.device)
    gr = torch.empty_like(gl)
    with _on(g):
        _lib.call("adv_psv_build_bwd_f32", _ptr(g), _ptr(sh), _ptr(gl), _ptr(gr), b, c2 // 2, d, h, w, _stream(g))
    return gl, gr


def _shift_f(shift, b):
    if not (isinstance(shift, torch.Tensor) and shift.is_cuda and shift.dtype == torch.float32 and shift.is_contiguous()
            and shift.dim() == 2 and shift.shape[0] == b):
        raise TypeError("shift must be a contiguous float32 CUDA tensor [B,D]")
    return shift


def psv_build_lerp(left, right, shift, out=None):
    """``psv_build`` with FRACTIONAL per-plane disparities (float32 [B,D]): the right feature is sampled at x - shift by
    linear interpolation between its two integer neighbours (adv_psv_build_lerp_f32)."""
    l, r = _feat(left, "left"), _feat(right, "right")
    if l.dim() != 4 or l.shape != r.shape:
        raise ValueError("left/right must be [B,C,H,W] of equal shape")
    b, c, h, w = l.shape
    sh = _shift_f(shift, b)
    d = sh.shape[1]
    out = torch.empty((b, 2 * c, d, h, w), dtype=torch.float32, device=l.device) if out is None else _feat(out, "out")
    if tuple(out.shape) != (b, 2 * c, d, h, w):
        raise ValueError("out must be [B,2C,D,H,W]")
    with _on(l):
        _lib.call("adv_psv_build_lerp_f32", _ptr(l), _ptr(r), _ptr(sh), _ptr(out), b, c, d, h, w, _stream(l))
    return out


def psv_build_lerp_bwd(grad_cost, shift):
    g = _feat(grad_cost, "grad_cost")
    if g.dim() != 5 or g.shape[1] % 2:
        raise ValueError("grad_cost must be [B,2C,D,H,W]")
    b, c2, d, h, w = g.shape
    sh = _shift_f(shift, b)
    if sh.shape[1] != d:
        raise ValueError("shift has %d planes, grad_cost %d" % (sh.shape[1], d))
    gl = torch.empty((b, c2 // 2, h, w), dtype=torch.float32, device=g.device)
    gr = torch.empty_like(gl)
    with _on(g):
        _lib.call("adv_psv_build_lerp_bwd_f32", _ptr(g), _ptr(sh), _ptr(gl), _ptr(gr), b, c2 // 2, d, h, w, _stream(g))
    return gl, gr


class PsvBuildLerp(torch.autograd.Function):
    """autograd face of the interpolating cost volume (gradients to the features; the shifts are constants)"""

    @staticmethod
    def forward(ctx, left, right, shift):
        ctx.save_for_backward(shift)
        return psv_build_lerp(left.contiguous(), right.contiguous(), shift)

    @staticmethod
    def backward(ctx, grad_cost):
        (shift,) = ctx.saved_tensors
        gl, gr = psv_build_lerp_bwd(grad_cost.contiguous(), shift)
        return gl, gr, None


class PsvBuild(torch.autograd.Function):
    """autograd face of the two kernels, for detectors built on torch"""

    @staticmethod
    def forward(ctx, left, right, shift):
        ctx.save_for_backward(shift)
        return psv_build(left.contiguous(), right.contiguous(), shift)

    @staticmethod
    def backward(ctx, grad_cost):
        (shift,) = ctx.saved_tensors
        gl, gr = psv_build_bwd(grad_cost.contiguous(), shift)
        return gl, gr, None


# --------------------------------------------------------------------------------------------
# Stereo R-CNN RoI path (attack/Stereo-RCNN/stereo_rcnn.py:44-45,132-134; predict_and_save_pgd.py:300)
def roi_align(feat, rois, pooled, spatial_scale, sampling_ratio=0, out=None):
    """feat [B,C,H,W], rois [R,5] = (batch idx, x1, y1, x2, y2) -> [R,C,PH,PW] (legacy RoIAlign).  A roi with a NEGATIVE batch index is
    skipped: its rows of ``out`` (a tensor to write into; default: a new, uninitialised one) stay as they are."""
    f, r = _feat(feat, "feat"), _feat(rois, "rois")
    if f.dim() != 4 or r.dim() != 2 or r.shape[1] != 5:
        raise ValueError("feat must be [B,C,H,W], rois [R,5]")
    b, c, h, w = f.shape
    ph, pw = (pooled, pooled) if isinstance(pooled, int) else pooled
    if out is None:
        out = torch.empty((r.shape[0], c, ph, pw), dtype=torch.float32, device=f.device)
    elif tuple(_feat(out, "out").shape) != (r.shape[0], c, ph, pw):
        raise ValueError("out must be [R,C,PH,PW]")
    with _on(f):
        _lib.call("adv_roi_align_fwd_f32", _ptr(f), _ptr(r), _ptr(out), b, c, h, w, r.shape[0], ph, pw, float(spatial_scale),
                  int(sampling_ratio), _stream(f))
    return out


def roi_align_bwd_segments(n_rois):
    """into how many segments of consecutive roi indices ``roi_align_bwd`` splits its ordered sum: 1 up to 1024 rois, beyond that
    min(8, ceil(r / 512)) (include/advengine.h)"""
    return int(_lib.load().adv_roi_align_bwd_segments(int(n_rois)))


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale, sampling_ratio=0):
    """adjoint of ``roi_align`` w.r.t. the features, summed in a fixed order (bit-reproducible; no atomics)"""
    g, r = _feat(grad_out, "grad_out"), _feat(rois, "rois")
    b, c, h, w = feat_shape
    if g.dim() != 4 or g.shape[0] != r.shape[0] or g.shape[1] != c:
        raise ValueError("grad_out must be [R,C,PH,PW]")
    gf = torch.empty((b, c, h, w), dtype=torch.float32, device=g.device)
    work = torch.empty((max(1, int(_lib.load().adv_roi_align_bwd_workspace_ints(b, c, h, w, r.shape[0], g.shape[2], g.shape[3]))),), dtype=torch.int32, device=g.device)
    with _on(g):
        _lib.call("adv_roi_align_bwd_f32", _ptr(g), _ptr(r), _ptr(gf), b, c, h, w, r.shape[0], g.shape[2], g.shape[3],
                  float(spatial_scale), int(sampling_ratio), _ptr(work), _stream(g))
    return gf


class RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, pooled, spatial_scale, sampling_ratio):
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat.shape), spatial_scale, sampling_ratio)
        return roi_align(feat.contiguous(), rois.contiguous(), pooled, spatial_scale, sampling_ratio)

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        shape, scale, sr = ctx.meta
        return roi_align_bwd(grad_out.contiguous(), rois, shape, scale, sr), None, None, None, None


class PyramidRoIAlign(torch.autograd.Function):
    """RoI pooling over a feature pyramid (attack/Stereo-RCNN/stereo_rcnn.py:110-141: every roi is pooled from the level its size
    selects) with shapes known on the host: ``owner`` [R] int64 holds each roi's index into ``feats``; level l pools the whole roi list
    with the rois it does not own marked skipped (batch index -1) - the forward launches write disjoint rows of ONE output, the backward
    is the ordered gather per level.  No compaction, no read-back, no element-wise combination of per-level results."""

    @staticmethod
    def forward(ctx, rois, owner, pooled, scales, sampling_ratio, *feats):
        r = rois.contiguous()
        # zeros, not empty: a roi that no level owns (an owner index out of range - NaN / inf boxes from diverged RPN deltas under attack) is
        # skipped by every launch; its rows must be the same bytes on every run
        out = torch.zeros((r.shape[0], feats[0].shape[1], pooled, pooled), dtype=torch.float32, device=r.device)
        mine = []
        minus = torch.full_like(r[:, 0], -1.0)
        for l, f in enumerate(feats):
            m = r.clone()
            m[:, 0] = torch.where(owner == l, r[:, 0], minus)
            roi_align(f.contiguous(), m, pooled, scales[l], sampling_ratio, out=out)
            mine.append(m)
        ctx.save_for_backward(*mine)
        ctx.meta = ([tuple(f.shape) for f in feats], tuple(scales), sampling_ratio)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        shapes, scales, sr = ctx.meta
        g = grad_out.contiguous()
        grads = tuple(roi_align_bwd(g, m, shapes[l], scales[l], sr) if ctx.needs_input_grad[5 + l] else None for l, m in enumerate(ctx.saved_tensors))
        return (None, None, None, None, None) + grads


def nms(boxes, scores, thresh):
    """``nms(boxes[order], scores[order], thresh)`` of attack/Stereo-RCNN/predict_and_save_pgd.py:300: boxes
    [N,4] must already be in descending-score order (the reference sorts before calling); returns the kept
    indices (int64, ascending = descending score).  ``scores`` is accepted for signature parity and unused."""
    bx = _feat(boxes, "boxes")
    if bx.dim() != 2 or bx.shape[1] != 4:
        raise ValueError("boxes must be [N,4]")
    n = bx.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=bx.device)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=bx.device)
    count = torch.zeros((1,), dtype=torch.int32, device=bx.device)
    work = torch.empty((max(1, n * ((n + 63) // 64)),), dtype=torch.int64, device=bx.device)
    with _on(bx):
        _lib.call("adv_nms_f32", _ptr(bx), n, float(thresh), _ptr(keep), _ptr(count), _ptr(work), _stream(bx))
    return keep[:int(count.item())]


def nms_padded(boxes, scores, thresh):
    """the same suppression with NO host read-back (capturable in a hipGraph): -> keep [N] int64 whose first ``count`` entries are the
    kept indices (the rest is -1), count [1] int32 on the device"""
    bx = _feat(boxes, "boxes")
    if bx.dim() != 2 or bx.shape[1] != 4:
        raise ValueError("boxes must be [N,4]")
    n = bx.shape[0]
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=bx.device)
    count = torch.zeros((1,), dtype=torch.int32, device=bx.device)
    if n:
        work = torch.empty((max(1, n * ((n + 63) // 64)),), dtype=torch.int64, device=bx.device)
        with _on(bx):
            _lib.call("adv_nms_f32", _ptr(bx), n, float(thresh), _ptr(keep), _ptr(count), _ptr(work), _stream(bx))
    return keep, count


# --------------------------------------------------------------------------------------------
# dense 3x3x3 convolution on the float32 matrix cores (the contraction applied to the K7 cost volume)
def conv3d_k3_prep(weight, transpose=False):
    """[Cout,Cin,3,3,3] -> the kernel's layout [27, Cin', 32*ceil(Cout'/32)]; transpose=True prepares the adjoint
    (gradient w.r.t. the input).  Do it once per weight tensor - the attacks never change the weights."""
    wt = _feat(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight must be [Cout,Cin,3,3,3]")
    cout, cin = wt.shape[:2]
    cin_p, cout_p = (cout, cin) if transpose else (cin, cout)
    out = torch.empty((27, cin_p, 32 * ((cout_p + 31) // 32)), dtype=torch.float32, device=wt.device)
    with _on(wt):
        _lib.call("adv_conv3d_k3_prep_weights_f32", _ptr(wt), _ptr(out), cout, cin, int(transpose), _stream(wt))
    return out


ALL_TAPS = (1 << 27) - 1


def _i3(v):
    return None if v is None else (ctypes.c_int32 * 3)(*[int(a) for a in v])


def _residual(residual, out):
    if residual is None:
        return None
    r = _feat(residual, "residual")
    if r.shape != out.shape or r.data_ptr() == out.data_ptr():
        raise ValueError("residual must have the result's shape %s and must not be the result" % (tuple(out.shape),))
    return _ptr(r)


def _conv3d_ex(x, w_prep, cout, stride=1, relu=False, bias=None, tap_mask=ALL_TAPS, out=None, out_stride=None, out_offset=None,
               class_masks=None, residual=None):
    xi, wp = _feat(x, "x"), _feat(w_prep, "w_prep")
    if xi.dim() != 5 or wp.dim() != 3 or wp.shape[0] != 27 or wp.shape[1] != xi.shape[1] or wp.shape[2] < cout:
        raise ValueError("x must be [B,Cin,D,H,W] and w_prep [27,Cin,>=cout]")
    b, cin, d, h, w = xi.shape
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    grid = tuple((v + 1) // 2 for v in (d, h, w)) if stride == 2 else (d, h, w)
    if out is None:
        out = torch.empty((b, cout) + grid, dtype=torch.float32, device=xi.device)
        dims = None
    else:
        out = _feat(out, "out")
        if out.dim() != 5 or out.shape[0] != b or out.shape[1] != cout:
            raise ValueError("out must be [B,cout,D',H',W']")
        dims = tuple(out.shape[2:])
    cm = None if class_masks is None else (ctypes.c_uint32 * 8)(*[int(m) for m in class_masks])
    with _on(xi):
        _lib.call("adv_conv3d_k3_ex_f32", _ptr(xi), _ptr(wp), None if bias is None else _ptr(bias), _residual(residual, out), _ptr(out), b, cin,
                  cout, d, h, w, int(stride),
                  int(relu), int(tap_mask), cm, 0 if cm is None else cin // 8, _i3(dims), _i3(out_stride if dims else None),
                  _i3(out_offset if dims else None), _stream(xi))
    return out


def conv3d_k3(x, w_prep, cout, relu=False, bias=None, residual=None):
    """conv3d(x [B,Cin,D,H,W], stride 1, padding 1) (+ bias [cout]) (+ residual [B,cout,D,H,W]) (+ ReLU) with prepared weights
    -> [B,cout,D,H,W]"""
    if bias is not None or residual is not None:
        return _conv3d_ex(x, w_prep, cout, 1, relu, bias, residual=residual)
    xi, wp = _feat(x, "x"), _feat(w_prep, "w_prep")
    if xi.dim() != 5 or wp.dim() != 3 or wp.shape[0] != 27 or wp.shape[1] != xi.shape[1] or wp.shape[2] < cout:
        raise ValueError("x must be [B,Cin,D,H,W] and w_prep [27,Cin,>=cout]")
    b, cin, d, h, w = xi.shape
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    with _on(xi):
        _lib.call("adv_conv3d_k3_f32", _ptr(xi), _ptr(wp), _ptr(y), b, cin, cout, d, h, w, int(relu), _stream(xi))
    return y


def conv3d_k3_masked(x, w_prep, cout, mask):
    """conv3d(x, w_prep) zeroed where ``mask`` <= 0 - the backward of a layer whose input (``mask``) is a ReLU output it alone consumes;
    returns None when the shape is not one the main matrix kernel takes (the caller then masks with relu_backward)"""
    xi, wp, mk = _feat(x, "x"), _feat(w_prep, "w_prep"), _feat(mask, "mask")
    b, cin, d, h, w = xi.shape
    if tuple(mk.shape) != (b, cout, d, h, w):
        raise ValueError("mask must be [B,cout,D,H,W]")
    if cin % 4 or cout <= 8:
        return None
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    with _on(xi):
        rc = _lib.load().adv_conv3d_k3_masked_f32(_ptr(xi), _ptr(wp), _ptr(mk), _ptr(y), b, cin, cout, d, h, w, _stream(xi))
    if rc == _lib.ADV_EINVAL:
        return None
    _lib.check("adv_conv3d_k3_masked_f32", rc)
    return y


class Conv3dWinoPrep:
    """a 3x3x3 / stride 1 layer's weights transformed for the Winograd kernel (csrc/wino2d.hip, adv_conv3d_wino_f32): G g G^T per channel
    pair and depth tap, for the forward and for the backward w.r.t. the input; each made on first use, once"""

    def __init__(self, weight):
        wt = _feat(weight.detach().contiguous(), "weight")
        if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
            raise ValueError("weight must be [Cout,Cin,3,3,3]")
        self.cout, self.cin, self._wt, self._u = int(wt.shape[0]), int(wt.shape[1]), wt, {}

    def u(self, transpose):
        t = self._u.get(bool(transpose))
        if t is None:
            n = int(_lib.load().adv_conv3d_wino_prep_floats(self.cout, self.cin, int(transpose)))
            t = torch.empty((n,), dtype=torch.float32, device=self._wt.device)
            with _on(self._wt):
                _lib.call("adv_conv3d_wino_prep_weights_f32", _ptr(self._wt), _ptr(t), self.cout, self.cin, int(transpose), _stream(self._wt))
            self._u[bool(transpose)] = t
        return t


def _conv3d_wino_call(x, u, cin, cout, bias, residual, relu, mask, tile):
    xi = _feat(x, "x")
    if xi.dim() != 5 or xi.shape[1] != cin:
        raise ValueError("x must be [B,%d,D,H,W]" % cin)
    b, _, d, h, w = xi.shape
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    with _on(xi):
        _lib.call("adv_conv3d_wino_f32", _ptr(xi), _ptr(u), None if bias is None else _ptr(bias), _like(residual, y, "residual"), _like(mask, y, "mask"),
                  _ptr(y), b, cin, cout, d, h, w, int(bool(relu)), int(tile), _stream(xi))
    return y


def conv3d_wino(x, prep, bias=None, residual=None, relu=False, mask=None, tile=-1):
    """conv3d(x [B,Cin,D,H,W], 3x3x3, stride 1, padding 1) (+ bias) (+ residual) (ReLU) (zeroed where mask <= 0) by the Winograd kernel:
    2.25x fewer multiply-adds than conv3d_k3, its own order of float operations (oracle: conv3d_wino)"""
    return _conv3d_wino_call(x, prep.u(False), prep.cin, prep.cout, bias, residual, relu, mask, tile)


def conv3d_wino_dgrad(grad, prep, residual=None, mask=None, tile=-1):
    """the backward w.r.t. the input of the same layer: grad [B,Cout,D,H,W] -> [B,Cin,D,H,W] (+ residual) (zeroed where mask <= 0)"""
    return _conv3d_wino_call(grad, prep.u(True), prep.cout, prep.cin, None, residual, False, mask, tile)


def space_to_depth2(x, out=None):
    """[B,C,D,H,W] -> [B,8C,ceil(D/2),ceil(H/2),ceil(W/2)]: the eight parity sub-volumes side by side in the channel dimension"""
    xi = _feat(x, "x")
    b, c, d, h, w = xi.shape
    shape = (b, 8 * c, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2)
    out = torch.empty(shape, dtype=torch.float32, device=xi.device) if out is None else _feat(out, "out")
    if tuple(out.shape) != shape:
        raise ValueError("out must be %s" % (shape,))
    with _on(xi):
        _lib.call("adv_space_to_depth2_f32", _ptr(xi), _ptr(out), b, c, d, h, w, _stream(xi))
    return out


def conv3d_k3_s2_prep(weight):
    """[Cout,Cin,3,3,3] -> (w_prep over the 8*Cin space-to-depth channels, the eight class tap masks, the plain w_prep for the
    direct route).
    out[o] = W0 x[2o-1] + W1 x[2o] + W2 x[2o+1] per axis = W1 e[o] + W0 odd[o-1] + W2 odd[o] with e[j] = x[2j], odd[j] = x[2j+1]:
    the even sub-volume uses tap index 1 (offset 0) with kernel element 1; the odd one tap 0 (offset -1) with element 0 and
    tap 1 with element 2."""
    wt = _feat(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight must be [Cout,Cin,3,3,3]")
    cout, cin = wt.shape[:2]
    pairs = {0: ((1, 1),), 1: ((0, 0), (1, 2))}                                   # parity -> ((tap index, kernel element), ...)
    w8 = torch.zeros((cout, 8 * cin, 3, 3, 3), dtype=torch.float32, device=wt.device)
    masks = []
    for p in range(8):
        mask = 0
        for td, kd in pairs[p >> 2]:
            for th, kh in pairs[(p >> 1) & 1]:
                for tw, kw in pairs[p & 1]:
                    w8[:, p * cin:(p + 1) * cin, td, th, tw] = wt[:, :, kd, kh, kw]
                    mask |= 1 << (td * 9 + th * 3 + tw)
        masks.append(mask)
    return conv3d_k3_prep(w8), tuple(masks), conv3d_k3_prep(wt)


def conv3d_k3_s2_stage_channels(x, cout):
    """2 or 4: the input channels per stage (= the float32 accumulation order) ``conv3d_k3_s2(x, conv3d_k3_prep(w), cout)`` will use -
    2 is the direct strided matrix kernel (W % 4 == 0, 16-byte aligned x), 4 the scalar-staging kernel; the oracle takes it as ``chunk``"""
    xi = _feat(x, "x")
    return int(_lib.load().adv_conv3d_k3_s2_stage_channels(_ptr(xi), int(cout), int(xi.shape[4])))


def conv3d_k3_s2(x, prep, cout, relu=False, bias=None, route="auto"):
    """the strided 3x3x3 convolution of an hourglass: stride 2, padding 1 -> [B,cout,ceil(D/2),ceil(H/2),ceil(W/2)].
    ``prep`` = ``conv3d_k3_s2_prep(weight)`` holds the weights for both routes:
      "direct"  the strided matrix kernel on the raw input (W % 4 == 0: two-channel stages, no permuted copy - the fastest
                route; otherwise the scalar-staging kernel, slow);
      "s2d"     space-to-depth + the stride-1 MFMA kernel with per-class tap masks (any shape; no wasted matrix work);
      "auto"    direct where the matrix kernel takes it, else s2d.
    The routes accumulate in different orders (last-bit differences; each is bit-exact against the oracle run its way).
    A plain ``conv3d_k3_prep(weight)`` tensor is accepted too (direct only)."""
    if isinstance(prep, torch.Tensor):
        return _conv3d_ex(x, prep, cout, 2, relu, bias)
    w_prep8, masks = prep[0], prep[1]
    plain = prep[2] if len(prep) > 2 else None
    if route not in ("auto", "direct", "s2d"):
        raise ValueError("route must be auto, direct or s2d")
    if plain is not None and (route == "direct" or (route == "auto" and conv3d_k3_s2_stage_channels(x, cout) == 2)):
        return _conv3d_ex(x, plain, cout, 2, relu, bias)
    if route == "direct":
        raise ValueError("this prep holds no weights for the direct route")
    return _conv3d_ex(space_to_depth2(x), w_prep8, cout, 1, relu, bias, class_masks=masks)


def conv_transpose3d_k3_s2_prep(weight_t):
    """ConvTranspose3d weights [Cin,Cout,3,3,3] (kernel 3, stride 2, padding 1, output_padding 1) -> the eight output
    parity classes [(w_prep, tap_mask, (pd,ph,pw))].  Output voxel o = 2j + p takes input j + (t - 1) through kernel tap k:
    p = 0: (t, k) = (1, 1);  p = 1: (t, k) = (1, 2), (2, 0)  per axis - so class (pd,ph,pw) is an ordinary convolution over
    the INPUT grid with 1-8 taps, written to every second output voxel."""
    wt = _feat(weight_t, "weight_t")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight_t must be [Cin,Cout,3,3,3]")
    pairs = {0: ((1, 1),), 1: ((1, 2), (2, 0))}
    out = []
    for pd in (0, 1):
        for ph in (0, 1):
            for pw in (0, 1):
                wc = torch.zeros((wt.shape[1], wt.shape[0], 3, 3, 3), dtype=torch.float32, device=wt.device)
                mask = 0
                for td, kd in pairs[pd]:
                    for th, kh in pairs[ph]:
                        for tw, kw in pairs[pw]:
                            wc[:, :, td, th, tw] = wt[:, :, kd, kh, kw].t()
                            mask |= 1 << (td * 9 + th * 3 + tw)
                out.append((conv3d_k3_prep(wc.contiguous()), mask, (pd, ph, pw)))
    return out


def conv_transpose3d_k3_s2(x, classes, cout, relu=False, bias=None, residual=None):
    """conv_transpose3d(x [B,Cin,D,H,W], kernel 3, stride 2, padding 1, output_padding 1) -> [B,cout,2D,2H,2W]: the eight
    output parity classes (1-8 taps each) as ONE launch of the persistent masked MFMA kernel, the class being a tile index.
    ``residual`` [B,cout,2D,2H,2W] (a skip connection) is added in the epilogue, after the bias and before the ReLU."""
    xi = _feat(x, "x")
    b, cin, d, h, w = xi.shape
    if len(classes) != 8 or [tuple(c[2]) for c in classes] != [(k >> 2 & 1, k >> 1 & 1, k & 1) for k in range(8)]:
        raise ValueError("classes must be the eight (w_prep, tap_mask, (pd,ph,pw)) of conv_transpose3d_k3_s2_prep, in its order")
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    out = torch.empty((b, cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=xi.device)
    wps = (ctypes.c_void_p * 8)(*[_feat(c[0], "w_prep").data_ptr() for c in classes])
    masks = (ctypes.c_uint32 * 8)(*[int(c[1]) for c in classes])
    with _on(xi):
        _lib.call("adv_conv_transpose3d_k3_s2_f32", _ptr(xi), wps, masks, None if bias is None else _ptr(bias), _residual(residual, out),
                  _ptr(out), b, cin, cout, d, h, w, int(relu), _stream(xi))
    return out


def conv_transpose3d_k3_s2_dgrad(grad, classes, cout, residual=None, mask=None):
    """(conv_transpose3d(grad) + residual) zeroed where mask <= 0, in ONE launch: the backward of a strided 3x3x3 convolution whose input
    (``mask``, a ReLU output) also feeds a skip path whose gradient is ``residual``.  Returns None where the library has no fused kernel
    for the call (misaligned tensors, W < 4): the caller then adds / masks in passes of its own."""
    xi, mk = _feat(grad, "grad"), _feat(mask, "mask")
    b, cin, d, h, w = xi.shape
    out = torch.empty((b, cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=xi.device)
    if tuple(mk.shape) != tuple(out.shape):
        raise ValueError("mask must be laid out like the result %s" % (tuple(out.shape),))
    wps = (ctypes.c_void_p * 8)(*[_feat(c[0], "w_prep").data_ptr() for c in classes])
    masks = (ctypes.c_uint32 * 8)(*[int(c[1]) for c in classes])
    with _on(xi):
        rc = _lib.load().adv_conv_transpose3d_k3_s2_dgrad_f32(_ptr(xi), wps, masks, _residual(residual, out), _ptr(mk), _ptr(out), b, cin, cout, d, h, w,
                                                             _stream(xi))
    if rc == _lib.ADV_EINVAL:
        return None
    _lib.check("adv_conv_transpose3d_k3_s2_dgrad_f32", rc)
    return out


def relu_backward(grad, y):
    """grad where y > 0 else 0 (threshold_backward), one pass; ``y`` is the output of the fused-ReLU convolution"""
    g, yy = _feat(grad.contiguous(), "grad"), _feat(y, "y")
    if g.shape != yy.shape:
        raise ValueError("grad and y must have the same shape")
    out = torch.empty_like(g)
    if g.numel() == 0:
        return out
    with _on(g):
        _lib.call("adv_relu_backward_f32", _ptr(g), _ptr(yy), _ptr(out), g.numel(), _stream(g))
    return out


class Conv3dK3(torch.autograd.Function):
    """y = [relu](conv3d(x, weight) [+ bias]); gradient flows to x only (the attacks differentiate w.r.t. the images, the
    detector's weights are constants), through the same kernel family with the transposed / flipped weights (cout a
    multiple of 4 -> the matrix kernel, cout 1..3 -> the narrow vector-ALU kernel).  Any other cout: pass
    ``w_prep_t=None`` and the original ``weight``: the backward then uses torch's conv3d_input.  With ``relu`` the
    activation is fused into the forward's epilogue and its mask is applied to the incoming gradient; ``residual`` (a skip
    connection, [B,cout,D,H,W]) is added in the same epilogue and receives that gradient unchanged."""

    @staticmethod
    def forward(ctx, x, w_prep, w_prep_t, cout, weight=None, bias=None, relu=False, residual=None, mask_input=False, wino=None):
        """<round 3> chains, as ops.Conv2dAuto: ``mask_input`` - x is a ReLU output this layer alone consumes, the gradient returned for it
        is already masked with x > 0 (in the dgrad kernel's epilogue); ``relu="consumer"`` - this layer's ReLU mask is applied by its only
        consumer's backward, not here.
        ``wino`` (a Conv3dWinoPrep of the same weights): each direction is computed by the direct kernel or by the Winograd kernel
        (csrc/wino2d.hip: 2.25x fewer multiply-adds, same epilogues), whichever measured faster for the layer shape at first use."""
        x = x.contiguous()
        ctx.has_t, ctx.has_res = w_prep_t is not None, residual is not None
        ctx.mask_own, ctx.mask_input = bool(relu) and relu != "consumer", bool(mask_input)
        ctx.xshape, ctx.wino = tuple(x.shape), wino
        res = None if residual is None else residual.contiguous()
        direct = lambda: conv3d_k3(x, w_prep, cout, relu=bool(relu), bias=bias, residual=res)       # noqa: E731
        if wino is not None and cout >= 4:
            by_wino = lambda: conv3d_wino(x, wino, bias, res, bool(relu))                            # noqa: E731
            key = ("f3", x.shape[1], cout, tuple(x.shape), res is not None, bool(relu))
            took = _Conv2dChoice.pick(key, {"direct": direct, "wino": by_wino}) == "wino"
            y = by_wino() if took else direct()
            if took:
                WINO_DIRECT_EQUIV_FLOPS[0] += 54 * y.numel() * x.shape[1]
        else:
            y = direct()
        ctx.save_for_backward(w_prep_t if ctx.has_t else weight, y if ctx.mask_own else None, x if mask_input else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        w, y, x_in = ctx.saved_tensors
        if ctx.mask_own:
            grad_y = relu_backward(grad_y, y)
        gres = grad_y if ctx.has_res else None      # the skip connection receives the (masked) gradient as it is
        if ctx.has_t:
            g = grad_y.contiguous()

            def direct():
                gx = conv3d_k3_masked(g, w, ctx.xshape[1], x_in) if ctx.mask_input else None
                if gx is None:
                    gx = conv3d_k3(g, w, ctx.xshape[1])
                    if ctx.mask_input:
                        gx = relu_backward(gx, x_in)
                return gx

            if ctx.wino is not None and ctx.xshape[1] >= 4:
                by_wino = lambda: conv3d_wino_dgrad(g, ctx.wino, mask=x_in if ctx.mask_input else None)      # noqa: E731
                key = ("b3", ctx.xshape[1], g.shape[1], ctx.xshape, ctx.mask_input)
                took = _Conv2dChoice.pick(key, {"direct": direct, "wino": by_wino}) == "wino"
                gx = by_wino() if took else direct()
                if took:
                    WINO_DIRECT_EQUIV_FLOPS[0] += 54 * gx.numel() * g.shape[1]
            else:
                gx = direct()
        else:
            gx = torch.nn.grad.conv3d_input(ctx.xshape, w, grad_y, padding=1)
            if ctx.mask_input:
                gx = relu_backward(gx, x_in)
        return gx, None, None, None, None, None, None, gres, None, None


class Conv3dK3S2(torch.autograd.Function):
    """y = conv3d(x, weight, stride 2, padding 1); the gradient w.r.t. x is the transposed convolution of grad_y with the
    same weights (``classes_t = conv_transpose3d_k3_s2_prep(weight)``), cropped to x's size when a dimension is odd.

    <round 4> An hourglass's down-sampling layer reads a tensor that ALSO feeds the matching up-sampling layer's skip connection.  Two
    flags take the addition of the two gradients and the producer's ReLU mask into this layer's backward launch (as ops.Conv2dAuto's):
      ``skip_out=True``    returns (y, x_skip), x_skip an alias of x: hand it to the up-sampling layer as its ``residual``; this layer's backward
                           then receives both gradients of x and adds the skip path's in the transposed kernel's epilogue;
      ``mask_input=True``  x is a ReLU output whose only consumers are this layer and that skip path, and its producer left the mask to us
                           (relu="consumer"): the gradient returned is already multiplied by (x > 0).
    The caller vouches for the topology; same float operations in the same order as autograd's addition and a relu-backward pass."""

    @staticmethod
    def forward(ctx, x, w_prep, classes_t, cout, bias=None, relu=False, mask_input=False, skip_out=False):
        ctx.classes_t, ctx.xshape = classes_t, tuple(x.shape)
        ctx.relu = bool(relu) and relu != "consumer"        # "consumer": the only consumer's backward applies this layer's ReLU mask (Conv3dK3 mask_input)
        ctx.mask_input = bool(mask_input)
        x = x.contiguous()
        y = conv3d_k3_s2(x, w_prep, cout, relu=bool(relu), bias=bias)
        ctx.save_for_backward(y if ctx.relu else None, x if mask_input else None)
        return (y, x) if skip_out else y

    @staticmethod
    def backward(ctx, grad_y, grad_skip=None):
        y, x_in = ctx.saved_tensors
        if ctx.relu:
            grad_y = relu_backward(grad_y, y)
        d, h, w = ctx.xshape[2:]
        gy = grad_y.contiguous()
        even = (2 * gy.shape[2], 2 * gy.shape[3], 2 * gy.shape[4]) == (d, h, w)
        skip = None if grad_skip is None else grad_skip.contiguous()
        g = None
        if even and ctx.mask_input:                         # transposed convolution + skip gradient + mask in one launch
            g = conv_transpose3d_k3_s2_dgrad(gy, ctx.classes_t, ctx.xshape[1], residual=skip, mask=x_in)
        if g is None:
            g = conv_transpose3d_k3_s2(gy, ctx.classes_t, ctx.xshape[1], residual=skip if even else None)
            if tuple(g.shape[2:]) != (d, h, w):
                g = g[:, :, :d, :h, :w].contiguous()
            if skip is not None and not even:
                g = g + skip
            if ctx.mask_input:
                g = relu_backward(g, x_in)
        return g, None, None, None, None, None, None, None


class ConvTranspose3dK3S2(torch.autograd.Function):
    """y = conv_transpose3d(x, weight_t, stride 2, padding 1, output_padding 1); the gradient w.r.t. x is the strided
    convolution of grad_y with weight_t read as [out = Cin, in = Cout] (``w_prep_fwd = conv3d_k3_s2_prep(weight_t)``)."""

    @staticmethod
    def forward(ctx, x, classes, w_prep_fwd, cout, bias=None, relu=False, residual=None):
        """relu="consumer": the ReLU is applied here, its backward mask by this layer's only consumer (ops.BevFold mask_input)"""
        ctx.w_prep_fwd, ctx.cin, ctx.has_res = w_prep_fwd, x.shape[1], residual is not None
        ctx.relu = bool(relu) and relu != "consumer"
        y = conv_transpose3d_k3_s2(x.contiguous(), classes, cout, relu=bool(relu), bias=bias, residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(y if ctx.relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        if ctx.relu:
            grad_y = relu_backward(grad_y, y)
        return (conv3d_k3_s2(grad_y.contiguous(), ctx.w_prep_fwd, ctx.cin), None, None, None, None, None,
                grad_y if ctx.has_res else None)        # the skip connection receives the (masked) gradient as it is


# --------------------------------------------------------------------------------------------
# 2D convolutions of the detectors' backbones on the float32 matrix cores (csrc/conv2d.hip)
def conv2d_supported(x, weight, stride=1, padding=0, dilation=1):
    """does libadvengine have a kernel for this layer?  1x1 / stride 1 / no padding (a GEMM per image) and 3x3 / stride 1 /
    padding = dilation in (1, 2); strided layers and other kernel sizes stay on MIOpen"""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2] == weight.shape[3] and stride == 1):
        return False
    if x.dim() != 4 or x.numel() < 4 or x.shape[0] * weight.shape[0] * x.shape[2] * x.shape[3] < 4:
        return False            # the kernels load whole float4s (clamped into the tensor): fewer than four floats are refused
    k = weight.shape[2]
    return (k == 1 and padding == 0 and dilation == 1) or (k == 3 and dilation in (1, 2) and padding == dilation)


class Conv2dPrep:
    """the weights of one layer in the kernels' layout, for the forward and for the backward w.r.t. the input - prepared once (the
    attacks never change the weights)"""

    def __init__(self, weight, stride=1, padding=0, dilation=1):
        wt = _feat(weight.detach().contiguous(), "weight")
        if wt.dim() != 4 or wt.shape[2] != wt.shape[3] or stride != 1:
            raise ValueError("weight must be [Cout,Cin,k,k], stride 1")
        self.cout, self.cin, self.k, self.stride, self.padding, self.dilation = int(wt.shape[0]), int(wt.shape[1]), int(wt.shape[2]), 1, int(padding), int(dilation)
        if not ((self.k == 1 and padding == 0 and dilation == 1) or (self.k == 3 and dilation in (1, 2) and padding == dilation)):
            raise ValueError("only 1x1 / padding 0 and 3x3 / padding = dilation in (1, 2) layers have a kernel here")
        self.device = wt.device
        self.fwd, self.bwd = self._prep(wt, False), self._prep(wt, True)
        self.has_wino = self.k == 3 and self.dilation == 1      # Winograd F(2x2,3x3) route (csrc/wino2d.hip): prepared on first use
        self._wt, self._wino = (wt if self.has_wino else None), {}

    def wino(self, transpose):
        """the layer's transformed weights G g G^T for adv_conv2d_wino_f32 (forward / backward w.r.t. the input), made once"""
        if not self.has_wino:
            raise ValueError("the Winograd route needs a 3x3 / dilation 1 layer")
        t = self._wino.get(bool(transpose))
        if t is None:
            t = self._wino[bool(transpose)] = self._prep(self._wt, transpose, "wino")
        return t

    def _prep(self, wt, transpose, kind=None):
        floats, prep = {1: ("adv_conv2d_1x1_prep_floats", "adv_conv2d_1x1_prep_weights_f32"),
                        3: ("adv_conv2d_3x3_prep_floats", "adv_conv2d_3x3_prep_weights_f32"),
                        "wino": ("adv_conv2d_wino_prep_floats", "adv_conv2d_wino_prep_weights_f32")}[kind or self.k]
        n = int(getattr(_lib.load(), floats)(self.cout, self.cin, int(transpose)))
        out = torch.empty((n,), dtype=torch.float32, device=wt.device)
        with _on(wt):
            _lib.call(prep, _ptr(wt), _ptr(out), self.cout, self.cin, int(transpose), _stream(wt))
        return out


def _like(t, out, name):
    if t is None:
        return None
    t = _feat(t, name)
    if t.shape != out.shape or t.data_ptr() == out.data_ptr():
        raise ValueError("%s must have the result's shape %s and must not be the result" % (name, tuple(out.shape)))
    return _ptr(t)


def _conv2d_call(x, prep, w_prep, cin, cout, bias, residual, relu, mask, tile, wino=False):
    xi = _feat(x, "x")
    if xi.dim() != 4 or xi.shape[1] != cin:
        raise ValueError("x must be [B,%d,H,W]" % cin)
    b, _, h, w = xi.shape
    y = torch.empty((b, cout, h, w), dtype=torch.float32, device=xi.device)
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    bp = None if bias is None else _ptr(bias)
    with _on(xi):
        if wino:
            _lib.call("adv_conv2d_wino_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin,
                      cout, h, w, int(bool(relu)), int(tile), _stream(xi))
        elif prep.k == 1:
            _lib.call("adv_conv2d_1x1_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout,
                      h * w, int(bool(relu)), int(tile), _stream(xi))
        else:
            _lib.call("adv_conv2d_3x3_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout,
                      h, w, prep.dilation, int(bool(relu)), int(min(tile, 2)), _stream(xi))
    return y


def conv2d(x, prep, bias=None, residual=None, relu=False, mask=None, tile=-1, wino=False):
    """conv2d(x [B,Cin,H,W], prep) (+ bias [Cout]) (+ residual [B,Cout,H,W]) (ReLU) (zeroed where mask <= 0) -> [B,Cout,H,W].
    wino=True (3x3 / dilation 1 layers): the Winograd F(2x2,3x3) kernel - 2.25x fewer multiply-adds, its own order of float operations
    (oracle: conv2d_wino)"""
    if wino:
        return _conv2d_call(x, prep, prep.wino(False), prep.cin, prep.cout, bias, residual, relu, mask, tile, True)
    return _conv2d_call(x, prep, prep.fwd, prep.cin, prep.cout, bias, residual, relu, mask, tile)


def conv2d_dgrad(grad, prep, hw=None, residual=None, mask=None, tile=-1, wino=False):
    """the backward w.r.t. the input of the same layer: grad [B,Cout,H,W] -> [B,Cin,H,W] (+ residual: a gradient arriving over a
    skip path) (zeroed where mask <= 0: with mask = the layer's own input, a ReLU output, this is the gradient w.r.t. the previous
    layer's pre-activation)"""
    if wino:
        return _conv2d_call(grad, prep, prep.wino(True), prep.cout, prep.cin, None, residual, False, mask, tile, True)
    return _conv2d_call(grad, prep, prep.bwd, prep.cout, prep.cin, None, residual, False, mask, tile)


class Conv2d(torch.autograd.Function):
    """y = [relu](conv2d(x, prep) [+ bias] [+ residual]); gradients flow to x and to the residual only (the detector's weights are
    constants in an attack).  With ``relu`` the incoming gradient is masked with y > 0 first (ops.relu_backward)."""

    @staticmethod
    def forward(ctx, x, prep, bias=None, residual=None, relu=False):
        ctx.prep, ctx.relu, ctx.has_res = prep, bool(relu), residual is not None
        y = conv2d(x.contiguous(), prep, bias, None if residual is None else residual.contiguous(), relu)
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        g = grad_y.contiguous()
        if ctx.relu:
            g = relu_backward(g, y)
        return conv2d_dgrad(g, ctx.prep), None, None, (g if ctx.has_res else None), None


def bias_act_(y, bias=None, residual=None, relu=False):
    """y <- [relu](y + bias[c] + residual) in place, one pass (the epilogue of a convolution another library computed)"""
    yi = _feat(y, "y")
    if yi.dim() < 2:
        raise ValueError("y must be [B,C,...]")
    b, c = yi.shape[0], yi.shape[1]
    hw = yi.numel() // max(1, b * c)
    if yi.numel() == 0 or (bias is None and residual is None and not relu):
        return yi
    if b * c > 65535:                     # beyond the launch's plane index: torch's own operators
        if bias is not None:
            yi += bias.view(1, -1, *([1] * (yi.dim() - 2)))
        if residual is not None:
            yi += residual
        return yi.relu_() if relu else yi
    if bias is not None:
        bias = _feat(bias, "bias")
    with _on(yi):
        _lib.call("adv_bias_act_f32", _ptr(yi), None if bias is None else _ptr(bias), _like(residual, yi, "residual"), b * c, c, hw, int(bool(relu)),
                  _stream(yi))
    return yi


class BiasAct(torch.autograd.Function):
    """``[relu](y + bias[c] + residual)`` in place as ONE pass behind a convolution torch computed (adopt.py's layers without a kernel here,
    the bird's-eye view's transposed 2D layers), with the backward autograd cannot derive from a raw kernel call: the incoming gradient
    masked with (result > 0) when the ReLU is fused; the residual (a skip connection) receives that masked gradient as it is"""

    @staticmethod
    def forward(ctx, y, bias=None, relu=False, residual=None):
        out = bias_act_(y, bias, None if residual is None else residual.contiguous(), relu)
        ctx.mark_dirty(y)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, grad):
        (out,) = ctx.saved_tensors
        g = relu_backward(grad.contiguous(), out) if ctx.relu else grad
        return g, None, None, (g if ctx.has_res else None)


# direct-convolution FLOPs (2 x MACs) of the calls that took a Winograd route since the last reset: such a call EXECUTES 2.25x fewer
# multiply-adds on the matrix cores than it is credited with (F(2x2,3x3): 16 products per 4 outputs instead of 36) - tools/bench_end_to_end.py
# turns this into executed FLOPs per step and a matrix-pipe utilisation bound next to the direct-equivalent roofline fraction
WINO_DIRECT_EQUIV_FLOPS = [0]


class _Conv2dChoice:
    """Per layer shape and direction: who computes it - this package's direct kernel ("hip"), its Winograd kernel ("wino") or torch's
    operator ("", MIOpen / rocBLAS + one fused element-wise pass).  The three sum in different float orders, so the choice is part of
    the result: it comes from the committed table ``routes_gfx950.json`` (routes.py; a fixed rule for shapes the table does not hold),
    the same in every process and on every rank.  Only ``ADV_ROUTES=measure`` (tools/make_routes.py, the per-layer benches) times the
    candidates, which is how the table is generated (round 3 timed at first use in every run: results differed run to run)."""

    @staticmethod
    def pick(key, fns):
        """the route name for ``key`` among ``fns`` (name -> callable)"""
        return routes.choose(key, fns)

    @staticmethod
    def get(key, hip_fn, torch_fn, wino_fn=None):
        """-> "hip" (direct implicit GEMM), "wino" (Winograd on the matrix cores, 3x3 layers) or "" (torch / MIOpen)"""
        fns = {"hip": hip_fn, "": torch_fn}
        if wino_fn is not None:
            fns["wino"] = wino_fn
        return routes.choose(key, fns)


class Conv2dAuto(torch.autograd.Function):
    """y = [relu](conv2d(x) [+ bias] [+ residual]) and its backward w.r.t. x and the residual, each direction computed by whichever of
    {csrc/conv2d.hip, torch} measured faster for this layer shape (see _Conv2dChoice).  ``prep`` = Conv2dPrep of the layer, ``weight``
    the original tensor for torch's side.

    The ReLU's backward without a pass of its own, for CHAINS (a -> this layer is a's only consumer):
      ``mask_input=True``   x is a ReLU output consumed by this layer alone: the gradient returned for x is already multiplied by
                            (x > 0) - in the dgrad kernel's epilogue when libadvengine computes it - i.e. it is the gradient w.r.t.
                            the producer's PRE-activation;
      ``relu="consumer"``   this layer applies the ReLU in its forward but does not mask in its backward: its only consumer was
                            called with mask_input=True and has done it (y > 0 there is the same mask).
    The caller vouches for the topology; the result is the same gradient, bit for bit (the mask multiplies the same float once).

    Skip connections without the autograd engine's addition:
      ``skip_out=True``     returns (y, x_skip) with x_skip an alias of x.  A residual block hands x_skip to its last layer as the
                            ``residual``; this layer's backward then receives BOTH gradients of x - through its own convolution and over
                            the skip path - and adds the second in the dgrad kernel's epilogue (before the mask of ``mask_input``, which
                            then covers both: x may be a ReLU output whose only consumers are this layer and its block's skip path)."""

    @staticmethod
    def forward(ctx, x, prep, weight, bias=None, residual=None, relu=False, mask_input=False, skip_out=False):
        x = x.contiguous()
        res = None if residual is None else residual.contiguous()
        pad, dil = prep.padding, prep.dilation
        do_relu = bool(relu)
        key = ("f", prep.k, prep.cin, prep.cout, dil, tuple(x.shape), res is not None, do_relu)

        def by_torch():                  # MIOpen's convolution + ONE element-wise pass (bias, skip connection, ReLU)
            return bias_act_(F.conv2d(x, weight, None, 1, pad, dil), bias, res, do_relu)

        use = _Conv2dChoice.get(key, lambda: conv2d(x, prep, bias, res, do_relu), by_torch,
                                (lambda: conv2d(x, prep, bias, res, do_relu, wino=True)) if prep.has_wino else None)
        y = conv2d(x, prep, bias, res, do_relu, wino=(use == "wino")) if use else by_torch()
        if use == "wino":
            WINO_DIRECT_EQUIV_FLOPS[0] += 18 * y.numel() * prep.cin
        ctx.prep, ctx.has_res, ctx.xshape = prep, res is not None, tuple(x.shape)
        ctx.mask_own = do_relu and relu != "consumer"          # mask the incoming gradient with y > 0 here
        ctx.mask_input = bool(mask_input)
        ctx.save_for_backward(weight, y if ctx.mask_own else None, x if mask_input else None)
        return (y, x) if skip_out else y

    @staticmethod
    def backward(ctx, grad_y, grad_skip=None):
        weight, y, x_in = ctx.saved_tensors
        prep = ctx.prep
        g = grad_y.contiguous()
        if ctx.mask_own:
            g = relu_backward(g, y)
        skip = None if grad_skip is None else grad_skip.contiguous()      # the gradient of x over the block's skip path
        key = ("b", prep.k, prep.cin, prep.cout, prep.dilation, ctx.xshape, ctx.mask_input, skip is not None)

        def by_torch():
            gx = torch.ops.aten.convolution_backward(g, _shape_only(ctx.xshape, g), weight, None, [1, 1], [prep.padding, prep.padding],
                                                     [prep.dilation, prep.dilation], False, [0, 0], 1, [True, False, False])[0]
            if skip is not None:
                gx = gx + skip
            return relu_backward(gx, x_in) if ctx.mask_input else gx

        hip = lambda wino=False: conv2d_dgrad(g, prep, residual=skip, mask=x_in if ctx.mask_input else None, wino=wino)      # noqa: E731
        use = _Conv2dChoice.get(key, hip, by_torch, (lambda: hip(True)) if prep.has_wino else None)
        gx = hip(use == "wino") if use else by_torch()
        if use == "wino":
            WINO_DIRECT_EQUIV_FLOPS[0] += 18 * gx.numel() * prep.cout
        return gx, None, None, None, (g if ctx.has_res else None), None, None, None


_SHAPE_DUMMY = {}


def _shape_only(shape, like):
    """convolution_backward wants the input TENSOR although the gradient w.r.t. the input reads only its shape: one cached dummy per shape"""
    key = (tuple(shape), like.device)
    t = _SHAPE_DUMMY.get(key)
    if t is None:
        t = torch.empty(shape, dtype=torch.float32, device=like.device)
        _SHAPE_DUMMY[key] = t
    return t


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


# --------------------------------------------------------------------------------------------
# after the 3D convolutions (csrc/volume.hip): fused depth regression, grid_sample on 5-D volumes, sigmoid focal loss
def depth_regress(cost, depth_values, out_size, align_corners=False, with_stats=False):
    """cost [B,D,h,w], depth_values [Do] -> depth [B,H,W] = sum_k softmax_k(trilinear_upsample(cost, (Do,H,W)))[k] * depth_values[k],
    without ever writing the up-sampled volume.  with_stats: also the per-pixel softmax (max, sum) [B,2,H,W] the backward needs."""
    ci, zv = _feat(cost, "cost"), _feat(depth_values, "depth_values")
    if ci.dim() != 4 or zv.dim() != 1 or len(out_size) != 3 or zv.shape[0] != int(out_size[0]):
        raise ValueError("cost must be [B,D,h,w], out_size (Do,H,W) and depth_values [Do]")
    b, d, h, w = ci.shape
    do, ho, wo = (int(v) for v in out_size)
    depth = torch.empty((b, ho, wo), dtype=torch.float32, device=ci.device)
    stats = torch.empty((b, 2, ho, wo), dtype=torch.float32, device=ci.device) if with_stats else None
    with _on(ci):
        _lib.call("adv_depth_regress_f32", _ptr(ci), _ptr(zv), _ptr(depth), None if stats is None else _ptr(stats), b, d, h, w, do, ho, wo,
                  int(bool(align_corners)), _stream(ci))
    return (depth, stats) if with_stats else depth


def depth_regress_bwd(cost, depth_values, depth, stats, grad_depth, align_corners=False):
    ci, zv, dp, st, g = (_feat(cost, "cost"), _feat(depth_values, "depth_values"), _feat(depth, "depth"), _feat(stats, "stats"),
                         _feat(grad_depth, "grad_depth"))
    b, d, h, w = ci.shape
    _, ho, wo = dp.shape
    if tuple(st.shape) != (b, 2, ho, wo) or tuple(g.shape) != (b, ho, wo):
        raise ValueError("stats must be [B,2,H,W] and grad_depth [B,H,W]")
    work = torch.empty((b, d, ho, wo), dtype=torch.float32, device=ci.device)
    gc = torch.empty_like(ci)
    with _on(ci):
        _lib.call("adv_depth_regress_bwd_f32", _ptr(ci), _ptr(zv), _ptr(dp), _ptr(st), _ptr(g), _ptr(work), _ptr(gc), b, d, h, w, int(zv.shape[0]),
                  ho, wo, int(bool(align_corners)), _stream(ci))
    return gc


class DepthRegress(torch.autograd.Function):
    """depth = DepthRegress.apply(cost [B,D,h,w], depth_values [Do], (Do,H,W), align_corners); gradient w.r.t. cost only"""

    @staticmethod
    def forward(ctx, cost, depth_values, out_size, align_corners=False):
        cost = cost.contiguous()
        depth, stats = depth_regress(cost, depth_values, out_size, align_corners, with_stats=True)
        ctx.save_for_backward(cost, depth_values, depth, stats)
        ctx.align = bool(align_corners)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        cost, zv, depth, stats = ctx.saved_tensors
        return depth_regress_bwd(cost, zv, depth, stats, grad_depth.contiguous(), ctx.align), None, None, None


def grid_sample3d(vol, grid, align_corners=False):
    """torch.nn.functional.grid_sample(vol [B,C,D,H,W], grid [B,Z,Y,X,3], bilinear, zeros) - the same bits as torch on the CPU"""
    vi, gi = _feat(vol, "vol"), _feat(grid, "grid")
    if vi.dim() != 5 or gi.dim() != 5 or gi.shape[0] != vi.shape[0] or gi.shape[4] != 3:
        raise ValueError("vol must be [B,C,D,H,W] and grid [B,Z,Y,X,3]")
    b, c, d, h, w = vi.shape
    zo, yo, xo = gi.shape[1:4]
    out = torch.empty((b, c, zo, yo, xo), dtype=torch.float32, device=vi.device)
    with _on(vi):
        _lib.call("adv_grid_sample3d_f32", _ptr(vi), _ptr(gi), _ptr(out), b, c, d, h, w, zo, yo, xo, int(bool(align_corners)), _stream(vi))
    return out


class GridSamplePlan:
    """The backward's gather plan for one grid (build once per calibration): for every cell of the volume the sorted list of
    (output voxel, weight) that sample it."""

    def __init__(self, grid, vol_dims, align_corners=False):
        gi = _feat(grid, "grid")
        b, zo, yo, xo, _ = gi.shape
        d, h, w = (int(v) for v in vol_dims)
        nbytes = int(_lib.load().adv_grid_sample3d_plan_bytes(b, d, h, w, zo, yo, xo))
        if nbytes <= 0:
            raise ValueError("grid / volume too large for a 32-bit plan")
        self.buf = torch.empty((nbytes // 4,), dtype=torch.int32, device=gi.device)
        self.dims, self.out_dims, self.batch, self.align = (d, h, w), (zo, yo, xo), b, bool(align_corners)
        with _on(gi):
            _lib.call("adv_grid_sample3d_plan_f32", _ptr(gi), _ptr(self.buf), b, d, h, w, zo, yo, xo, int(self.align), _stream(gi))


def grid_sample3d_bwd(grad_out, plan, channels_last=True):
    """gradient w.r.t. the sampled volume; ``channels_last``: through a channels-last copy of grad_out (one run of C floats per list
    entry instead of C cache lines) - the same bits, less HBM traffic"""
    g = _feat(grad_out, "grad_out")
    b, c = g.shape[:2]
    if g.dim() != 5 or b != plan.batch or tuple(g.shape[2:]) != tuple(plan.out_dims):
        raise ValueError("grad_out must be [B,C,Z,Y,X] of the plan's grid")
    d, h, w = plan.dims
    zo, yo, xo = plan.out_dims
    gv = torch.empty((b, c, d, h, w), dtype=torch.float32, device=g.device)
    with _on(g):
        if channels_last:
            work = torch.empty((int(_lib.load().adv_grid_sample3d_bwd_workspace_floats(b, c, zo, yo, xo)),), dtype=torch.float32, device=g.device)
            _lib.call("adv_grid_sample3d_bwd_ws_f32", _ptr(g), _ptr(plan.buf), _ptr(gv), _ptr(work), b, c, d, h, w, zo, yo, xo, _stream(g))
        else:
            _lib.call("adv_grid_sample3d_bwd_f32", _ptr(g), _ptr(plan.buf), _ptr(gv), b, c, d, h, w, zo, yo, xo, _stream(g))
    return gv


class GridSample3d(torch.autograd.Function):
    """out = GridSample3d.apply(vol, grid, plan): gradient w.r.t. vol only (the grid is a function of the calibration), as a
    deterministic gather over ``plan = GridSamplePlan(grid, vol.shape[2:], align_corners)``"""

    @staticmethod
    def forward(ctx, vol, grid, plan):
        ctx.plan = plan
        return grid_sample3d(vol.contiguous(), grid, plan.align)

    @staticmethod
    def backward(ctx, grad_out):
        return grid_sample3d_bwd(grad_out.contiguous(), ctx.plan), None, None


def bilinear_up(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for x [B,C,h,w] (csrc/resize.hip)"""
    xi = _feat(x, "x")
    if xi.dim() != 4 or len(size) != 2 or min(size) < 1:
        raise ValueError("x must be [B,C,h,w] and size (ho, wo)")
    b, c, h, w = xi.shape
    out = torch.empty((b, c, int(size[0]), int(size[1])), dtype=torch.float32, device=xi.device)
    with _on(xi):
        _lib.call("adv_bilinear_up_f32", _ptr(xi), _ptr(out), b * c, h, w, int(size[0]), int(size[1]), _stream(xi))
    return out


def bilinear_up_bwd(grad_out, in_hw):
    """the adjoint of bilinear_up as a fixed-order gather: reproducible bit for bit (torch's backward scatters with atomicAdd)"""
    g = _feat(grad_out, "grad_out")
    if g.dim() != 4:
        raise ValueError("grad_out must be [B,C,ho,wo]")
    b, c, ho, wo = g.shape
    gin = torch.empty((b, c, int(in_hw[0]), int(in_hw[1])), dtype=torch.float32, device=g.device)
    with _on(g):
        _lib.call("adv_bilinear_up_bwd_f32", _ptr(g), _ptr(gin), b * c, int(in_hw[0]), int(in_hw[1]), ho, wo, _stream(g))
    return gin


class BilinearUp(torch.autograd.Function):
    """the FPN top-down path's up-sampling (attack/Stereo-RCNN/stereo_rcnn.py:92-108) with a deterministic backward"""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = tuple(x.shape[2:])
        return bilinear_up(x.contiguous(), size)

    @staticmethod
    def backward(ctx, grad):
        return bilinear_up_bwd(grad.contiguous(), ctx.in_hw), None


def bev_fold(v, pool):
    """[B,C,Z,Y,X] -> [B, C * (Y // pool), Z, X]: F.avg_pool3d(v, (1, pool, 1)).permute(0, 1, 3, 2, 4).reshape(...) in one pass"""
    vi = _feat(v, "v")
    if vi.dim() != 5 or pool < 1 or pool > vi.shape[3]:
        raise ValueError("v must be [B,C,Z,Y,X] with Y >= pool")
    b, c, z, y, x = vi.shape
    out = torch.empty((b, c * (y // pool), z, x), dtype=torch.float32, device=vi.device)
    with _on(vi):
        _lib.call("adv_bev_fold_f32", _ptr(vi), _ptr(out), b, c, z, y, x, int(pool), _stream(vi))
    return out


def bev_fold_bwd(grad_out, shape, pool, mask=None):
    """the fold's backward; ``mask`` (the forward's input, a ReLU output) zeroes the gradient where it is <= 0"""
    g = _feat(grad_out, "grad_out")
    b, c, z, y, x = shape
    if tuple(g.shape) != (b, c * (y // pool), z, x):
        raise ValueError("grad_out must be [B, C * (Y // pool), Z, X]")
    if mask is not None:
        mask = _feat(mask, "mask")
        if tuple(mask.shape) != tuple(shape):
            raise ValueError("mask must be laid out like the fold's input")
    gv = torch.empty(tuple(shape), dtype=torch.float32, device=g.device)
    with _on(g):
        _lib.call("adv_bev_fold_bwd_f32", _ptr(g), None if mask is None else _ptr(mask), _ptr(gv), b, c, z, y, x, int(pool), _stream(g))
    return gv


class BevFold(torch.autograd.Function):
    """the bird's-eye-view fold of the 3D geometric volume (height pooled by ``pool`` and folded into the channels), forward and backward
    one HBM-bound pass each (csrc/volume.hip) instead of torch's pooling kernel + permuting copy and their two backward passes"""

    @staticmethod
    def forward(ctx, v, pool, mask_input=False):
        """``mask_input``: v is a ReLU output this fold alone consumes and its producer left the mask to us (relu="consumer"): the gradient
        returned is already multiplied by (v > 0)"""
        ctx.shape, ctx.pool = tuple(v.shape), int(pool)
        v = v.contiguous()
        ctx.save_for_backward(v if mask_input else None)
        return bev_fold(v, pool)

    @staticmethod
    def backward(ctx, grad_out):
        (v,) = ctx.saved_tensors
        return bev_fold_bwd(grad_out.contiguous(), ctx.shape, ctx.pool, mask=v), None, None


def sigmoid_focal_loss(logits, targets, gamma=2.0, alpha=0.25, want_grad=False):
    """logits [N,K], targets int32 [N] (0 = background, c = class c, < 0 = ignored) -> per-element loss [N,K] (and d loss / d logit)"""
    li = _feat(logits, "logits")
    if li.dim() != 2 or not (isinstance(targets, torch.Tensor) and targets.is_cuda and targets.dtype == torch.int32 and targets.is_contiguous()
                             and tuple(targets.shape) == (li.shape[0],)):
        raise ValueError("logits must be [N,K] float32 and targets [N] int32, contiguous, on the GPU")
    loss = torch.empty_like(li)
    grad = torch.empty_like(li) if want_grad else None
    with _on(li):
        _lib.call("adv_sigmoid_focal_loss_f32", _ptr(li), _ptr(targets), _ptr(loss), None if grad is None else _ptr(grad), int(li.shape[0]),
                  int(li.shape[1]), float(gamma), float(alpha), _stream(li))
    return (loss, grad) if want_grad else loss


class SigmoidFocalLoss(torch.autograd.Function):
    """sum of the per-element focal losses (the reduction the FCOS-style heads use before dividing by the positive count)"""

    @staticmethod
    def forward(ctx, logits, targets, gamma=2.0, alpha=0.25):
        loss, grad = sigmoid_focal_loss(logits.contiguous(), targets, gamma, alpha, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.sum()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None
