"""RoIAlign forward / ordered backward, pyramid pooling, NMS (csrc/roi.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)

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


def roi_gout_channel_last(grad_out):
    """grad_out [R,C,PH,PW] -> its channel-last copy (flat, channels padded to 8) for ``roi_align_bwd(..., gcl=)``: made once when several
    feature maps were pooled with the same roi list"""
    g = _feat(grad_out, "grad_out")
    if g.dim() != 4:
        raise ValueError("grad_out must be [R,C,PH,PW]")
    r, c, ph, pw = g.shape
    gcl = torch.empty((max(4, int(_lib.load().adv_roi_gout_channel_last_floats(c, r, ph, pw))),), dtype=torch.float32, device=g.device)
    if r:
        with _on(g):
            _lib.call("adv_roi_gout_channel_last_f32", _ptr(g), _ptr(gcl), c, r, ph, pw, _stream(g))
    return gcl


def roi_align_bwd_cl(gcl, pooled_hw, rois, feat_shape, spatial_scale, sampling_ratio=0):
    """``roi_align_bwd`` from the channel-last copy of grad_out (``roi_gout_channel_last``): the same bits"""
    r = _feat(rois, "rois")
    b, c, h, w = feat_shape
    ph, pw = pooled_hw
    if gcl.dtype != torch.float32 or gcl.dim() != 1 or gcl.numel() < int(_lib.load().adv_roi_gout_channel_last_floats(c, r.shape[0], ph, pw)):
        raise ValueError("gcl is not the channel-last copy of a [R,%d,%d,%d] gradient" % (c, ph, pw))
    gf = torch.empty((b, c, h, w), dtype=torch.float32, device=gcl.device)
    work = torch.empty((max(1, int(_lib.load().adv_roi_align_bwd_workspace_ints(b, c, h, w, r.shape[0], ph, pw))),), dtype=torch.int32, device=gcl.device)
    with _on(gcl):
        _lib.call("adv_roi_align_bwd_cl_f32", _ptr(gcl), _ptr(r), _ptr(gf), b, c, h, w, r.shape[0], ph, pw, float(spatial_scale), int(sampling_ratio),
                  _ptr(work), _stream(gcl))
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
        # the per-level roi lists in five launches whatever the number of levels (a clone + compare + select + column write per level before)
        levels = torch.arange(len(feats), device=r.device)[:, None]
        mine = r.unsqueeze(0).repeat(len(feats), 1, 1)
        mine[:, :, 0] = torch.where(owner[None, :] == levels, r[None, :, 0], -1.0)
        for l, f in enumerate(feats):
            roi_align(f.contiguous(), mine[l], pooled, scales[l], sampling_ratio, out=out)
        ctx.save_for_backward(mine)
        ctx.meta = ([tuple(f.shape) for f in feats], tuple(scales), sampling_ratio)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        shapes, scales, sr = ctx.meta
        g = grad_out.contiguous()
        mine, = ctx.saved_tensors
        gcl = roi_gout_channel_last(g)                  # one transposition for all levels (each level's launch made its own before)
        hw = tuple(g.shape[2:])
        grads = tuple(roi_align_bwd_cl(gcl, hw, mine[l], shapes[l], scales[l], sr) if ctx.needs_input_grad[5 + l] else None for l in range(len(shapes)))
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


__all__ = [n for n in dir() if not n.startswith("__")]
