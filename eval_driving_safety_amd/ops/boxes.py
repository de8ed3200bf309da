"""Box arithmetic of the proposal / target stage (csrc/boxes.hip): IoU rows with their maxima, the six stereo regression targets, decoding +
clipping of stereo proposals - each a chain of ~15 torch one-liners as one launch, the same float32 expressions in the same order.
Part of the ``ops`` package."""
from ._base import *       # noqa: F401,F403


def _boxes(t, name, cols=4):
    b = _feat(t.contiguous(), name)
    if b.dim() != 2 or b.shape[1] != cols:
        raise ValueError("%s must be [N,%d]" % (name, cols))
    return b


def box_iou_rows(a, b, want_matrix=True):
    """a [N,4] against b [M,4] (M >= 1), legacy +1 areas -> (iou [N,M] or None, best [N], arg [N] int64: the first maximum of each row)"""
    aa, bb = _boxes(a, "a"), _boxes(b, "b")
    n, m = aa.shape[0], bb.shape[0]
    if m < 1:
        raise ValueError("b must hold at least one box")
    iou = torch.empty((n, m), dtype=torch.float32, device=aa.device) if want_matrix else None
    best = torch.empty((n,), dtype=torch.float32, device=aa.device)
    arg = torch.empty((n,), dtype=torch.int64, device=aa.device)
    if n:
        with _on(aa):
            _lib.call("adv_box_iou_rows_f32", _ptr(aa), _ptr(bb), None if iou is None else _ptr(iou), _ptr(best), _ptr(arg), n, m, _stream(aa))
    return iou, best, arg


def box_encode6(src, gt_left, gt_right, arg, src_right=None):
    """[N,6]: (dx, dy, log dw, log dh) from src onto gt_left[arg] and (dx, log dw) from src_right (default: src) onto gt_right[arg]"""
    s, gl, gr = _boxes(src, "src"), _boxes(gt_left, "gt_left"), _boxes(gt_right, "gt_right")
    sr = None if src_right is None else _boxes(src_right, "src_right")
    if sr is not None and sr.shape != s.shape:
        raise ValueError("src_right must have src's shape")
    if gl.shape != gr.shape or gl.shape[0] < 1 or arg.dtype != torch.int64 or arg.shape != (s.shape[0],) or arg.device != s.device:
        raise ValueError("gt_left / gt_right must be [M,4] with M >= 1, arg int64 [N] on src's device")
    out = torch.empty((s.shape[0], 6), dtype=torch.float32, device=s.device)
    if s.shape[0]:
        ar = arg.contiguous()
        with _on(s):
            _lib.call("adv_box_encode6_f32", _ptr(s), None if sr is None else _ptr(sr), _ptr(gl), _ptr(gr), _ptr(ar), _ptr(out), s.shape[0], gl.shape[0], _stream(s))
    return out


def box_decode_stereo(anchors, deltas, width, height, min_size=0.0):
    """-> (left [N,4], right [N,4], big [N] int64): anchors moved by deltas[:, (0,1,2,3)] / deltas[:, (4,1,5,3)], clipped to the image;
    big = 1 where both widths and the left height reach ``min_size``"""
    a, d = _boxes(anchors, "anchors"), _boxes(deltas, "deltas", 6)
    if a.shape[0] != d.shape[0]:
        raise ValueError("one row of deltas per anchor")
    n = a.shape[0]
    left = torch.empty((n, 4), dtype=torch.float32, device=a.device)
    right = torch.empty((n, 4), dtype=torch.float32, device=a.device)
    big = torch.empty((n,), dtype=torch.int64, device=a.device)
    if n:
        with _on(a):
            _lib.call("adv_box_decode_stereo_f32", _ptr(a), _ptr(d), _ptr(left), _ptr(right), _ptr(big), n, float(width), float(height), float(min_size),
                      _stream(a))
    return left, right, big


def box_partition_stereo(left, right, big):
    """stable partition of the box pairs: those with ``big`` != 0 first, in order (what gathering with argsort(1 - big, stable=True) gives);
    -> (left', right', nvalid [1] int64 = how many are big; all of them - and nothing moved - if none is).  No read-back."""
    l, r = _boxes(left, "left"), _boxes(right, "right")
    if r.shape != l.shape or big.dtype != torch.int64 or big.shape != (l.shape[0],) or big.device != l.device or l.shape[0] < 1:
        raise ValueError("left / right [N,4] (N >= 1), big int64 [N] on their device")
    ol, orr = torch.empty_like(l), torch.empty_like(r)
    nvalid = torch.empty((1,), dtype=torch.int64, device=l.device)
    bg = big.contiguous()
    with _on(l):
        _lib.call("adv_box_partition_stereo_f32", _ptr(l), _ptr(r), _ptr(bg), _ptr(ol), _ptr(orr), _ptr(nvalid), l.shape[0], _stream(l))
    return ol, orr, nvalid


def box_sample_rois(keep, nvalid, left, right, gt_left, gt_right, n_rois):
    """the rois of one image without a read-back: candidates = ground-truth pairs (may be None), then left/right[keep[j]] for the valid
    entries of the padded list ``keep`` (0 <= keep[j] < nvalid: a prefix), sampled in order with replacement
    -> (rois_left [R,5], rois_right [R,5], left [R,4], right [R,4])"""
    l, r = _boxes(left, "left"), _boxes(right, "right")
    kp = keep.contiguous()
    if kp.dtype != torch.int64 or kp.dim() != 1 or kp.shape[0] < 1 or nvalid.dtype != torch.int64 or nvalid.numel() != 1:
        raise ValueError("keep int64 [K >= 1], nvalid int64 [1]")
    n_gt = 0 if gt_left is None else int(gt_left.shape[0])
    gl = gr = None
    if n_gt:
        gl, gr = _boxes(gt_left, "gt_left"), _boxes(gt_right, "gt_right")
    R = int(n_rois)
    rl = torch.empty((R, 5), dtype=torch.float32, device=l.device)
    rr = torch.empty((R, 5), dtype=torch.float32, device=l.device)
    ol = torch.empty((R, 4), dtype=torch.float32, device=l.device)
    orr = torch.empty((R, 4), dtype=torch.float32, device=l.device)
    nv = nvalid.contiguous()
    with _on(l):
        _lib.call("adv_box_sample_rois_f32", _ptr(kp), kp.shape[0], _ptr(nv), _ptr(l), _ptr(r), None if gl is None else _ptr(gl), None if gr is None else _ptr(gr),
                  n_gt, R, _ptr(rl), _ptr(rr), _ptr(ol), _ptr(orr), _stream(l))
    return rl, rr, ol, orr


class MaskedLossMean(torch.autograd.Function):
    """sum_i weight[i] * l(pred[i], target[i]) / max(scale * sum(weight), 1) - l = smooth-L1 (beta 1) summed over the columns of row i, or
    (``bce``) binary cross-entropy with logits - as two launches forward and one backward (torch: six or seven + four or five).  The
    gradient w.r.t. ``pred`` is torch's, bit for bit; the value is summed in this kernel's own fixed order (last-bit differences).
    ``weight`` [N] and ``target`` are constants."""

    @staticmethod
    def forward(ctx, pred, target, weight, scale, bce):
        p = _feat(pred.contiguous(), "pred")
        t = _feat(target.detach().contiguous(), "target")
        w = _feat(weight.detach().contiguous(), "weight")
        rows = p.shape[0]
        k = 1 if p.dim() == 1 else p.shape[1]
        if p.dim() not in (1, 2) or t.shape != p.shape or w.shape != (rows,) or rows < 1 or (bce and k != 1):
            raise ValueError("pred / target [N] or [N,K], weight [N]; the cross-entropy form takes [N]")
        out2 = torch.empty((2,), dtype=torch.float32, device=p.device)
        work = torch.empty((int(_lib.load().adv_masked_loss_workspace_floats()),), dtype=torch.float32, device=p.device)
        with _on(p):
            _lib.call("adv_masked_loss_f32", _ptr(p), _ptr(t), _ptr(w), _ptr(out2), _ptr(work), rows, k, float(scale), int(bool(bce)), _stream(p))
        ctx.save_for_backward(p, t, w, out2)
        ctx.meta = (rows, k, bool(bce))
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        p, t, w, out2 = ctx.saved_tensors
        rows, k, bce = ctx.meta
        gg = _feat(g.contiguous().reshape(1), "grad")
        grad = torch.empty_like(p)
        with _on(p):
            _lib.call("adv_masked_loss_bwd_f32", _ptr(p), _ptr(t), _ptr(w), _ptr(out2), _ptr(gg), _ptr(grad), rows, k, int(bce), _stream(p))
        return grad, None, None, None, None


def masked_smooth_l1_mean(pred, target, weight, scale):
    """((smooth_l1(pred, target) * weight[:, None]).sum() / (scale * weight.sum()).clamp(min=1)) as ops.MaskedLossMean"""
    return MaskedLossMean.apply(pred, target, weight, scale, False)


def masked_bce_mean(logits, label, keep):
    """((bce_with_logits(logits, label) * keep).sum() / keep.sum().clamp(min=1)) as ops.MaskedLossMean"""
    return MaskedLossMean.apply(logits, label, keep, 1.0, True)


class ObjectiveChain(torch.autograd.Function):
    """sum_k (terms[k] * exp(-u[k]) + u[k]) in the order attack/Stereo-RCNN/pgd_attack.py:165-171 adds it (one launch; the script's loop is
    ~45 scalar launches and ~30 more backward), differentiable w.r.t. ``terms`` [n]; u [n] is a constant of the attack"""

    @staticmethod
    def forward(ctx, terms, u):
        t, uu = _feat(terms.contiguous(), "terms"), _feat(u.detach().contiguous(), "u")
        if t.dim() != 1 or uu.shape != t.shape or not 1 <= t.shape[0] <= 64:
            raise ValueError("terms and u must be [n], 1 <= n <= 64")
        loss = torch.empty((1,), dtype=torch.float32, device=t.device)
        w = torch.empty_like(t)
        with _on(t):
            _lib.call("adv_objective_chain_f32", _ptr(t), _ptr(uu), _ptr(loss), _ptr(w), t.shape[0], _stream(t))
        ctx.save_for_backward(w)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        w, = ctx.saved_tensors
        return g * w, None


class RpnHeadPack(torch.autograd.Function):
    """(scores [N], deltas [N,6]) of ALL pyramid levels from the RPN head's per-level outputs [B, 7A, H_l, W_l] (A objectness maps, then six
    regression maps per anchor), N = sum_l B * H_l * W_l * A in (level, image, pixel, anchor) order - what the per-level
    ``permute(0, 2, 3, 1).reshape`` of the two slices and the two concatenations give, with ``bounded`` deltas = 0.5 * tanh(raw): one launch
    per level each way instead of ~12, the same values and the same gradient (tests/test_boxes.py)"""

    @staticmethod
    def forward(ctx, anchors, bounded, *heads):
        hs = [_feat(h.contiguous(), "head") for h in heads]
        a = int(anchors)
        if not hs or any(h.dim() != 4 or h.shape[1] != 7 * a for h in hs):
            raise ValueError("every head output must be [B, 7 * anchors, H, W]")
        counts = [h.shape[0] * h.shape[2] * h.shape[3] * a for h in hs]
        n = sum(counts)
        scores = torch.empty((n,), dtype=torch.float32, device=hs[0].device)
        deltas = torch.empty((n, 6), dtype=torch.float32, device=hs[0].device)
        off = 0
        with _on(hs[0]):
            for h, c in zip(hs, counts):
                _lib.call("adv_rpn_pack_fwd_f32", _ptr(h), scores.data_ptr() + 4 * off, deltas.data_ptr() + 24 * off, h.shape[0], a, h.shape[2] * h.shape[3],
                          int(bool(bounded)), _stream(h))
                off += c
        ctx.save_for_backward(*hs)
        ctx.meta = (a, bool(bounded), counts)
        return scores, deltas

    @staticmethod
    def backward(ctx, g_scores, g_deltas):
        a, bounded, counts = ctx.meta
        hs = ctx.saved_tensors
        gs = None if g_scores is None else g_scores.contiguous()
        gd = None if g_deltas is None else g_deltas.contiguous()
        outs, off = [], 0
        with _on(hs[0]):
            for h, c in zip(hs, counts):
                g = torch.empty_like(h)
                _lib.call("adv_rpn_pack_bwd_f32", _ptr(h), None if gs is None else gs.data_ptr() + 4 * off, None if gd is None else gd.data_ptr() + 24 * off,
                          _ptr(g), h.shape[0], a, h.shape[2] * h.shape[3], int(bounded), _stream(h))
                outs.append(g)
                off += c
        return (None, None) + tuple(outs)


__all__ = [n for n in dir() if not n.startswith("__")]
