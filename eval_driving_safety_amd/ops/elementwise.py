"""element-wise helpers shared by the convolution modules (csrc/volume.hip).  Part of the ``ops`` package."""
from ._base import *       # noqa: F401,F403


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


def stem_pool(t, bias=None):
    """(maxpool3x3/s2/p1(relu(t + bias)), code) from the stem convolution's raw output ``t`` [B,C,H,W] in one pass (csrc/volume.hip); the code
    bytes (argmax position, or 15 where relu passes no gradient) are all the backward needs"""
    x = _feat(t, "t")
    if x.dim() != 4:
        raise ValueError("t must be [B,C,H,W]")
    b, c, h, w = x.shape
    bb = None if bias is None else _feat(bias, "bias")
    if bb is not None and tuple(bb.shape) != (c,):
        raise ValueError("bias must be [C]")
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, c, oh, ow), dtype=torch.float32, device=x.device)
    code = torch.empty((b, c, oh, ow), dtype=torch.uint8, device=x.device)
    if y.numel():
        with _on(x):
            _lib.call("adv_stem_pool_fwd_f32", _ptr(x), None if bb is None else _ptr(bb), _ptr(y), _ptr(code), b * c, c, h, w, _stream(x))
    return y, code


def stem_pool_bwd(grad_y, code, in_hw):
    """the gradient w.r.t. ``t`` [B,C,H,W] of ``stem_pool`` from the gradient w.r.t. its output and the code"""
    g = _feat(grad_y.contiguous(), "grad_y")
    if code.dtype != torch.uint8 or code.shape != g.shape or not code.is_contiguous() or code.device != g.device:
        raise ValueError("code must be the contiguous uint8 tensor stem_pool returned")
    b, c = g.shape[:2]
    h, w = int(in_hw[0]), int(in_hw[1])
    if tuple(g.shape[2:]) != ((h - 1) // 2 + 1, (w - 1) // 2 + 1):
        raise ValueError("grad_y does not belong to an input of %dx%d" % (h, w))
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=g.device)
    if out.numel():
        with _on(g):
            _lib.call("adv_stem_pool_bwd_f32", _ptr(g), _ptr(code), _ptr(out), b * c, h, w, _stream(g))
    return out


class StemPool(torch.autograd.Function):
    """F.max_pool2d(F.relu(t + bias[None, :, None, None]), 3, 2, 1) as one kernel forward and one backward (gradient w.r.t. t only: the
    bias is a folded batch-norm shift of a frozen network)"""

    @staticmethod
    def forward(ctx, t, bias=None):
        y, code = stem_pool(t, bias)
        ctx.save_for_backward(code)
        ctx.in_hw = tuple(t.shape[2:])
        return y

    @staticmethod
    def backward(ctx, grad_y):
        code, = ctx.saved_tensors
        return stem_pool_bwd(grad_y, code, ctx.in_hw), None


__all__ = [n for n in dir() if not n.startswith("__")]
