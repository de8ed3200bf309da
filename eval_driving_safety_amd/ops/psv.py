"""K7: the plane-sweep cost volume (csrc/psv.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)

# --------------------------------------------------------------------------------------------
# K7: plane-sweep cost volume (reached through the detector call, attack/DSGN/pgd_attack.py:308)
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


__all__ = [n for n in dir() if not n.startswith("__")]
