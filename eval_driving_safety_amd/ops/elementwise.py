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


__all__ = [n for n in dir() if not n.startswith("__")]
