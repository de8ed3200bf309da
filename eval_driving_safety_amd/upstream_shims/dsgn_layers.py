"""Drop-in for the upstream DSGN package of operator WRAPPERS - ``dsgn.layers`` (``BuildCostVolume``, ``SigmoidFocalLoss``, ``nms``,
``ROIAlign``), the Python face of the checkout's compiled ``dsgn._C`` - for a checkout whose own wrappers cannot be imported on an MI355X.
``upstream_shims.ext_C`` replaces the extension itself and is what ``install("dsgn")`` registers by default; this module is the
alternative one level up (``install(table={"dsgn.layers": "dsgn_layers", ...})``).  Reached from the reference at

    attack/DSGN/pgd_attack.py:308   model(imgL, imgR, calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R=...)   -> BuildCostVolume
    attack/DSGN/pgd_attack.py:324   RPN3DLoss(cfg)(bbox_cls, bbox_reg, bbox_centerness, targets, calib, calib_R, ...)  -> SigmoidFocalLoss
    attack/DSGN/pgd_attack.py:220   make_fcos3d_postprocessor(cfg)(...)                                                 -> nms

Class names, constructor and call signatures follow the maskrcnn-benchmark layer package DSGN derives from [UPSTREAM-UNVERIFIED]."""
import torch
import torch.nn as nn

from . import ext_C
from .roi_layers import ROIAlign, nms, roi_align

__all__ = ["BuildCostVolume", "build_cost_volume", "SigmoidFocalLoss", "sigmoid_focal_loss_cuda", "ROIAlign", "roi_align", "nms"]

IS_LIBADVENGINE_SHIM = True


class _BuildCostVolume(torch.autograd.Function):
    """cost[b, :C, d] = left, cost[b, C:, d] = right displaced by shift[b, d] feature pixels (zero where the source column is outside the
    image); gradients to the two feature maps, summed over the planes in plane order - the shifts are constants"""

    @staticmethod
    def forward(ctx, left, right, shift):
        ctx.save_for_backward(shift)
        return ext_C.build_cost_volume_forward(left, right, shift)

    @staticmethod
    def backward(ctx, grad_cost):
        (shift,) = ctx.saved_tensors
        gl, gr = ext_C.build_cost_volume_backward(grad_cost, shift)
        return gl, gr, None


build_cost_volume = _BuildCostVolume.apply


class BuildCostVolume(nn.Module):
    """``BuildCostVolume()(left [B,C,H,W], right [B,C,H,W], shift [B,D] or [D]) -> [B,2C,D,H,W]``"""

    def forward(self, left, right, shift):
        return build_cost_volume(left, right, shift)

    def __repr__(self):
        return "BuildCostVolume() [libadvengine]"


class _SigmoidFocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        loss, grad = ext_C._focal(logits, targets, logits.shape[1], gamma, alpha, True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (grad,) = ctx.saved_tensors
        return grad * d_loss, None, None, None


sigmoid_focal_loss_cuda = _SigmoidFocalLoss.apply


class SigmoidFocalLoss(nn.Module):
    """``SigmoidFocalLoss(gamma, alpha)(logits [N,K], targets [N]) -> scalar`` (the sum over all elements, as upstream)"""

    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        return sigmoid_focal_loss_cuda(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return "%s(gamma=%s, alpha=%s) [libadvengine]" % (type(self).__name__, self.gamma, self.alpha)
