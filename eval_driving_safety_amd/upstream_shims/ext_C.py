"""Drop-in for the COMPILED extension module of the upstream checkouts - ``dsgn._C`` (DSGN) and ``model._C`` (Stereo R-CNN), both built
from a ``csrc/`` of CUDA sources in the maskrcnn-benchmark style - so that the checkout's own Python wrappers (``dsgn/layers/*.py``,
``model/roi_layers/*.py``) run UNCHANGED on an MI355X, with nothing hipified:

    upstream symbol (flat functions on torch tensors)                         libadvengine entry point (include/advengine.h)
    build_cost_volume_forward(left, right, shift) -> cost [B,2C,D,H,W]        adv_psv_build_f32 (int shifts) / adv_psv_build_lerp_f32 (float)
    build_cost_volume_backward(grad_cost, shift) -> (grad_left, grad_right)   adv_psv_build_bwd_f32 / adv_psv_build_lerp_bwd_f32
    sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha)     adv_sigmoid_focal_loss_f32 (loss)
    sigmoid_focalloss_backward(logits, targets, d_losses, K, gamma, alpha)    adv_sigmoid_focal_loss_f32 (d loss / d logit) x d_losses
    nms(dets [N,4], scores [N], threshold) -> kept indices (int64)            adv_nms_f32 behind a stable descending sort
    roi_align_forward(input, rois, scale, ph, pw, sampling_ratio)             adv_roi_align_fwd_f32
    roi_align_backward(grad, rois, scale, ph, pw, B, C, H, W, sampling_ratio) adv_roi_align_bwd_f32 (fixed-order gather, no atomics)

These are what the detector calls behind ``attack/DSGN/pgd_attack.py:308`` (``StereoNet.forward``: the plane-sweep cost volume), ``:324``
(``RPN3DLoss``: the focal term) and ``:220`` (``make_fcos3d_postprocessor``: box NMS) - SURVEY 2.2.  The upstream sources are not in the
reference tree, so the symbol names and argument orders above are [UPSTREAM-UNVERIFIED]: they follow the maskrcnn-benchmark extension
that both checkouts derive from, and ``upstream_shims.install(table=...)`` lets a user put this module (or ``dsgn_layers``, one level up)
under whatever names their checkout imports.  No CPU path: CPU tensors raise."""
import torch

__all__ = ["build_cost_volume_forward", "build_cost_volume_backward", "sigmoid_focalloss_forward", "sigmoid_focalloss_backward", "nms",
           "roi_align_forward", "roi_align_backward"]

IS_LIBADVENGINE_SHIM = True


def _ops(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s (libadvengine shim): no CPU path - the tensors must live on the ROCm device" % what)
    from .. import ops
    return ops


def _shift_arg(shift, b, dev):
    """per-plane disparities as the kernels take them: [B,D], int32 (whole feature pixels) or float32 (fractional: linear interpolation).
    A 1-D [D] tensor is shared by the batch; integer dtypes other than int32 are converted, float64 / float16 become float32."""
    s = shift.to(dev)
    if s.dim() == 1:
        s = s.unsqueeze(0).expand(b, -1)
    if s.dim() != 2 or s.shape[0] != b:
        raise ValueError("shift must be [D] or [B,D] (one disparity per depth plane), got %s for a batch of %d" % (tuple(shift.shape), b))
    s = s.to(torch.float32) if s.is_floating_point() else s.to(torch.int32)
    return s.contiguous()


def build_cost_volume_forward(left, right, shift):
    ops = _ops(left, "build_cost_volume_forward")
    l, r = left.float().contiguous(), right.float().contiguous()
    s = _shift_arg(shift, l.shape[0], l.device)
    return ops.psv_build_lerp(l, r, s) if s.is_floating_point() else ops.psv_build(l, r, s)


def build_cost_volume_backward(grad_cost, shift):
    ops = _ops(grad_cost, "build_cost_volume_backward")
    g = grad_cost.float().contiguous()
    s = _shift_arg(shift, g.shape[0], g.device)
    return ops.psv_build_lerp_bwd(g, s) if s.is_floating_point() else ops.psv_build_bwd(g, s)


def _focal(logits, targets, num_classes, gamma, alpha, want_grad):
    ops = _ops(logits, "sigmoid_focalloss")
    li = logits.float().contiguous()
    if li.dim() != 2 or int(num_classes) != li.shape[1]:
        raise ValueError("logits must be [N, num_classes]; got %s with num_classes %s" % (tuple(logits.shape), num_classes))
    t = targets.to(device=li.device, dtype=torch.int32).reshape(-1).contiguous()
    return ops.sigmoid_focal_loss(li, t, float(gamma), float(alpha), want_grad=want_grad)


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    """per-element losses [N,K]; targets: 0 = background, c = class c (logit column c - 1), negative = ignored"""
    return _focal(logits, targets, num_classes, gamma, alpha, False)


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    _, grad = _focal(logits, targets, num_classes, gamma, alpha, True)
    return grad * d_losses.to(grad.dtype)


def nms(dets, scores, threshold):
    from . import roi_layers
    return roi_layers.nms(dets, scores, threshold)


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    ops = _ops(input, "roi_align_forward")
    return ops.roi_align(input.float().contiguous(), rois.to(device=input.device, dtype=torch.float32).contiguous(),
                         (int(pooled_height), int(pooled_width)), float(spatial_scale), int(sampling_ratio))


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio):
    ops = _ops(grad, "roi_align_backward")
    if tuple(grad.shape[2:]) != (int(pooled_height), int(pooled_width)):
        raise ValueError("grad must be [R,C,%d,%d]" % (pooled_height, pooled_width))
    return ops.roi_align_bwd(grad.float().contiguous(), rois.to(device=grad.device, dtype=torch.float32).contiguous(),
                             (int(batch_size), int(channels), int(height), int(width)), float(spatial_scale), int(sampling_ratio))


def __getattr__(name):
    if name.startswith("roi_pool") or name.startswith("deform"):
        raise AttributeError("_C.%s: the libadvengine shim provides the cost volume, focal loss, NMS and RoIAlign entry points - what the "
                             "detectors of attack/DSGN and attack/Stereo-RCNN reach; %s has no kernel here" % (name, name))
    raise AttributeError(name)
