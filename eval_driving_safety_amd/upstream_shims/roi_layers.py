"""Drop-in for the upstream Stereo R-CNN package ``model.roi_layers`` - the one the reference's own files import:

    attack/Stereo-RCNN/stereo_rcnn.py:18        from model.roi_layers import ROIAlign
    attack/Stereo-RCNN/stereo_rcnn.py:44-45     ROIAlign((cfg.POOLING_SIZE, cfg.POOLING_SIZE), 1.0/16.0, 0)        (and POOLING_SIZE*2)
    attack/Stereo-RCNN/stereo_rcnn.py:132-134   self.RCNN_roi_align(feat_maps[i], rois[idx_l], scale)             (THREE arguments)
    attack/Stereo-RCNN/{pgd_attack,patch_attack,predict_and_save_pgd,predict_and_save_patch}.py:25-27   from model.roi_layers import nms
    attack/Stereo-RCNN/predict_and_save_pgd.py:300   keep = nms(cls_boxes_left[order, :], cls_scores[order], cfg.TEST.NMS)

Upstream that package is a compiled CUDA extension (``model._C``), which does not exist on an MI355X.  ``upstream_shims.install()``
registers THIS module as ``model.roi_layers`` before the checkout's model code is imported, so the upstream network runs its RoI path
on csrc/roi.hip: ``adv_roi_align_fwd_f32`` / the deterministic ``adv_roi_align_bwd_f32`` (no float atomics: two runs give the same
gradient bits, unlike the upstream backward) and ``adv_nms_f32`` (bit-exact kept indices).  Same names, argument order and meaning.
"""
import torch
import torch.nn as nn

__all__ = ["ROIAlign", "roi_align", "nms"]


def _scale(v):
    return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio=0):
    """functional form (upstream ``roi_align = _ROIAlign.apply``): input [B,C,H,W], rois [K,5] = (batch index, x1, y1, x2, y2)"""
    from .. import ops
    if not input.is_cuda:
        raise RuntimeError("model.roi_layers (libadvengine shim): no CPU path - the features must live on the ROCm device")
    size = (int(output_size), int(output_size)) if isinstance(output_size, int) else (int(output_size[0]), int(output_size[1]))
    rois = rois.to(device=input.device, dtype=torch.float32)
    return ops.RoIAlign.apply(input.float(), rois, size, _scale(spatial_scale), int(sampling_ratio))


class ROIAlign(nn.Module):
    """``ROIAlign(output_size, spatial_scale, sampling_ratio)``; ``forward(input, rois[, spatial_scale])`` - the reference passes the
    pyramid level's scale per call (stereo_rcnn.py:129,132-134), overriding the constructor's 1/16"""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois, spatial_scale=None):
        return roi_align(input, rois, self.output_size, self.spatial_scale if spatial_scale is None else spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s) [libadvengine]" % (type(self).__name__, self.output_size, self.spatial_scale,
                                                                                             self.sampling_ratio)


def nms(dets, scores, thresh):
    """``nms(boxes [N,4], scores [N], thresh) -> kept indices`` (int64, in descending-score order; IoU with the legacy +1 areas,
    suppressed when IoU > thresh) - the upstream contract.  The boxes need not be sorted (the reference sorts before calling,
    predict_and_save_pgd.py:295-300; the proposal layer does not): a STABLE descending sort by score fixes the order of ties, which
    the upstream CUDA sort leaves open, so equal inputs give equal indices on every run."""
    from .. import ops
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    if not dets.is_cuda:
        raise RuntimeError("model.roi_layers.nms (libadvengine shim): no CPU path")
    scores = scores.reshape(-1)
    order = torch.sort(scores, descending=True, stable=True)[1]
    keep = ops.nms(dets[order, :4].float().contiguous(), scores[order].contiguous(), _scale(thresh))
    return order[keep]


def __getattr__(name):
    if name in ("ROIPool", "roi_pool"):
        raise AttributeError("model.roi_layers.%s: the libadvengine shim provides ROIAlign / roi_align / nms - what the reference's files "
                             "import (cfg.POOLING_MODE 'align'); RoI max-pooling has no kernel here" % name)
    raise AttributeError(name)
