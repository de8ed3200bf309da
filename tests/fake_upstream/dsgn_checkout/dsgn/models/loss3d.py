import torch

from dsgn.layers import SigmoidFocalLoss


class RPN3DLoss:
    """focal classification term through the checkout's compiled operator + two plain terms; (total, cls, reg, centerness) as upstream"""

    def __init__(self, cfg):
        self.cfg = cfg
        self.cls_loss_func = SigmoidFocalLoss(2.0, 0.25)

    def __call__(self, bbox_cls, bbox_reg, bbox_centerness, targets, calib, calib_R, ious=None, labels_map=None):
        t = targets[0].bbox.sum() * 1e-3 + targets[0].box3d.sum() * 1e-3
        k = bbox_cls.shape[1]
        logits = bbox_cls.permute(0, 2, 3, 1).reshape(-1, k)
        n = logits.shape[0]
        idx = torch.arange(n, device=logits.device)
        labels = torch.where(idx % 5 == 0, torch.ones_like(idx), torch.zeros_like(idx))
        labels = torch.where(idx % 11 == 3, -torch.ones_like(idx), labels).int()                 # a few ignored locations
        cls = self.cls_loss_func(logits, labels) / max(1, int((idx % 5 == 0).sum()))
        reg, ctr = (bbox_reg * bbox_reg).mean(), bbox_centerness.mean() + t
        return cls + reg + ctr, cls, reg, ctr
