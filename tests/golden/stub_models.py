"""Stand-in detector objects with the UPSTREAM call signatures, shared by tests/golden/make_golden.py (which runs the reference's own
objective statements around them) and tests/test_adapters.py (which runs adapters.DsgnAdapter / StereoRcnnAdapter around the same
objects): what is pinned is the objective GLUE of attack/DSGN/pgd_attack.py:300-336 and attack/Stereo-RCNN/pgd_attack.py:151-174 -
loss weights, masks, the uncertainty weighting, what backward() leaves in the image gradients - not a detector.

Test infrastructure written for this repository: no reference code, nothing under eval_driving_safety_amd/ imports it."""
import torch
import torch.nn as nn


class StubDsgn(nn.Module):
    """``model(imgL, imgR, calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R=...) -> dict`` (pgd_attack.py:308): seeded 3x3
    convolutions; ``depth_preds`` is a [B,H,W] tensor, which is what the script's ``o[mask[0]]`` (:316) requires of eval mode"""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.f = nn.Conv2d(3, 4, 3, padding=1)
        self.d = nn.Conv2d(8, 1, 3, padding=1)
        self.h = nn.Conv2d(8, 3, 3, padding=1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, imgL, imgR, fu, baseline, proj, calibs_Proj_R=None):
        both = torch.cat([torch.tanh(self.f(imgL)), torch.tanh(self.f(imgR))], 1)
        scale = (fu.to(both.dtype) * baseline.to(both.dtype) / 20.0).view(-1, 1, 1)
        depth = 20.0 + scale * self.d(both)[:, 0]                       # [B,H,W], around the valid depth range
        head = self.h(both) + 0.01 * (proj.to(both.dtype).sum() - calibs_Proj_R.to(both.dtype).sum())
        return {"depth_preds": depth, "bbox_cls": head[:, :1], "bbox_reg": head[:, 1:2], "bbox_centerness": head[:, 2:]}


class StubRpn3dLoss:
    """``RPN3DLoss(cfg)(bbox_cls, bbox_reg, bbox_centerness, targets, calib, calib_R, ious=, labels_map=) -> (total, cls, reg, ctr)`` (:324-326)"""

    def __init__(self, cfg):
        self.k = float(getattr(cfg, "stub_gain", 1.0))

    def __call__(self, cls, reg, ctr, targets, calib, calib_R, ious=None, labels_map=None):
        t = targets[0]                                                   # a [H,W] tensor standing in for the label set
        l_cls = torch.nn.functional.binary_cross_entropy_with_logits(cls[:, 0], (t > 0).to(cls.dtype).expand_as(cls[:, 0]))
        l_reg = torch.nn.functional.smooth_l1_loss(reg[:, 0], t.expand_as(reg[:, 0]))
        l_ctr = (ctr * ctr).mean() * ious
        return self.k * (l_cls + l_reg + l_ctr), l_cls, l_reg, l_ctr


class StubStereoRcnn(nn.Module):
    """nine inputs -> the fifteen outputs of stereo_rcnn.py:324-326, the six losses (positions 8..13) as tensors of several elements
    so that the script's ``.mean()`` matters"""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.f = nn.Conv2d(3, 6, 3, padding=1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)

    def forward(self, left, right, im_info, gl, gr, gm, gdo, gk, nb):
        a, b = self.f(left / 64.0), self.f(right / 64.0)
        s = im_info[0, 2]
        losses = [(a[:, 0] * b[:, 0]).mean(dim=1) * s, (a[:, 1] - b[:, 1]).abs().mean(dim=2), (a[:, 2] ** 2).mean(dim=(1, 2)),
                  torch.nn.functional.softplus(b[:, 3]).mean(dim=1), (a[:, 4] * gl[0, 0, 0]).mean(dim=2), torch.tanh(a[:, 5] + b[:, 5]).mean(dim=1)]
        rois = torch.zeros(1, 4, 5)
        return (rois, rois, None, None, None, None, None, None) + tuple(losses) + (None,)
