import torch
import torch.nn as nn
import torch.nn.functional as F


class feature_extraction(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 4, 3, stride=4, padding=1)
        self.conv2 = nn.Conv2d(4, 4, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class StereoNet(nn.Module):
    """same call signature and output dict as the upstream StereoNet (attack/DSGN/pgd_attack.py:215-222,308)"""

    def __init__(self, cfg=None):
        super().__init__()
        torch.manual_seed(5)
        self.cfg = cfg
        self.feature_extraction = feature_extraction()
        self.head = nn.Conv2d(8, 3, 1)

    def forward(self, imgL, imgR, calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R=None):
        assert calibs_Proj_R is not None and len(calibs_fu) == imgL.shape[0]
        fl, fr = self.feature_extraction(imgL), self.feature_extraction(imgR)
        both = torch.cat([fl, fr], dim=1)
        depth = 20.0 + 5.0 * torch.tanh(F.interpolate((fl * fr).sum(1, keepdim=True), size=imgL.shape[2:], mode="bilinear", align_corners=False))
        out = self.head(both)
        # eval mode: depth_preds is one [B,H,W] tensor that the scripts iterate over the batch dimension
        return {"depth_preds": depth.squeeze(1), "bbox_cls": out[:, 0], "bbox_reg": out[:, 1], "bbox_centerness": out[:, 2]}
