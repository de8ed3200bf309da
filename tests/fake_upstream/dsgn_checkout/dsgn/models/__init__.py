"""A small torch-module network with the upstream DSGN ``StereoNet``'s call signature and output dict (attack/DSGN/pgd_attack.py:215-222,308)
and the PSMNet / DSGN family's building blocks - ``convbn`` / ``convbn_3d`` = Sequential(conv, BatchNorm) helpers, a plane-sweep
concatenation volume, a 3D hourglass (stride-2 convolution down, transposed convolution up with a skip connection), a 3x3x3 score layer,
soft-argmin depth.  Not DSGN: a stand-in for the tests, weights seeded, batch-norm statistics non-trivial."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def convbn(cin, cout, k, stride, pad, dilation):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout))


def convbn_3d(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=k, padding=pad, stride=stride, bias=False), nn.BatchNorm3d(cout))


class feature_extraction(nn.Module):
    def __init__(self):
        super().__init__()
        self.firstconv = nn.Sequential(convbn(3, 8, 3, 2, 1, 1), nn.ReLU(inplace=True), convbn(8, 8, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.down = nn.Sequential(convbn(8, 8, 3, 2, 1, 1), nn.ReLU(inplace=True))
        self.dilated = nn.Sequential(convbn(8, 8, 3, 1, 1, 2), nn.ReLU(inplace=True))
        self.lastconv = nn.Sequential(convbn(8, 8, 3, 1, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(8, 4, kernel_size=1, padding=0, stride=1, bias=False))

    def forward(self, x):
        return self.lastconv(self.dilated(self.down(self.firstconv(x))))          # 4 x 1/4


class hourglass(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, 2 * c, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(2 * c, 2 * c, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(2 * c, c, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(c))

    def forward(self, x):
        return F.relu(self.conv5(self.conv2(self.conv1(x))) + x)


class StereoNet(nn.Module):
    PLANES = (0, 2, 4, 6, 8, 10, 12, 14)        # feature-pixel disparities of the eight planes

    def __init__(self, cfg=None):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.cfg = cfg
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True))
        self.hg = hourglass(8)
        self.classif1 = nn.Sequential(convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv3d(8, 1, kernel_size=3, padding=1, stride=1, bias=False))
        self.head = nn.Conv2d(8, 3, 1)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d)):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / m.weight[0].numel()) ** 0.5)
                    if m.bias is not None:
                        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                    m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))

    def forward(self, imgL, imgR, calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R=None):
        assert calibs_Proj_R is not None and len(calibs_fu) == imgL.shape[0]
        fl, fr = self.feature_extraction(imgL), self.feature_extraction(imgR)
        w = fr.shape[-1]
        cost = torch.stack([torch.cat([fl, fr if d == 0 else F.pad(fr, (d, 0))[..., :w]], 1) for d in self.PLANES], 2)     # [B,8,D,h,w]
        score = self.classif1(self.hg(self.dres0(cost)))[:, 0]                                                              # [B,D,h,w]
        prob = torch.softmax(score, 1)
        depth = (prob * (10.0 + 2.5 * torch.arange(len(self.PLANES), device=prob.device, dtype=prob.dtype)).view(1, -1, 1, 1)).sum(1, keepdim=True)
        depth = F.interpolate(depth, size=imgL.shape[2:], mode="bilinear", align_corners=False)
        out = self.head(torch.cat([fl, fr], dim=1))
        # eval mode: depth_preds is one [B,H,W] tensor that the scripts iterate over the batch dimension
        return {"depth_preds": depth.squeeze(1), "bbox_cls": out[:, 0], "bbox_reg": out[:, 1], "bbox_centerness": out[:, 2]}
