import torch
import torch.nn as nn
import torch.nn.functional as F


class resnet(nn.Module):
    """nine inputs -> fifteen outputs, the six losses at positions 8..13 (attack/Stereo-RCNN/stereo_rcnn.py:143-144,324-326)"""

    def __init__(self, classes, num_layers=101, pretrained=False):
        super().__init__()
        assert num_layers == 101 and not pretrained
        self.classes = classes

    def create_architecture(self):
        torch.manual_seed(9)
        self.c1 = nn.Conv2d(3, 4, 7, stride=16, padding=3)
        self.c2 = nn.Conv2d(8, 6, 1)

    def forward(self, im_left, im_right, im_info, gt_l, gt_r, gt_m, gt_dim_orien, gt_kpts, num_boxes):
        f = torch.cat([self.c1(im_left / 64.0), self.c1(im_right / 64.0)], 1)
        o = self.c2(F.relu(f))
        box = gt_l.reshape(-1)[:4].mean() * 1e-4
        losses = [(o[:, k] * o[:, k]).mean(dim=(1, 2)) + box for k in range(6)]
        n = 4
        rois = torch.zeros(1, n, 5, device=im_left.device)
        rois[0, :, 1:] = torch.tensor([[300., 250., 500., 400.], [310., 255., 505., 398.], [900., 260., 1100., 420.], [50., 300., 120., 380.]],
                                      device=im_left.device)
        s = torch.sigmoid(o.mean(dim=(1, 2, 3)))
        cls_prob = torch.stack([1 - s, s], 1).unsqueeze(1).repeat(1, n, 1) * torch.tensor([[[1.0], [0.9], [0.8], [0.01]]], device=im_left.device)
        bbox_pred = torch.zeros(1, n, 12, device=im_left.device)
        dim = torch.zeros(1, n, 10, device=im_left.device)
        kpts = torch.rand(1, n, 4 * 28, device=im_left.device)
        lp, rp = torch.rand(1, n, 28, device=im_left.device), torch.rand(1, n, 28, device=im_left.device)
        rois_right = rois.clone()
        rois_right[0, :, 1] -= 38.0          # the loader's right eye is the left one shifted by 38 network pixels
        rois_right[0, :, 3] -= 38.0
        return (rois, rois_right, cls_prob, bbox_pred, dim, kpts, lp, rp) + tuple(losses) + (None,)
