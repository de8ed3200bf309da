"""A small torch-module network with the upstream Stereo R-CNN's CALL SURFACE and MODULE NAMING (attack/Stereo-RCNN/stereo_rcnn.py):
nine inputs -> fifteen outputs, the six losses at positions 8..13 (:143-144,324-326); ``RCNN_layer0..2`` bottom-up with
torchvision-style Bottlenecks (conv1/bn1/conv2/bn2/conv3/bn3/downsample), ``RCNN_toplayer`` / ``RCNN_latlayer1`` / ``RCNN_smooth1``
top-down (:157-169), ``RCNN_roi_align = ROIAlign((7, 7), 1/16, 0)`` from ``model.roi_layers`` called with THREE arguments (:44,132-134)
and ``nms`` from the same package in the proposal step.  Not Stereo R-CNN: a stand-in for the tests, weights seeded."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from model.roi_layers import ROIAlign, nms


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


def _down(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(cout))


class resnet(nn.Module):
    def __init__(self, classes, num_layers=101, pretrained=False):
        super().__init__()
        assert num_layers == 101 and not pretrained
        self.classes = classes

    def create_architecture(self):
        g = torch.Generator().manual_seed(9)
        self.RCNN_layer0 = nn.Sequential(nn.Conv2d(3, 8, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(8), nn.ReLU(inplace=True),
                                         nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.RCNN_layer1 = nn.Sequential(Bottleneck(8, 4, 1, _down(8, 16, 1)), Bottleneck(16, 4))
        self.RCNN_layer2 = nn.Sequential(Bottleneck(16, 8, 2, _down(16, 32, 2)))
        self.RCNN_toplayer = nn.Conv2d(32, 16, kernel_size=1, stride=1, padding=0)
        self.RCNN_latlayer1 = nn.Conv2d(16, 16, kernel_size=1, stride=1, padding=0)
        self.RCNN_smooth1 = nn.Conv2d(16, 16, kernel_size=3, stride=1, padding=1)
        self.RCNN_roi_align = ROIAlign((7, 7), 1.0 / 16.0, 0)
        self.RCNN_top = nn.Sequential(nn.Conv2d(32, 12, kernel_size=7, stride=7, padding=0), nn.ReLU(True), nn.Conv2d(12, 6, kernel_size=1), nn.ReLU(True))
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan = m.weight[0].numel()
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
                    if m.bias is not None:
                        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                elif isinstance(m, nn.BatchNorm2d):          # trained statistics are not (0, 1): folding them must matter
                    m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))

    def pyramid(self, im):
        c2 = self.RCNN_layer1(self.RCNN_layer0(im / 64.0))                   # 16 x 1/4
        c3 = self.RCNN_layer2(c2)                                            # 32 x 1/8
        p3 = self.RCNN_toplayer(c3)
        up = self._upsample_add(p3, self.RCNN_latlayer1(c2))
        return self.RCNN_smooth1(up)                                         # 16 x 1/4

    def _upsample_add(self, x, y):           # the method name and meaning of the detector class this stands in for (top-down path)
        return F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=False) + y

    def forward(self, im_left, im_right, im_info, gt_l, gt_r, gt_m, gt_dim_orien, gt_kpts, num_boxes):
        dev = im_left.device
        p2_l, p2_r = self.pyramid(im_left), self.pyramid(im_right)
        n = 4
        cand = torch.tensor([[300., 250., 500., 400.], [310., 255., 505., 398.], [900., 260., 1100., 420.], [50., 300., 120., 380.]], device=dev)
        prior = torch.tensor([1.0, 0.9, 0.8, 0.01], device=dev)
        keep = nms(cand, prior, 0.99)                                        # the proposal step: nothing overlaps that much - all four, by score
        rois = torch.zeros(1, n, 5, device=dev)
        rois[0, :, 1:] = cand[keep]
        rois_right = rois.clone()
        rois_right[0, :, 1] -= 38.0          # the loader's right eye is the left one shifted by 38 network pixels
        rois_right[0, :, 3] -= 38.0
        scale = p2_l.size(2) / im_info[0][0]                                 # stereo_rcnn.py:129
        pooled = torch.cat((self.RCNN_roi_align(p2_l, rois.view(-1, 5), scale), self.RCNN_roi_align(p2_r, rois_right.view(-1, 5), scale)), 1)
        o = self.RCNN_top(pooled).flatten(1)                                 # [n, 6]
        box = gt_l.reshape(-1)[:4].mean() * 1e-4
        losses = [(o[:, k] * o[:, k]) * 0.1 + (p2_l[:, k] * p2_r[:, k]).mean() + box for k in range(6)]
        s = 0.5 + 0.4 * torch.tanh(o.mean())
        cls_prob = torch.stack([1 - s, s]).view(1, 1, 2).repeat(1, n, 1) * prior[keep].view(1, n, 1)
        bbox_pred = torch.zeros(1, n, 12, device=dev)
        dim = torch.zeros(1, n, 10, device=dev)
        g = torch.Generator().manual_seed(21)               # a fixed draw: two forward calls on the same input return the same tensors
        kpts = torch.rand(1, n, 4 * 28, generator=g).to(dev)
        lp, rp = torch.rand(1, n, 28, generator=g).to(dev), torch.rand(1, n, 28, generator=g).to(dev)
        return (rois, rois_right, cls_prob, bbox_pred, dim, kpts, lp, rp) + tuple(losses) + (None,)
