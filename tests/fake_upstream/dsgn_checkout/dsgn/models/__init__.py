"""A small torch-module network with the upstream DSGN ``StereoNet``'s call signature and output dict (attack/DSGN/pgd_attack.py:215-222,308)
and the PSMNet / DSGN family's building blocks - ``convbn`` / ``convbn_3d`` = Sequential(conv, BatchNorm) helpers, a plane-sweep
concatenation volume built by the checkout's COMPILED operator (``from dsgn.layers import BuildCostVolume`` - importable only where a
``dsgn._C`` exists), a 3D hourglass (stride-2 convolution down, transposed convolution up with a skip connection), a 3x3x3 score layer,
depth regression written the PSMNet way (``F.interpolate(trilinear)`` -> ``F.softmax`` -> a ``disparityregression`` module), the plane-sweep
features resampled into a 3D geometric volume with ``F.grid_sample`` on a grid computed from the projection matrix, a voxel convolution,
the height axis folded into channels, a bird's-eye-view head.  Not DSGN: a stand-in for the tests, weights seeded, batch-norm statistics
non-trivial."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from dsgn.layers import BuildCostVolume


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


class disparityregression(nn.Module):
    def forward(self, x, depth):
        return torch.sum(x * depth[None, :, None, None], 1)


class StereoNet(nn.Module):
    DEPTHS = (10.0, 12.5, 15.0, 17.5, 20.0, 22.5, 25.0, 27.5)        # metres: the eight planes of the sweep
    UP_PLANES = 32                                                     # planes of the up-sampled cost volume (4 x: ``maxdisp / downsample``)
    VOXELS = (6, 4, 10)                                                # (Z, Y, X) cells of the 3D geometric volume
    RANGE = ((-8.0, 8.0), (-1.0, 3.0), (10.0, 27.5))                   # metres: X, Y, Z extent of that volume

    def __init__(self, cfg=None):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.cfg = cfg
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True))
        self.hg = hourglass(8)
        self.classif1 = nn.Sequential(convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv3d(8, 1, kernel_size=3, padding=1, stride=1, bias=False))
        self.build_cost = BuildCostVolume()
        self.dispregression = disparityregression()
        self.register_buffer("depth", torch.linspace(self.DEPTHS[0], self.DEPTHS[-1], self.UP_PLANES))
        self.voxel_conv = nn.Sequential(convbn_3d(8, 8, 3, 1, 1), nn.ReLU(inplace=True))
        self.bev_conv = nn.Sequential(convbn(8 * self.VOXELS[1], 16, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.head = nn.Conv2d(16, 3, 1)
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

    def _voxel_grid(self, calibs_Proj, image_hw, dev):
        """normalised sampling coordinates [B,Z,Y,X,3] = (u, v, plane) of every voxel centre: projected with the left camera matrix, the
        depth axis linear in metres over the sweep's planes (a function of the calibration only)"""
        (x0, x1), (y0, y1), (z0, z1) = self.RANGE
        nz, ny, nx = self.VOXELS
        zs = z0 + (z1 - z0) * (torch.arange(nz, dtype=torch.float32) + 0.5) / nz
        ys = y0 + (y1 - y0) * (torch.arange(ny, dtype=torch.float32) + 0.5) / ny
        xs = x0 + (x1 - x0) * (torch.arange(nx, dtype=torch.float32) + 0.5) / nx
        zz, yy, xx = torch.meshgrid(zs, ys, xs, indexing="ij")
        pts = torch.stack([xx, yy, zz, torch.ones_like(xx)], -1)                                     # [Z,Y,X,4]
        P = torch.as_tensor(calibs_Proj, dtype=torch.float32).cpu()                                  # [B,3,4] (DataParallel may have moved it)
        uvw = torch.einsum("bij,zyxj->bzyxi", P, pts)
        u, v = uvw[..., 0] / uvw[..., 2], uvw[..., 1] / uvw[..., 2]
        h, w = image_hw
        d = (zz - self.DEPTHS[0]) / (self.DEPTHS[-1] - self.DEPTHS[0])
        grid = torch.stack([2 * u / (w - 1) - 1, 2 * v / (h - 1) - 1, (2 * d - 1).expand_as(u)], -1)
        return grid.to(dev).contiguous()

    def forward(self, imgL, imgR, calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R=None):
        assert calibs_Proj_R is not None and len(calibs_fu) == imgL.shape[0]
        fl, fr = self.feature_extraction(imgL), self.feature_extraction(imgR)
        # per-plane disparity in FEATURE pixels: fu * baseline / depth / 4 (fractional: the operator interpolates)
        fb = (torch.as_tensor(calibs_fu, dtype=torch.float32) * torch.as_tensor(calibs_baseline, dtype=torch.float32).abs()).to(fl.device)
        shift = fb[:, None] / torch.tensor(self.DEPTHS, device=fl.device)[None, :] / 4.0
        cost = self.build_cost(fl, fr, shift)                                                                    # [B,8,D,h,w]
        out = self.hg(self.dres0(cost))
        score = self.classif1(out)                                                                               # [B,1,D,h,w]
        score = F.interpolate(score, [self.UP_PLANES, imgL.shape[2], imgL.shape[3]], mode="trilinear", align_corners=False)
        score = torch.squeeze(score, 1)
        pred = F.softmax(score, dim=1)
        depth = self.dispregression(pred, self.depth)                                                            # [B,H,W]
        grid = self._voxel_grid(calibs_Proj, imgL.shape[2:], out.device)
        voxel = self.voxel_conv(F.grid_sample(out, grid, align_corners=False))                                   # [B,8,Z,Y,X]
        b, c, nz, ny, nx = voxel.shape
        bev = voxel.permute(0, 1, 3, 2, 4).reshape(b, c * ny, nz, nx)
        head = self.head(self.bev_conv(bev))
        # eval mode: depth_preds is one [B,H,W] tensor that the scripts iterate over the batch dimension
        return {"depth_preds": depth, "bbox_cls": head[:, 0:1], "bbox_reg": head[:, 1], "bbox_centerness": head[:, 2]}
