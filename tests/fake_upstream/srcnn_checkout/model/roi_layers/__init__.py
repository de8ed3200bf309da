"""Stands where the upstream checkout's COMPILED ``model.roi_layers`` (a CUDA extension) stands: ROIAlign and nms with the upstream
names and signatures, written here in plain torch (slow, differentiable, runs on any device) so that the stand-in detector works
without a GPU.  On an MI355X ``eval_driving_safety_amd.upstream_shims.install()`` replaces this package by the libadvengine one -
tests/test_upstream_binding.py checks that it does, and that both give the same numbers."""
import math

import torch
import torch.nn as nn

IS_TORCH_STAND_IN = True


def _bilinear(feat, y, x):
    """feat [C,H,W]; y [ny], x [nx] sample coordinates -> [C,ny,nx] with the legacy RoIAlign border rules"""
    h, w = feat.shape[1:]
    oky = ((y >= -1.0) & (y <= h)).to(feat.dtype)
    okx = ((x >= -1.0) & (x <= w)).to(feat.dtype)
    y, x = y.clamp(min=0), x.clamp(min=0)
    y0, x0 = y.floor().long(), x.floor().long()
    ty, tx = y0 >= h - 1, x0 >= w - 1
    y0, x0 = torch.where(ty, torch.full_like(y0, h - 1), y0), torch.where(tx, torch.full_like(x0, w - 1), x0)
    y1, x1 = torch.where(ty, y0, y0 + 1), torch.where(tx, x0, x0 + 1)
    ly = torch.where(ty, torch.zeros_like(y), y - y0.to(y.dtype))
    lx = torch.where(tx, torch.zeros_like(x), x - x0.to(x.dtype))
    hy, hx = 1 - ly, 1 - lx
    f = lambda yy, xx: feat[:, yy][:, :, xx]                                   # noqa: E731
    out = (hy[:, None] * hx[None, :]) * f(y0, x0) + (hy[:, None] * lx[None, :]) * f(y0, x1) + \
          (ly[:, None] * hx[None, :]) * f(y1, x0) + (ly[:, None] * lx[None, :]) * f(y1, x1)
    return out * (oky[:, None] * okx[None, :])


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio=0):
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    scale = float(spatial_scale)
    outs = []
    for r in rois:
        b = int(r[0])
        x1, y1, x2, y2 = [float(v) * scale for v in r[1:5]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bh, bw = rh / ph, rw / pw
        gy = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gx = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        ys = y1 + (torch.arange(ph * gy, device=input.device, dtype=input.dtype) // gy) * bh + \
            ((torch.arange(ph * gy, device=input.device, dtype=input.dtype) % gy) + 0.5) * bh / gy
        xs = x1 + (torch.arange(pw * gx, device=input.device, dtype=input.dtype) // gx) * bw + \
            ((torch.arange(pw * gx, device=input.device, dtype=input.dtype) % gx) + 0.5) * bw / gx
        v = _bilinear(input[b], ys, xs)                                        # [C, ph*gy, pw*gx]
        outs.append(v.view(v.shape[0], ph, gy, pw, gx).mean(dim=(2, 4)))
    return torch.stack(outs) if outs else input.new_zeros((0, input.shape[1], ph, pw))


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois, spatial_scale):
        return roi_align(input, rois, self.output_size, spatial_scale, self.sampling_ratio)


def nms(dets, scores, thresh):
    """greedy suppression, IoU with the legacy +1 areas, highest score first -> kept indices (int64)"""
    order = torch.sort(scores.reshape(-1), descending=True, stable=True)[1].tolist()
    b = dets[:, :4].detach().cpu().double()
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    keep = []
    for i in order:
        ok = True
        for j in keep:
            iw = min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]) + 1
            ih = min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]) + 1
            if iw > 0 and ih > 0 and float(iw * ih / (area[i] + area[j] - iw * ih)) > thresh:
                ok = False
                break
        if ok:
            keep.append(i)
    return torch.tensor(keep, dtype=torch.int64, device=dets.device)
