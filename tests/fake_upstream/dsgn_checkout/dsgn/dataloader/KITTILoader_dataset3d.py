import types

import numpy as np
import torch
import torch.utils.data
from PIL import Image

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _load(path):
    u8 = np.array(Image.open(path).convert("RGB"))
    x = torch.from_numpy(np.ascontiguousarray(u8.transpose(2, 0, 1))).float() / 255.0
    for c in range(3):
        x[c] = (x[c] - MEAN[c]) / STD[c]
    out = torch.zeros((1, 3, 384, 1248))
    out[0, :, :x.shape[1], :x.shape[2]] = x
    return out, (u8.shape[0], u8.shape[1])


class Calibration:
    def __init__(self, shift):
        self.P = np.array([[721.5377, 0, 609.5593, shift], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
        self.f_u = 721.5377

    def project_image_to_velo(self, pts):
        return np.stack([pts[:, 2], -(pts[:, 0] - 609.5593) * pts[:, 2] / 721.5377, -(pts[:, 1] - 172.854) * pts[:, 2] / 721.5377], 1)


def _calib(shift):
    return Calibration(shift)


class myImageFloder(torch.utils.data.Dataset):
    def __init__(self, left, right, left_disparity, training, split=None, cfg=None):
        self.left, self.right, self.training = left, right, training

    def __len__(self):
        return len(self.left)

    def __getitem__(self, i):
        imgL, size = _load(self.left[i])
        imgR, _ = _load(self.right[i])
        index = int(self.left[i].split("/")[-1].split(".")[0])
        rs = np.random.RandomState(index)
        depth = torch.from_numpy((rs.rand(384, 1248) * 60).astype(np.float32))
        depth[torch.from_numpy(rs.rand(384, 1248) > 0.05)] = 0
        calib, calib_R = _calib(44.85728), _calib(-339.5242)
        if not self.training:
            return imgL, imgR, depth, calib, calib_R, size, index
        target = types.SimpleNamespace(bbox=torch.rand(2, 4) * 300, box3d=torch.rand(2, 7))
        return imgL, imgR, depth, calib, calib_R, index, target, "ious", "labels_map"
