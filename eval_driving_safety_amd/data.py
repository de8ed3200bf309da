"""Stereo-pair sources for the drivers.  The reference takes its samples from the upstream DSGN /
Stereo R-CNN loaders (not in the reference tree); these are the minimal stand-ins: a seeded synthetic
KITTI-shaped source (no dataset is reachable offline) and a folder reader for KITTI's image_2/image_3."""
import os

import numpy as np
import torch

from .attacks import StereoBatch

KITTI_W, KITTI_H = 1242, 375
DSGN_MEAN = (0.485, 0.456, 0.406)
DSGN_STD = (0.229, 0.224, 0.225)
SRCNN_PIXEL_MEANS = (102.9801, 115.9465, 122.7717)


def _synthetic_u8(gen, h, w):
    low = torch.randint(0, 256, (3, h // 8 + 2, w // 8 + 2), generator=gen, dtype=torch.int32).float()
    up = torch.nn.functional.interpolate(low[None], size=(h, w), mode="bilinear", align_corners=False)[0]
    noise = torch.randint(-8, 9, (3, h, w), generator=gen, dtype=torch.int32).float()
    return (up + noise).clamp_(0, 255).round_()


def dsgn_transform(u8_chw, pad_to=(384, 1248)):
    """u8 RGB [3,h,w] -> normalised float32 zero-padded bottom/right to the network size
    (ToTensor, ImageNet normalise, pad: the upstream DSGN test-time transform, UPSTREAM-UNVERIFIED)."""
    x = u8_chw.float() / 255.0
    for c in range(3):
        x[c] = (x[c] - DSGN_MEAN[c]) / DSGN_STD[c]
    out = torch.zeros((3,) + tuple(pad_to), dtype=torch.float32)
    out[:, :x.shape[1], :x.shape[2]] = x
    return out


class SyntheticStereo:
    """Seeded KITTI-shaped stereo pairs: the right eye is the left one shifted by a constant disparity.
    Every pair has its own seed, so a rank can skip the batches it does not own at no cost (``shard``)."""

    def __init__(self, n_pairs, model_kind="dsgn", batch=1, seed=0, first_index=0):
        self.n, self.kind, self.batch, self.seed, self.first = n_pairs, model_kind, batch, seed, first_index

    def __len__(self):
        return (self.n + self.batch - 1) // self.batch

    def _pair(self, i):
        gen = torch.Generator().manual_seed(self.seed * 1000003 + i)
        if self.kind == "dsgn":
            left = _synthetic_u8(gen, KITTI_H, KITTI_W)
            right = torch.roll(left, shifts=-24, dims=2)
            return dsgn_transform(left), dsgn_transform(right), (KITTI_W, KITTI_H)
        left = _synthetic_u8(gen, 600, 1987)
        right = torch.roll(left, shifts=-38, dims=2)
        m = torch.tensor(SRCNN_PIXEL_MEANS).view(3, 1, 1)
        return left - m, right - m, None

    def _batch(self, k):
        ls, rs, names, sizes = [], [], [], []
        for i in range(k * self.batch, min(self.n, (k + 1) * self.batch)):
            l, r, size = self._pair(i)
            ls.append(l)
            rs.append(r)
            sizes.append(size)
            names.append("%06d" % (self.first + i))
        return StereoBatch(torch.stack(ls), torch.stack(rs), names, sizes if self.kind == "dsgn" else None)

    def shard(self, rank, world):
        """the batches k with k % world == rank, in order"""
        for k in range(rank, len(self), world):
            yield self._batch(k)

    def __iter__(self):
        return self.shard(0, 1)


class KittiFolder:
    """``<data_path>/image_2/NNNNNN.png`` + ``image_3`` for the indices of a split file
    (``--data_path`` / ``--split_file`` of attack/DSGN/pgd_attack.py:38-45)."""

    def __init__(self, data_path, split_file, batch=1):
        with open(split_file) as f:
            self.ids = [l.strip() for l in f if l.strip()]
        self.root, self.batch = data_path, batch

    def __len__(self):
        return (len(self.ids) + self.batch - 1) // self.batch

    def _batch(self, k):
        from PIL import Image
        ls, rs, names, sizes = [], [], [], []
        for name in self.ids[k * self.batch:(k + 1) * self.batch]:
            pair = []
            for eye in ("image_2", "image_3"):
                im = Image.open(os.path.join(self.root, eye, name + ".png")).convert("RGB")
                pair.append(torch.from_numpy(np.ascontiguousarray(np.array(im).transpose(2, 0, 1))))
            sizes.append((pair[0].shape[2], pair[0].shape[1]))
            ls.append(dsgn_transform(pair[0]))
            rs.append(dsgn_transform(pair[1]))
            names.append(name)
        return StereoBatch(torch.stack(ls), torch.stack(rs), names, sizes)

    def shard(self, rank, world):
        for k in range(rank, len(self), world):
            yield self._batch(k)

    def __iter__(self):
        return self.shard(0, 1)
