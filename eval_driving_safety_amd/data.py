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
    (``--data_path`` / ``--split_file`` of attack/DSGN/pgd_attack.py:38-45).

    ``workers`` > 0 decodes ahead of the attack on a thread pool (PNG inflate and the float conversion release the
    GIL), ``prefetch`` batches deep, and hands over page-locked tensors so that the driver's host-to-device copy does
    not block - the counterpart of the reference's 12 DataLoader worker processes (attack/DSGN/pgd_attack.py:79,130-133);
    ``workers=0`` decodes synchronously on the attack thread (the reference's ``--debug`` setting)."""

    def __init__(self, data_path, split_file, batch=1, workers=0, prefetch=2, pin=None, as_u8=False, pad_to=(384, 1248)):
        with open(split_file) as f:
            self.ids = [l.strip() for l in f if l.strip()]
        self.root, self.batch, self.workers, self.prefetch = data_path, batch, int(workers), max(1, int(prefetch))
        self.pin = torch.cuda.is_available() if pin is None else pin
        # as_u8: hand over the decoded 8-bit pixels [B,h,w,3]; ToTensor / Normalize / zero padding then run on the device
        # (ops.import_u8, same bits as dsgn_transform) - a quarter of the upload and no float conversion on the decode threads
        self.as_u8, self.pad_to = bool(as_u8), tuple(pad_to)

    def __len__(self):
        return (len(self.ids) + self.batch - 1) // self.batch

    def _eye(self, eye, name):
        from PIL import Image
        with Image.open(os.path.join(self.root, eye, name + ".png")) as im:
            a = np.array(im.convert("RGB"))
        if self.as_u8:      # already in the common buffer shape (the network frame), so that a batch is a plain stack
            buf = np.zeros((self.pad_to[0], self.pad_to[1], 3), np.uint8)
            a = a[:self.pad_to[0], :self.pad_to[1]]         # an oversize frame is clipped to the network frame, and so is its reported size
            buf[:a.shape[0], :a.shape[1]] = a
            return torch.from_numpy(buf), (a.shape[1], a.shape[0])
        u8 = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return dsgn_transform(u8, self.pad_to), (u8.shape[2], u8.shape[1])

    def _assemble(self, names, eyes):
        """eyes: [(tensor, size)] in the order L0, R0, L1, R1 ..."""
        if self.as_u8:      # images of slightly different sizes share one buffer shape, the (w, h) of each travels in ``sizes``
            ls, rs = torch.stack([e[0] for e in eyes[0::2]]), torch.stack([e[0] for e in eyes[1::2]])
            if self.pin:
                ls, rs = ls.pin_memory(), rs.pin_memory()
            return StereoBatch(ls, rs, list(names), [e[1] for e in eyes[0::2]], pad_to=self.pad_to)
        ls, rs = torch.stack([e[0] for e in eyes[0::2]]), torch.stack([e[0] for e in eyes[1::2]])
        if self.pin:
            ls, rs = ls.pin_memory(), rs.pin_memory()
        return StereoBatch(ls, rs, list(names), [e[1] for e in eyes[0::2]])

    def _batch(self, k):
        names = self.ids[k * self.batch:(k + 1) * self.batch]
        return self._assemble(names, [self._eye(eye, n) for n in names for eye in ("image_2", "image_3")])

    def shard(self, rank, world):
        ks = range(rank, len(self), world)
        if self.workers <= 0:
            for k in ks:
                yield self._batch(k)
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="kitti-decode") as pool:
            pending = deque()

            def submit(k):
                names = self.ids[k * self.batch:(k + 1) * self.batch]
                pending.append((names, [pool.submit(self._eye, eye, n) for n in names for eye in ("image_2", "image_3")]))

            it = iter(ks)
            for k in it:
                submit(k)
                if len(pending) > self.prefetch:
                    break
            while pending:
                names, futs = pending.popleft()
                batch = self._assemble(names, [f.result() for f in futs])
                nxt = next(it, None)
                if nxt is not None:
                    submit(nxt)
                yield batch

    def __iter__(self):
        return self.shard(0, 1)
