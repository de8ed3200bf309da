"""Host-side geometry of the round universal patch: size from the ratio, random centre per image,
fake ground truth.  Integer / RNG logic only - no pixel arithmetic happens here (the disc itself is
evaluated analytically inside the paste kernel).  Citations relative to the reference root."""
import random as _random

# column bands of the patch centre as fractions of the image width:
#   'random'                 attack/DSGN/patch_attack.py:240, attack/Stereo-RCNN/patch_attack.py:82
#   'sp_left/straight/right' attack/DSGN/predict_and_save_patch.py:366-373,
#                            attack/Stereo-RCNN/predict_and_save_patch.py:87-94
ATK_MODES = {"random": (0.2, 0.8), "sp_left": (0.2, 0.4), "sp_straight": (0.4, 0.6), "sp_right": (0.6, 0.8)}

DSGN_SHAPE = (384, 1248)     # hard-asserted at attack/DSGN/patch_attack.py:318-320
SRCNN_SHAPE = (600, 1987)    # hard-asserted at attack/Stereo-RCNN/patch_attack.py:170-172
RIGHT_EYE_SHIFT = 40 * 1.6   # attack/DSGN/patch_attack.py:243 - fixed 64 px disparity for both models (quirk Q12)


def init_patch_dims(short_side, patch_ratio):
    """(patch_dim, radius) as init_patch computes them: attack/DSGN/patch_attack.py:213-218
    (short side 384), attack/Stereo-RCNN/patch_attack.py:60-65 (600)."""
    patch_dim = int(short_side * patch_ratio)
    if patch_dim % 2 == 0:
        patch_dim += 1
    return patch_dim, int(patch_dim / 2)


class CenterSampler:
    """generate_round_mask's centre draw (attack/DSGN/patch_attack.py:239-243): two inclusive
    ``randint`` calls, row first, on a Python ``random`` stream.  The reference never seeds that
    stream (quirk Q9); here it is an explicit ``random.Random(seed)`` so runs can be reproduced, and
    ``CenterSampler(rng=random)`` reproduces the reference's use of the module-level stream."""

    def __init__(self, h, w, radius, atk_mode="random", seed=None, rng=None):
        if atk_mode not in ATK_MODES:
            raise Exception("Patch attack mode NOT found.")      # predict_and_save_patch.py:375
        self.h, self.w, self.radius = int(h), int(w), int(radius)
        self.band = ATK_MODES[atk_mode]
        self.rng = rng if rng is not None else _random.Random(seed)

    def draw(self):
        """-> ([cy, cx_left], [cy, cx_right])"""
        cy = self.rng.randint(int(self.h * 0.4), int(self.h - self.radius - 1))
        cx = self.rng.randint(int(self.w * self.band[0]), int(self.w * self.band[1]))
        return [cy, cx], [cy, int(cx - RIGHT_EYE_SHIFT)]


# attack/DSGN/patch_attack.py:341-354 - the one fake car every image is trained towards
DSGN_FAKE_BBOX = (569.33, 180.88, 613.91, 225.02)                   # x1, y1, x2, y2
DSGN_FAKE_BOX3D = (1.65, 1.67, 3.64, -0.78, 1.98, 29.11, -1.60)     # h, w, l, x, y, z, theta


def inject_fake_target_dsgn(bbox, box3d):
    """attack/DSGN/patch_attack.py:336-354 on ``targets[0].bbox`` [n,4] / ``.box3d`` [n,7]
    (torch tensors, modified in place): zero every box, then box 0 <- the fixed fake car."""
    bbox.zero_()
    box3d.zero_()
    if bbox.shape[0] > 0:
        for k, v in enumerate(DSGN_FAKE_BBOX):
            bbox[0, k] = v
        for k, v in enumerate(DSGN_FAKE_BOX3D):
            box3d[0, k] = v
    return bbox, box3d


def inject_fake_target_srcnn(gt_left, gt_right, gt_merge, center_l, center_r, radius):
    """attack/Stereo-RCNN/patch_attack.py:187-207: gt boxes [1,K,5] zeroed, row 0 <- the patch's
    bounding square per eye (x1,y1,x2,y2), merge = left; class column stays 0; returns num_boxes = 1."""
    for t in (gt_left, gt_right, gt_merge):
        t.zero_()
    for t, c in ((gt_left, center_l), (gt_right, center_r), (gt_merge, center_l)):
        t[0, 0, 0] = c[1] - radius
        t[0, 0, 1] = c[0] - radius
        t[0, 0, 2] = c[1] + radius
        t[0, 0, 3] = c[0] + radius
    return 1
