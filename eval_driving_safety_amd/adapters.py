"""Model adapters: what the attack drivers call for "detector forward + loss + backward".

The detectors are not part of the reference tree (they are the upstream DSGN / Stereo R-CNN
repositories the user clones), so they are not part of this package either.  An adapter wraps whatever
detector the caller has and returns d loss / d image for the stacked stereo batch:

    loss, grad = adapter.loss_and_grad(x, extra)      # x, grad: [2B,3,H,W]; left eyes first

``DsgnAdapter`` / ``StereoRcnnAdapter`` restate the objective lines of the scripts around an upstream
model object handed in by the caller; ``ToyStereoAdapter`` is a small fixed-seed differentiable stereo
matcher used by the tests, smoke() and the CLI's ``--model toy`` plumbing mode.
"""
import torch
import torch.nn.functional as F

from .determinism import deterministic


def split_eyes(x):
    b = x.shape[0] // 2
    return x[:b], x[b:]


class _LeafGrad:
    """make the stacked buffer a leaf for one forward/backward, hand back its gradient"""

    def __init__(self, x):
        self.x = x

    def __enter__(self):
        self.x.grad = None
        self.x.requires_grad_(True)
        return self.x

    def __exit__(self, *a):
        self.x.requires_grad_(False)
        self.x.grad = None

    def take(self):
        g = self.x.grad
        if g is None:
            raise RuntimeError("the detector loss does not depend on the images")
        return g.detach()


def _freeze(model):
    """The attacks differentiate w.r.t. the images only (attack/DSGN/pgd_attack.py:300-336 read ``imgL.grad`` / ``imgR.grad``; the
    detector stays in eval mode, its weights are constants), but the reference leaves ``requires_grad`` on, so every ``backward``
    also computes every weight gradient and throws it away - a third of the detector step for the Stereo R-CNN-shaped network
    (profiles/r02_end_to_end_srcnn_shaped.json).  Switching it off changes no image gradient (``freeze=False`` keeps the
    reference's behaviour)."""
    params = getattr(model, "parameters", None)
    if callable(params):
        for p in params():
            p.requires_grad_(False)


class ToyStereoAdapter:
    """Fixed-seed two-layer siamese feature net + a 4-plane correlation 'cost volume' + smooth-L1 to a
    synthetic target.  NOT a detector - a deterministic differentiable stand-in with the same call shape,
    so that loop-level behaviour can be tested without DSGN / Stereo R-CNN."""

    graph_safe = True

    def __init__(self, device, seed=0, channels=8, planes=(0, 4, 8, 16)):
        gen = torch.Generator().manual_seed(seed)
        self.w1 = (torch.randn(channels, 3, 3, 3, generator=gen) * 0.2).to(device)
        self.w2 = (torch.randn(channels, channels, 3, 3, generator=gen) * 0.1).to(device)
        self.planes = planes
        self.target = 0.25

    def features(self, img):
        f = F.relu(F.conv2d(img, self.w1, stride=2, padding=1))
        return F.conv2d(f, self.w2, stride=2, padding=1)

    def loss(self, imgL, imgR):
        fl, fr = self.features(imgL), self.features(imgR)
        costs = []
        for d in self.planes:                       # left feature against the right feature shifted by d
            shifted = fr if d == 0 else F.pad(fr, (d, 0))[..., :fr.shape[-1]]
            costs.append((fl * shifted).mean(dim=1))
        cost = torch.stack(costs, dim=1)
        prob = torch.softmax(cost, dim=1)
        depth = sum(p * prob[:, i] for i, p in enumerate(self.planes))
        return F.smooth_l1_loss(depth / max(self.planes), torch.full_like(depth, self.target))

    @deterministic
    def loss_and_grad(self, x, extra=None):
        h = _LeafGrad(x)
        with h as leaf:
            imgL, imgR = split_eyes(leaf)
            loss = self.loss(imgL, imgR)
            loss.backward()
            return loss.detach(), h.take()

    def inject_fake_target(self, extra, centers_l, centers_r, radius):
        pass


class PsvStereoAdapter:
    """A DSGN-SHAPED depth branch with seeded random weights, for end-to-end timing of the attack loop and as
    the autograd consumer of the K7 and convolution kernels: siamese 2D features at 1/4 resolution (32 channels) ->
    plane-sweep concatenation volume [B,64,48,96,312] (HIP: ops.PsvBuildLerp, fractional per-plane disparities) ->
    3D convolutions on the float32 matrix cores -> softmax over the 48 depth planes -> expected depth -> smooth-L1
    against a sparse depth map, i.e. the ``disp_loss`` term of attack/DSGN/pgd_attack.py:310-319.
      hourglass=False  three convolutions 64->32->32->1 (round 1's stack)
      hourglass=True   a 3D hourglass as plane-sweep detectors use it: 64->32, 32->32, then 32->64 stride 2, 64->64,
                       64->64 stride 2, 64->64, transposed 64->64 (+ skip), transposed 64->32 (+ skip), 32->1;
                       every layer but the last carries a bias (a folded batch-norm) and a fused ReLU
      dsgn_head=True   the rest of the DSGN graph shape on libadvengine's kernels: depth = fused trilinear
                       up-sampling + softmax over 192 planes + expectation (ops.DepthRegress); plane-sweep features weighted by
                       the plane probabilities are resampled into a 3D geometric volume on a 0.2 m world grid
                       (ops.GridSample3d, [B,32,192,20,304]), one 3D convolution, height folded into channels -> bird's-eye-view
                       2D convolutions -> per-anchor classification / box / centerness maps; the objective becomes
                       depth smooth-L1 + (sigmoid focal + smooth-L1 + BCE) as attack/DSGN/pgd_attack.py:310-336 adds them.
    It is NOT DSGN (random weights, simplified heads, no trained parameters): detection parity is unpinned by construction; it
    exists so that "20-step PGD through a plane-sweep detector" can be measured end to end on this hardware."""

    graph_safe = True       # no host read-back, no data-dependent shape inside loss_and_grad: attacks.PgdAttack(graph=True) may capture it

    def __init__(self, device, seed=0, channels=32, planes=48, min_depth=2.0, depth_step=0.8, fu=721.5377,
                 baseline=0.54, downsample=4, mid=32, mfma_conv=True, interp=True, hourglass=False, dsgn_head=False,
                 cu=609.5593, cv=172.854, image_hw=(384, 1248)):
        from . import ops
        self.ops = ops
        gen = torch.Generator().manual_seed(seed)

        def w(*shape):
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            return (torch.randn(*shape, generator=gen) * (2.0 / fan_in) ** 0.5).to(device)

        self.f1, self.f2, self.f3 = w(16, 3, 3, 3), w(channels, 16, 3, 3), w(channels, channels, 3, 3)
        self.c1, self.c2, self.c3 = w(mid, 2 * channels, 3, 3, 3), w(mid, mid, 3, 3, 3), w(1, mid, 3, 3, 3)
        self.depth = (min_depth + depth_step * torch.arange(planes, dtype=torch.float32)).to(device)
        self.fu, self.baseline, self.downsample = fu, baseline, downsample
        self.device = device
        # interp=True: the fractional per-plane disparities fu*b/depth/4 go to the interpolating cost volume
        # (ops.PsvBuildLerp); interp=False rounds them to integers (ops.PsvBuild, round 1's behaviour)
        self.interp = interp
        # the 3x3x3 convolutions run on libadvengine's float32-MFMA kernels (weights re-laid-out once, for the
        # forward and for the adjoint); mfma_conv=False routes them through torch / MIOpen instead
        self.mfma_conv = mfma_conv and (2 * channels) % 4 == 0 and mid % 4 == 0
        self.hourglass, self.mid = hourglass, mid
        if self.mfma_conv:
            self.p1, self.p1t = ops.conv3d_k3_prep(self.c1), ops.conv3d_k3_prep(self.c1, transpose=True)
            self.p2, self.p2t = ops.conv3d_k3_prep(self.c2), ops.conv3d_k3_prep(self.c2, transpose=True)
            # 32 -> 1: the narrow vector-ALU kernels, forward (Cout = 1) and adjoint (Cin = 1)
            self.p3, self.p3t = ops.conv3d_k3_prep(self.c3), ops.conv3d_k3_prep(self.c3, transpose=True)
        if hourglass:
            m2 = 2 * mid
            self.hg = {"d1": w(m2, mid, 3, 3, 3), "m1": w(m2, m2, 3, 3, 3), "d2": w(m2, m2, 3, 3, 3), "m2": w(m2, m2, 3, 3, 3),
                       "u1": w(m2, m2, 3, 3, 3), "u2": w(m2, mid, 3, 3, 3)}                # u*: ConvTranspose layout [in, out, 3,3,3]
            self.hb = {k: (torch.randn(v.shape[1] if k.startswith("u") else v.shape[0], generator=gen) * 0.05).to(device)
                       for k, v in self.hg.items()}
            self.b1 = (torch.randn(mid, generator=gen) * 0.05).to(device)
            self.b2 = (torch.randn(mid, generator=gen) * 0.05).to(device)
            if self.mfma_conv:
                P, S2 = ops.conv3d_k3_prep, ops.conv3d_k3_s2_prep
                self.hp = {"d1": (S2(self.hg["d1"]), ops.conv_transpose3d_k3_s2_prep(self.hg["d1"])),
                           "d2": (S2(self.hg["d2"]), ops.conv_transpose3d_k3_s2_prep(self.hg["d2"])),
                           "m1": (P(self.hg["m1"]), P(self.hg["m1"], transpose=True)),
                           "m2": (P(self.hg["m2"]), P(self.hg["m2"], transpose=True)),
                           "u1": (ops.conv_transpose3d_k3_s2_prep(self.hg["u1"]), S2(self.hg["u1"])),
                           "u2": (ops.conv_transpose3d_k3_s2_prep(self.hg["u2"]), S2(self.hg["u2"]))}

        self.dsgn_head = dsgn_head
        if dsgn_head:
            self._init_dsgn_head(w, gen, min_depth, depth_step, cu, cv, image_hw)

    # -- the DSGN-shaped detection branch ------------------------------------------------------------------------
    VOXEL, X_RANGE, Y_RANGE, ANCHORS = 0.2, (-30.4, 30.4), (-1.0, 3.0), 2

    def _init_dsgn_head(self, w, gen, min_depth, depth_step, cu, cv, image_hw):
        ops, dev, mid = self.ops, self.device, self.mid
        planes = self.depth.shape[0]
        up = self.downsample * planes
        # depth of up-sampled plane k: the source coordinate of torch's linear interpolation (align_corners=False), unclamped
        src = (torch.arange(up, dtype=torch.float32) + 0.5) / self.downsample - 0.5
        self.depth_up = (min_depth + depth_step * src).to(dev).contiguous()
        self.up_size = (up, image_hw[0], image_hw[1])
        # world grid of the 3D geometric volume -> where each voxel centre lands in the plane-sweep volume (x: feature column,
        # y: feature row, z: depth plane), normalised for grid_sample(align_corners=True)
        v = self.VOXEL
        zs = min_depth + v * (torch.arange(int(round(depth_step * planes / v)), dtype=torch.float32) + 0.5)
        ys = self.Y_RANGE[0] + v * (torch.arange(int(round((self.Y_RANGE[1] - self.Y_RANGE[0]) / v)), dtype=torch.float32) + 0.5)
        xs = self.X_RANGE[0] + v * (torch.arange(int(round((self.X_RANGE[1] - self.X_RANGE[0]) / v)), dtype=torch.float32) + 0.5)
        Z, Y, X = torch.meshgrid(zs, ys, xs, indexing="ij")
        fh, fw = image_hw[0] // self.downsample, image_hw[1] // self.downsample
        col = (self.fu * X / Z + cu) / self.downsample
        row = (self.fu * Y / Z + cv) / self.downsample
        pl = (Z - min_depth) / depth_step
        grid = torch.stack([col / (fw - 1) * 2 - 1, row / (fh - 1) * 2 - 1, pl / (planes - 1) * 2 - 1], dim=-1)
        self.gv_grid = grid[None].to(dev).contiguous()                   # [1,Zg,Yg,Xg,3]: one calibration for the whole batch
        self.gv_dims = (planes, fh, fw)
        self._gv_plans = {}
        ypool = 4
        self.ypool = ypool
        bev_in = mid * (len(ys) // ypool)
        self.g3 = w(mid, mid, 3, 3, 3)
        self.gb3 = (torch.randn(mid, generator=gen) * 0.05).to(dev)
        if self.mfma_conv:
            self.gp3, self.gp3t = ops.conv3d_k3_prep(self.g3), ops.conv3d_k3_prep(self.g3, transpose=True)
        a = self.ANCHORS
        self.bev1, self.bev2 = w(64, bev_in, 3, 3), w(64, 64, 3, 3)
        self.head_cls, self.head_reg, self.head_ctr = w(a, 64, 3, 3), w(a * 7, 64, 3, 3), w(a, 64, 3, 3)
        self.cls_bias = float(-torch.log(torch.tensor(99.0)))            # the usual focal-loss prior: p = 0.01

    def _gv(self, b):
        """grid [b,Zg,Yg,Xg,3] and its backward plan (built once per batch size: the calibration is fixed)"""
        if b not in self._gv_plans:
            grid = self.gv_grid.expand(b, -1, -1, -1, -1).contiguous()
            self._gv_plans[b] = (grid, self.ops.GridSamplePlan(grid, self.gv_dims, align_corners=True))
        return self._gv_plans[b]

    def detection_maps(self, feat_vol, cost):
        """plane-sweep feature volume [B,32,D,h,w] + plane scores [B,D,h,w] -> (cls [B,A,Z,X], reg [B,7A,Z,X], ctr [B,A,Z,X])"""
        ops = self.ops
        prob = torch.softmax(cost, dim=1)
        grid, plan = self._gv(feat_vol.shape[0])
        gv = ops.GridSample3d.apply((feat_vol * prob[:, None]).contiguous(), grid, plan)          # [B,32,Zg,Yg,Xg]
        if self.mfma_conv:
            gv = ops.Conv3dK3.apply(gv, self.gp3, self.gp3t, self.mid, None, self.gb3, True)
        else:
            gv = F.relu(F.conv3d(gv, self.g3, self.gb3, padding=1))
        b, c, zg, yg, xg = gv.shape
        bev = F.avg_pool3d(gv, (1, self.ypool, 1)).permute(0, 1, 3, 2, 4).reshape(b, c * (yg // self.ypool), zg, xg)
        bev = F.relu(F.conv2d(bev, self.bev1, padding=1))
        bev = F.relu(F.conv2d(bev, self.bev2, padding=1))
        return (F.conv2d(bev, self.head_cls, padding=1) + self.cls_bias, F.conv2d(bev, self.head_reg, padding=1),
                F.conv2d(bev, self.head_ctr, padding=1))

    def detection_targets(self, boxes, zg, xg):
        """boxes [[x, z, l, w, ry], ...] per image (metres) -> (cls targets int32 [B*Z*X*A], reg targets [B,7A,Z,X], ctr [B,A,Z,X])"""
        v, a = self.VOXEL, self.ANCHORS
        b = len(boxes)
        cls = torch.zeros((b, zg, xg, a), dtype=torch.int32)
        reg = torch.zeros((b, a * 7, zg, xg))
        ctr = torch.zeros((b, a, zg, xg))
        zc = float(self.depth[0]) + v * (torch.arange(zg, dtype=torch.float32) + 0.5)
        xc = self.X_RANGE[0] + v * (torch.arange(xg, dtype=torch.float32) + 0.5)
        for i, bl in enumerate(boxes):
            for (x, z, l, wd, ry) in bl:
                inz, inx = (zc - z).abs() <= wd / 2, (xc - x).abs() <= l / 2
                m = inz[:, None] & inx[None, :]
                if not bool(m.any()):
                    continue
                k = 0 if abs(ry) < 0.785 else 1                                   # anchor = nearest of the two orientations
                cls[i, :, :, k][m] = 1
                dz, dx = (z - zc)[:, None].expand(zg, xg), (x - xc)[None, :].expand(zg, xg)
                vals = [dx, dz, torch.full_like(dx, l), torch.full_like(dx, wd), torch.full_like(dx, 1.5), torch.full_like(dx, 1.0),
                        torch.full_like(dx, ry)]
                for j, val in enumerate(vals):
                    reg[i, k * 7 + j][m] = val[m]
                ctr[i, k][m] = (1 - (dz.abs() / (wd / 2)).clamp(max=1))[m] * (1 - (dx.abs() / (l / 2)).clamp(max=1))[m]
        dev = self.device
        # The labels do not change over the N iterations of an attack: everything the loss needs from them is computed HERE, once -
        # the number of positives as a Python int, the positives as flat index lists, the targets already gathered - so that an
        # iteration reads nothing back from the device (no .item(), no bool(tensor), no boolean-mask indexing = nonzero + D2H).
        pos = (cls > 0).permute(0, 3, 1, 2).contiguous()                                      # [B,A,Z,X]
        posr = pos.repeat_interleave(7, dim=1)
        pos_idx, posr_idx = torch.nonzero(pos.reshape(-1)).view(-1), torch.nonzero(posr.reshape(-1)).view(-1)
        return {"cls": cls.reshape(-1).contiguous().to(dev), "npos": max(1, int(pos_idx.numel())), "any": bool(pos_idx.numel() > 0),
                "pos_idx": pos_idx.to(dev), "posr_idx": posr_idx.to(dev),
                "reg_pos": reg.reshape(-1)[posr_idx].to(dev), "ctr_pos": ctr.reshape(-1)[pos_idx].to(dev)}

    def detection_loss(self, maps, targets):
        """sigmoid focal / N_pos + smooth-L1 on the positives + BCE centerness on the positives (the three RPN3DLoss terms)"""
        cls, reg, ctr = maps
        t = targets
        logits = cls.permute(0, 2, 3, 1).reshape(-1, 1).contiguous()                       # [B*Z*X*A, 1]: one class ("Car")
        npos = t["npos"]
        l_cls = self.ops.SigmoidFocalLoss.apply(logits, t["cls"], 2.0, 0.25) / npos
        if t["any"]:                                                                       # host-side facts of the label set: no sync
            l_reg = F.smooth_l1_loss(reg.reshape(-1).index_select(0, t["posr_idx"]), t["reg_pos"], reduction="sum") / npos
            l_ctr = F.binary_cross_entropy_with_logits(ctr.reshape(-1).index_select(0, t["pos_idx"]), t["ctr_pos"], reduction="sum") / npos
        else:
            l_reg = l_ctr = cls.sum() * 0
        return l_cls + l_reg + l_ctr

    def shifts(self, b):
        disp = self.fu * self.baseline / self.depth / self.downsample       # feature-pixel disparity per plane
        if self.interp:
            return disp.to(torch.float32).repeat(b, 1).contiguous()
        return disp.round().to(torch.int32).repeat(b, 1).contiguous()

    def features(self, img):
        f = F.relu(F.conv2d(img, self.f1, stride=2, padding=1))
        f = F.relu(F.conv2d(f, self.f2, stride=2, padding=1))
        return F.conv2d(f, self.f3, padding=1)

    def _volume_net(self, cost, with_features=False):
        """cost volume [B,64,D,h,w] -> per-plane scores [B,D,h,w] (and the last feature volume [B,32,D,h,w])"""
        score, feat = self._volume_net_impl(cost)
        return (score, feat) if with_features else score

    def _volume_net_impl(self, cost):
        ops = self.ops
        if not self.hourglass:
            if self.mfma_conv:
                v = ops.Conv3dK3.apply(cost, self.p1, self.p1t, self.mid, None, None, True)
                v = ops.Conv3dK3.apply(v, self.p2, self.p2t, self.mid, None, None, True)
                return ops.Conv3dK3.apply(v, self.p3, self.p3t, 1).squeeze(1), v
            v = F.relu(F.conv3d(cost, self.c1, padding=1))
            v = F.relu(F.conv3d(v, self.c2, padding=1))
            return F.conv3d(v, self.c3, padding=1).squeeze(1), v
        g, hb, m2 = self.hg, self.hb, 2 * self.mid
        if self.mfma_conv:
            hp = self.hp
            s0 = ops.Conv3dK3.apply(cost, self.p1, self.p1t, self.mid, None, self.b1, True)
            s0 = ops.Conv3dK3.apply(s0, self.p2, self.p2t, self.mid, None, self.b2, True)
            s1 = ops.Conv3dK3S2.apply(s0, hp["d1"][0], hp["d1"][1], m2, hb["d1"], True)
            s1 = ops.Conv3dK3.apply(s1, hp["m1"][0], hp["m1"][1], m2, None, hb["m1"], True)
            s2 = ops.Conv3dK3S2.apply(s1, hp["d2"][0], hp["d2"][1], m2, hb["d2"], True)
            s2 = ops.Conv3dK3.apply(s2, hp["m2"][0], hp["m2"][1], m2, None, hb["m2"], True)
            u1 = ops.ConvTranspose3dK3S2.apply(s2, hp["u1"][0], hp["u1"][1], m2, hb["u1"], True, s1)       # relu(up(s2) + bias + s1): one launch
            u2 = ops.ConvTranspose3dK3S2.apply(u1, hp["u2"][0], hp["u2"][1], self.mid, hb["u2"], True, s0)
            return ops.Conv3dK3.apply(u2, self.p3, self.p3t, 1).squeeze(1), u2
        s0 = F.relu(F.conv3d(cost, self.c1, self.b1, padding=1))
        s0 = F.relu(F.conv3d(s0, self.c2, self.b2, padding=1))
        s1 = F.relu(F.conv3d(s0, g["d1"], hb["d1"], stride=2, padding=1))
        s1 = F.relu(F.conv3d(s1, g["m1"], hb["m1"], padding=1))
        s2 = F.relu(F.conv3d(s1, g["d2"], hb["d2"], stride=2, padding=1))
        s2 = F.relu(F.conv3d(s2, g["m2"], hb["m2"], padding=1))
        u1 = F.relu(F.conv_transpose3d(s2, g["u1"], hb["u1"], stride=2, padding=1, output_padding=1) + s1)
        u2 = F.relu(F.conv_transpose3d(u1, g["u2"], hb["u2"], stride=2, padding=1, output_padding=1) + s0)
        return F.conv3d(u2, self.c3, padding=1).squeeze(1), u2

    def plane_prob(self, imgL, imgR):
        fl, fr = self.features(imgL), self.features(imgR)
        build = self.ops.PsvBuildLerp if self.interp else self.ops.PsvBuild
        cost = build.apply(fl.contiguous(), fr.contiguous(), self.shifts(imgL.shape[0]))
        return torch.softmax(self._volume_net(cost), dim=1)                  # [B,D,h,w]

    def depth_pred(self, imgL, imgR):
        prob = self.plane_prob(imgL, imgR)
        depth = (prob * self.depth.view(1, -1, 1, 1)).sum(dim=1, keepdim=True)
        return F.interpolate(depth, scale_factor=self.downsample, mode="bilinear", align_corners=False).squeeze(1)

    # -- a bird's-eye-view box head on the plane probabilities, for the detect-under-attack drivers --------------------
    def _bev_weights(self):
        if not hasattr(self, "_bev"):
            gen = torch.Generator().manual_seed(4321)
            w1 = (torch.randn(16, 1, 3, 3, generator=gen) * 0.6).to(self.device)
            w2 = (torch.randn(9, 16, 3, 3, generator=gen) * 0.25).to(self.device)
            self._bev = (w1, w2)
        return self._bev

    @deterministic
    def detect(self, x, extra=None, topk=24, nms_thresh=0.25, cu=609.5593, cv=172.854):
        """-> per stereo pair a list of (cls, bbox[4], score, center[3], (h, w, l, ry)) for ``DetectUnderAttack`` /
        ``pixelio.write_kitti_labels``.  Plane probabilities -> occupancy over (depth plane, image column) -> a small 2D head
        (score + box regression per cell) -> the ``topk`` cells by score -> greedy NMS on their bird's-eye-view footprints
        (``ops.nms``, deterministic) -> 3D boxes and their projected 2D boxes.  Random weights: the boxes mean nothing, the
        PIPELINE (attacked images -> detector -> label files -> scenario conversion) is what this exercises."""
        b = x.shape[0] // 2
        if self.dsgn_head:
            return self._detect_bev(x, topk, nms_thresh, cu, cv)
        with torch.no_grad():
            prob = self.plane_prob(x[:b], x[b:])
            w1, w2 = self._bev_weights()
            bev = prob.amax(dim=2)                                           # [B,D,w]
            out = F.conv2d(F.relu(F.conv2d(bev[:, None] * 8.0, w1, padding=1)), w2, padding=1)     # [B,9,D,w]
            results = []
            for i in range(b):
                o = out[i]
                score = torch.sigmoid(o[0]).reshape(-1)
                top = torch.argsort(score, descending=True)[:topk]
                di, ui = torch.div(top, o.shape[2], rounding_mode="floor"), top % o.shape[2]
                reg = o[1:, di, ui]                                          # [8,K]
                z = self.depth[di] + 0.4 * torch.tanh(reg[1])
                u = (ui.float() + 0.5 + torch.tanh(reg[0])) * self.downsample
                xc = (u - cu) * z / self.fu
                hh, ww, ll = 1.5 + 0.2 * torch.tanh(reg[2]), 1.6 + 0.2 * torch.tanh(reg[3]), 3.9 + 0.5 * torch.tanh(reg[4])
                ry = torch.atan2(reg[5], reg[6] + 1e-6)
                yc = 1.0 + 0.3 * torch.tanh(reg[7])
                foot = torch.stack([xc - ll / 2, z - ww / 2, xc + ll / 2, z + ww / 2], 1) * 10.0   # decimetres: "+1" IoU areas stay small
                keep = self.ops.nms(foot.contiguous(), score[top].contiguous(), nms_thresh)
                dets = []
                for k in keep.tolist():
                    cx_, cy_, cz_ = float(xc[k]), float(yc[k]), float(z[k])
                    h_, w_, l_ = float(hh[k]), float(ww[k]), float(ll[k])
                    us = [(cx_ + sx * l_ / 2) * self.fu / cz_ + cu for sx in (-1, 1)]
                    vs = [(cy_ + sy * h_ / 2) * self.fu / cz_ + cv for sy in (-1, 1)]
                    bbox = [min(us), min(vs), max(us), max(vs)]
                    dets.append((2, bbox, float(score[top][k]), [cx_, cy_, cz_], (h_, w_, l_, float(ry[k]))))
                results.append(dets)
        return results

    def _detect_bev(self, x, topk, nms_thresh, cu, cv):
        """the DSGN-shaped head's detections: score = sigmoid(cls) * sigmoid(centerness) per (anchor, BEV cell) -> top-k ->
        box decode (cell centre + regressed offsets / size / height / angle) -> deterministic NMS on the BEV footprints"""
        b = x.shape[0] // 2
        v = self.VOXEL
        with torch.no_grad():
            _, (cls, reg, ctr) = self.forward_all(x[:b], x[b:])
            a, zg, xg = cls.shape[1:]
            score_map = torch.sigmoid(cls) * torch.sigmoid(ctr)
            results = []
            for i in range(b):
                score = score_map[i].reshape(-1)
                top = torch.argsort(score, descending=True, stable=True)[:topk]
                k = torch.div(top, zg * xg, rounding_mode="floor")
                zi = torch.div(top % (zg * xg), xg, rounding_mode="floor")
                xi = top % xg
                r = reg[i].view(a, 7, zg, xg)[k, :, zi, xi]                                   # [K,7]: dx dz l w h y ry
                xc = self.X_RANGE[0] + v * (xi.float() + 0.5) + r[:, 0]
                z = float(self.depth[0]) + v * (zi.float() + 0.5) + r[:, 1]
                z = z.clamp(min=1.0)
                ll, ww, hh = 3.9 + torch.tanh(r[:, 2]), 1.6 + 0.3 * torch.tanh(r[:, 3]), 1.5 + 0.3 * torch.tanh(r[:, 4])
                yc = 1.0 + 0.5 * torch.tanh(r[:, 5])
                ry = torch.where(k == 0, torch.zeros_like(z), torch.full_like(z, 1.5708)) + 0.785 * torch.tanh(r[:, 6])
                foot = torch.stack([xc - ll / 2, z - ww / 2, xc + ll / 2, z + ww / 2], 1) * 10.0      # decimetres, as in detect()
                sc = score[top].contiguous()
                keep = self.ops.nms(foot.contiguous(), sc, nms_thresh)
                dets = []
                for j in keep.tolist():
                    cx_, cy_, cz_ = float(xc[j]), float(yc[j]), float(z[j])
                    h_, w_, l_ = float(hh[j]), float(ww[j]), float(ll[j])
                    us = [(cx_ + sx * l_ / 2) * self.fu / cz_ + cu for sx in (-1, 1)]
                    vs = [(cy_ + sy * h_ / 2) * self.fu / cz_ + cv for sy in (-1, 1)]
                    dets.append((2, [min(us), min(vs), max(us), max(vs)], float(sc[j]), [cx_, cy_, cz_], (h_, w_, l_, float(ry[j]))))
                results.append(dets)
        return results

    def synthetic_extra(self, batch, seed=1):
        """a sparse synthetic depth map per pair (5 % of the pixels, 2 .. 40.4 m) for runs without a dataset; with the
        detection head also 1-4 car boxes per pair [x, z, l, w, ry] in metres"""
        import types
        b = len(batch)
        hh, ww = batch.pad_to if getattr(batch, "pad_to", None) else (batch.imgL.shape[2], batch.imgL.shape[3])
        gen = torch.Generator().manual_seed(seed)
        gt = torch.rand((b, hh, ww), generator=gen) * 38.4 + 2.0
        gt = torch.where(torch.rand((b, hh, ww), generator=gen) < 0.05, gt, torch.zeros(()))
        extra = types.SimpleNamespace(disp_true=gt.to(self.device))
        if self.dsgn_head:
            boxes = []
            for _ in range(b):
                n = int(torch.randint(1, 5, (1,), generator=gen))
                r = torch.rand((n, 5), generator=gen)
                boxes.append([(float(-20 + 40 * q[0]), float(6 + 30 * q[1]), 3.9, 1.6, float(-1.57 + 3.14 * q[4])) for q in r])
            extra.boxes = boxes
        return extra

    def forward_all(self, imgL, imgR):
        """-> (depth [B,H,W], detection maps or None): the whole DSGN-shaped graph"""
        fl, fr = self.features(imgL), self.features(imgR)
        build = self.ops.PsvBuildLerp if self.interp else self.ops.PsvBuild
        cost = build.apply(fl.contiguous(), fr.contiguous(), self.shifts(imgL.shape[0]))
        score, feat = self._volume_net(cost, with_features=True)
        depth = self.ops.DepthRegress.apply(score.contiguous(), self.depth_up, self.up_size, False)
        return depth, self.detection_maps(feat, score)

    def _depth_targets(self, gt):
        """the valid-depth mask of pgd_attack.py:269 as a flat index list + the depths it selects, computed once per ground-truth
        tensor (it does not change over the iterations of an attack): an iteration then gathers by index, with no nonzero / D2H"""
        # keyed on the tensor OBJECT (kept alive here) and its version counter: an address is not an identity - the caching allocator hands
        # the next batch's ground truth the address of the one just freed, with the same shape and version 0
        held = getattr(self, "_gt_key", None)
        if held is None or held[0] is not gt or held[1] != gt._version:
            lo, hi = float(self.depth[0]), float(self.depth[-1]) + 0.8
            mask = (gt > lo) & (gt <= hi)
            idx = torch.nonzero(mask.reshape(-1)).view(-1)
            self._gt_cache, self._gt_key = (idx, gt.reshape(-1).index_select(0, idx)), (gt, gt._version)
        return self._gt_cache

    @deterministic
    def loss_and_grad(self, x, extra):
        """extra.disp_true [B,H,W] sparse metric depth (0 = no measurement); mask as pgd_attack.py:269"""
        h = _LeafGrad(x)
        with h as leaf:
            imgL, imgR = split_eyes(leaf)
            gt_idx, gt_sel = self._depth_targets(extra.disp_true)
            if self.dsgn_head:      # pgd_attack.py:310-336: depth term + the detection head's three terms
                pred, maps = self.forward_all(imgL, imgR)
                key = (tuple(maps[0].shape), tuple(tuple(tuple(float(v) for v in bx) for bx in img) for img in extra.boxes))
                if getattr(self, "_tgt_key", None) != key:     # targets depend on the labels only: once per label set, not per step
                    self._tgt, self._tgt_key = self.detection_targets(extra.boxes, maps[0].shape[2], maps[0].shape[3]), key
                loss = F.smooth_l1_loss(pred.reshape(-1).index_select(0, gt_idx), gt_sel, reduction="mean") + self.detection_loss(maps, self._tgt)
            else:
                pred = self.depth_pred(imgL, imgR)
                loss = F.smooth_l1_loss(pred.reshape(-1).index_select(0, gt_idx), gt_sel, reduction="mean")
            loss.backward()
            return loss.detach(), h.take()

    def inject_fake_target(self, extra, centers_l, centers_r, radius):
        """attack/DSGN/patch_attack.py:336-354 for the detection head: every ground-truth box is dropped and ONE fake car is
        put where the reference puts it - box3d (h, w, l, x, y, z, theta) = (1.65, 1.67, 3.64, -0.78, 1.98, 29.11, -1.60)"""
        if self.dsgn_head and hasattr(extra, "boxes"):
            from . import patchgeom
            h, w, l, x, y, z, th = patchgeom.DSGN_FAKE_BOX3D
            extra.boxes = [[(float(x), float(z), float(l), float(w), float(th))] for _ in extra.boxes]


class DsgnAdapter:
    """attack/DSGN/pgd_attack.py:300-336 around an upstream DSGN ``StereoNet`` (eval mode) and its
    ``RPN3DLoss``.  ``extra`` must carry: calibs_fu, calibs_baseline, calibs_Proj, calibs_Proj_R (:262-266),
    disp_true [B,H,W], targets, calib, calib_R, ious, labels_map (the BatchCollator fields, :103-126).
    Written for batch 1 per sample exactly as the reference indexes it (``o[mask[0]]``, quirk Q3)."""

    DISP_WEIGHTS = [0.5, 0.7, 1.0]          # :313

    def __init__(self, model, cfg, rpn3d_loss_cls, freeze=True):
        self.model, self.cfg, self.rpn3d_loss_cls = model, cfg, rpn3d_loss_cls
        if freeze:
            _freeze(model)

    @deterministic
    def loss_and_grad(self, x, extra):
        cfg = self.cfg
        h = _LeafGrad(x)
        with h as leaf:
            imgL, imgR = split_eyes(leaf)
            outputs = self.model(imgL, imgR, extra.calibs_fu, extra.calibs_baseline, extra.calibs_Proj,
                                 calibs_Proj_R=extra.calibs_Proj_R)                              # :308
            loss = 0.
            if getattr(cfg, "PlaneSweepVolume", True) and cfg.loss_disp:                         # :310-319
                disp_true = extra.disp_true
                mask = ((disp_true > cfg.min_depth) & (disp_true <= cfg.max_depth)).detach()     # :269-270
                depth_preds = [torch.squeeze(o, 1) for o in outputs["depth_preds"]]
                wts = self.DISP_WEIGHTS
                for i, o in enumerate(depth_preds):
                    loss = loss + wts[3 - len(depth_preds) + i] * F.smooth_l1_loss(o[mask[0]], disp_true[mask],
                                                                                    reduction="mean")
            if cfg.RPN3D_ENABLE:                                                                 # :321-330
                rpn3d_loss = self.rpn3d_loss_cls(cfg)(
                    outputs["bbox_cls"], outputs["bbox_reg"], outputs["bbox_centerness"], extra.targets,
                    extra.calib, extra.calib_R, ious=extra.ious, labels_map=extra.labels_map)[0]
                loss = loss + rpn3d_loss
            self.model.zero_grad()                                                               # :333
            loss.backward()                                                                      # :336
            return loss.detach(), h.take()

    def inject_fake_target(self, extra, centers_l, centers_r, radius):
        from . import patchgeom
        if extra.targets is not None:                                                            # patch_attack.py:336
            patchgeom.inject_fake_target_dsgn(extra.targets[0].bbox.data, extra.targets[0].box3d.data)


class StereoRcnnAdapter:
    """attack/Stereo-RCNN/pgd_attack.py:151-174 around the upstream ``_StereoRCNN`` with the substitute
    files of the reference (losses computed in eval mode).  ``extra``: im_info, gt_boxes_left/right/merge,
    gt_dim_orien, gt_kpts, num_boxes; ``uncert`` = the six learned log-variances from the checkpoint (:97)."""

    def __init__(self, model, uncert, freeze=True):
        self.model, self.uncert = model, uncert
        if freeze:
            _freeze(model)

    @property
    def graph_safe(self):
        """capturable in a hipGraph (attacks.PgdAttack(graph=True)): only a detector that says so itself - the layer-list surrogates on
        their static path (surrogates.StereoRcnnShaped.graph_capturable); upstream's proposal layers read back and compact"""
        return bool(getattr(self.model, "graph_capturable", False))

    _EXTRA_TENSORS = ("im_info", "gt_boxes_left", "gt_boxes_right", "gt_boxes_merge", "gt_dim_orien", "gt_kpts", "num_boxes")

    def graph_extra_signature(self, extra):
        """what of a label set is a HOST constant of a captured iteration (attacks.PgdAttack(graph=True)): image size, number of boxes,
        tensor shapes.  Two label sets with the same signature can share a capture - their tensors are copied into the captured ones."""
        m = self.model
        if not getattr(m, "graph_capturable", False) or not all(torch.is_tensor(getattr(extra, k, None)) and getattr(extra, k).is_cuda
                                                                 for k in self._EXTRA_TENSORS):
            return None
        return (tuple(m._host_values(extra.im_info, 2)), int(m._host_values(extra.num_boxes, 1)[0])) + \
            tuple(tuple(getattr(extra, k).shape) for k in self._EXTRA_TENSORS)

    def graph_copy_extra(self, dst, src):
        for k in self._EXTRA_TENSORS:
            getattr(dst, k).copy_(getattr(src, k))

    def graph_clone_extra(self, extra):
        import types
        out = types.SimpleNamespace(**vars(extra))
        for k in self._EXTRA_TENSORS:
            setattr(out, k, getattr(extra, k).clone())
        return out

    @deterministic
    def loss_and_grad(self, x, extra):
        u = self.uncert
        h = _LeafGrad(x)
        with h as leaf:
            imgL, imgR = split_eyes(leaf)
            out = self.model(imgL, imgR, extra.im_info, extra.gt_boxes_left, extra.gt_boxes_right,
                             extra.gt_boxes_merge, extra.gt_dim_orien, extra.gt_kpts, extra.num_boxes)   # :156-163
            terms = out[8:14]   # rpn_loss_cls, rpn_loss_box_left_right, RCNN_loss_cls, _bbox, _dim_orien, _kpts
            if leaf.is_cuda and u.is_cuda and not u.requires_grad and all(t.numel() == 1 for t in terms):
                # the same six terms in the same order of float operations as one launch (ops.ObjectiveChain; a one-element mean is the element)
                from . import ops
                loss = ops.ObjectiveChain.apply(torch.cat([t.reshape(1) for t in terms]), u)
            else:
                loss = 0.
                for k in range(6):                                                               # :165-171
                    loss = loss + terms[k].mean() * torch.exp(-u[k]) + u[k]
            self.model.zero_grad()                                                               # :173
            loss.backward()                                                                      # :174
            return loss.detach(), h.take()

    def inject_fake_target(self, extra, centers_l, centers_r, radius):
        from . import patchgeom
        extra.num_boxes.data = torch.tensor(1)                                                   # patch_attack.py:188
        patchgeom.inject_fake_target_srcnn(extra.gt_boxes_left, extra.gt_boxes_right, extra.gt_boxes_merge,
                                           centers_l[0], centers_r[0], radius)


class _BilinearUp(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) with a DETERMINISTIC backward: torch's backward of the bilinear
    up-sampling scatters with atomicAdd (listed under torch.use_deterministic_algorithms), which would make the whole attack gradient
    differ from run to run in its last bits.  The up-sampling is linear and separable, up = A_h . x . A_w^T, so its adjoint is
    A_h^T . g . A_w - two small matrix products in a fixed order (the interpolation matrices are read off F.interpolate itself)."""

    _matrices = {}

    @staticmethod
    def _matrix(n_in, n_out, device):
        key = (n_in, n_out, device)
        m = _BilinearUp._matrices.get(key)
        if m is None:                           # once per (size pair, device): the matrices depend on nothing else
            eye = torch.eye(n_in, device=device).view(1, n_in, n_in, 1)
            m = F.interpolate(eye, size=(n_out, 1), mode="bilinear", align_corners=False)[0, :, :, 0].t().contiguous()     # [n_out, n_in]
            _BilinearUp._matrices[key] = m
        return m

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = tuple(x.shape[2:])
        ctx.size = tuple(size)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)

    @staticmethod
    def backward(ctx, g):
        (h, w), (ho, wo) = ctx.in_hw, ctx.size
        ah, aw = _BilinearUp._matrix(h, ho, g.device), _BilinearUp._matrix(w, wo, g.device)
        return torch.matmul(ah.t(), torch.matmul(g.contiguous(), aw)), None


class DsgnShapedAdapter(PsvStereoAdapter):
    """The DSGN-SHAPED graph with SURVEY App. B's LAYER LIST (all of it [UPSTREAM-UNVERIFIED]: DSGN's source is not in the reference
    tree - these are the published PSMNet / DSGN structures, random weights, batch-norms folded into biases):
      2D extractor (per eye, shared)  3 x conv3x3 32 (first stride 2) - 3 residual blocks 32 @1/2 - 16 blocks 64 @1/4 (first stride 2) -
                                      3 blocks 128 - 3 blocks 128 dilation 2 - four average-pool branches (64/32/16/8) -> 1x1 32 -> up -
                                      concat 320 -> conv3x3 128 -> conv1x1 32                                    (torch / MIOpen)
      plane-sweep volume              [B,64,48,H/4,W/4], fractional disparities                                 (csrc/psv.hip)
      dres0 / dres1                   64->32, 32->32 | 32->32, 32->32 (+ skip)                                  (csrc/conv3d.hip, float32 MFMA)
      3D hourglass                    32->64 /2, 64->64, 64->64 /2, 64->64, up 64->64 (+ skip), up 64->32 (+ skip)
      classif                         32->32, 32->1 -> fused trilinear up-sampling + softmax + expectation      (csrc/volume.hip)
      3D geometric volume             features x plane probability -> grid_sample -> [B,32,192,20,304]          (csrc/volume.hip)
      3DGV stack                      32->64, 3D hourglass 64->128 /2, 128, 128 /2, 128, up 128 (+ skip), up 64 (+ skip)
      bird's-eye view                 height pooled 20 -> 5 and folded into channels (320) -> conv3x3 128 -> 2D hourglass
                                      (128->256 /2, 256, 256 /2, 256, up 256 (+ skip), up 128 (+ skip))       (torch / MIOpen)
      heads                           two towers of 4 x conv3x3 128 -> class / box (7 per anchor) / centerness maps
    ``flops_fwd`` accumulates 2 x MACs of every convolution of a forward pass; a detector step (forward + backward w.r.t. the images)
    costs twice that.  ``torch_ops=True`` computes the same graph with torch's own operators (the parity reference of the tests)."""

    def __init__(self, device, seed=0, planes=48, image_hw=(384, 1248), fu=721.5377, cu=609.5593, cv=172.854, mfma_conv=True, torch_ops=False,
                 hip2d="auto", wino3d=True):
        self.wino3d = bool(wino3d)          # the stride-1 3x3x3 layers may take the Winograd kernel where it measures faster (ops.Conv3dK3 wino=)
        super().__init__(device, seed=seed, channels=32, planes=planes, mid=32, mfma_conv=mfma_conv and not torch_ops, interp=True,
                         hourglass=False, dsgn_head=True, fu=fu, cu=cu, cv=cv, image_hw=image_hw)
        self.torch_ops = bool(torch_ops)
        # hip2d: the 1x1 and 3x3 stride-1 2D layers (extractor, bird's-eye view, heads) on csrc/conv2d.hip; False: torch / MIOpen
        self.hip2d, self._p2 = hip2d, {}                   # True: always, "auto": where it measures faster than MIOpen, False: never
        self.flops_fwd = 0
        gen = torch.Generator().manual_seed(seed + 1000)
        dev = device

        def conv(cout, cin, *k, gain=1.0):
            fan = cin
            for v in k:
                fan *= v
            return (torch.randn(cout, cin, *k, generator=gen) * (gain * (2.0 / fan) ** 0.5)).to(dev), (torch.randn(cout, generator=gen) * 0.02).to(dev)

        # ---- 2D extractor: name -> (weight, bias, stride, padding, dilation)
        w2 = {}

        def add2(name, cout, cin, k, stride=1, dilation=1, gain=1.0):
            w, b = conv(cout, cin, k, k, gain=gain)
            w2[name] = (w, b, stride, dilation * (k // 2), dilation)

        add2("f0a", 32, 3, 3, stride=2)
        add2("f0b", 32, 32, 3)
        add2("f0c", 32, 32, 3)
        self.blocks = []                                   # (prefix, has projection)
        cin = 32
        for li, (width, n, stride, dil) in enumerate(((32, 3, 1, 1), (64, 16, 2, 1), (128, 3, 1, 1), (128, 3, 1, 2)), start=1):
            for i in range(n):
                pre = "l%d.%d" % (li, i)
                s = stride if i == 0 else 1
                add2(pre + ".a", width, cin, 3, stride=s, dilation=dil)
                add2(pre + ".b", width, width, 3, dilation=dil, gain=0.3)
                proj = s != 1 or cin != width
                if proj:
                    add2(pre + ".p", width, cin, 1, stride=s)
                self.blocks.append((pre, proj, li))
                cin = width
        for k in (64, 32, 16, 8):
            add2("spp%d" % k, 32, 128, 1)
        add2("last_a", 128, 320, 3)
        add2("last_b", 32, 128, 1, gain=0.15)
        # ---- bird's-eye view + heads
        add2("bev_a", 128, 320, 3)
        add2("bh1", 256, 128, 3, stride=2)
        add2("bh2", 256, 256, 3)
        add2("bh3", 256, 256, 3, stride=2)
        add2("bh4", 256, 256, 3)
        for t in ("ct", "rt"):
            for i in range(4):
                add2("%s%d" % (t, i), 128, 128, 3)
        a = self.ANCHORS
        add2("head_cls", a, 128, 3, gain=0.1)
        add2("head_reg", 7 * a, 128, 3, gain=0.1)
        add2("head_ctr", a, 128, 3, gain=0.1)
        self.w2 = w2
        self.wt2 = {"bh5": conv(256, 256, 3, 3), "bh6": conv(256, 128, 3, 3)}      # ConvTranspose2d layout [in, out, 3, 3]; bias [in]-shaped draw reused below
        self.wt2 = {k: (w, (torch.randn(w.shape[1], generator=gen) * 0.02).to(dev)) for k, (w, _) in self.wt2.items()}
        # ---- 3D layers: name -> dict(kind, weight, bias, cout, prepared weights)
        ops = self.ops
        self.w3 = {}

        def add3(name, kind, cout, cin, gain=1.0):
            if kind == "t2":                               # ConvTranspose3d layout [in, out, 3,3,3]
                w, _ = conv(cin, cout, 3, 3, 3, gain=gain)
                b = (torch.randn(cout, generator=gen) * 0.02).to(dev)
            else:
                w, b = conv(cout, cin, 3, 3, 3, gain=gain)
            e = {"kind": kind, "w": w, "b": b, "cout": cout, "cin": cin}
            if self.mfma_conv:
                if kind == "s1":
                    e["p"], e["pt"] = ops.conv3d_k3_prep(w), ops.conv3d_k3_prep(w, transpose=True)
                    e["wino"] = ops.Conv3dWinoPrep(w) if self.wino3d and cout >= 4 else None      # direct or Winograd: whichever measures faster per shape
                elif kind == "s2":
                    e["p"], e["pt"] = ops.conv3d_k3_s2_prep(w), ops.conv_transpose3d_k3_s2_prep(w)
                else:
                    e["p"], e["pt"] = ops.conv_transpose3d_k3_s2_prep(w), ops.conv3d_k3_s2_prep(w)
            self.w3[name] = e

        for name, kind, cout, cin, gain in (("dres0a", "s1", 32, 64, 1), ("dres0b", "s1", 32, 32, 1), ("dres1a", "s1", 32, 32, 1), ("dres1b", "s1", 32, 32, 0.3),
                                            ("hg1", "s2", 64, 32, 1), ("hg2", "s1", 64, 64, 1), ("hg3", "s2", 64, 64, 1), ("hg4", "s1", 64, 64, 1),
                                            ("hg5", "t2", 64, 64, 0.5), ("hg6", "t2", 32, 64, 0.5), ("cls_a", "s1", 32, 32, 1), ("cls_b", "s1", 1, 32, 1),
                                            ("gv1", "s1", 64, 32, 1), ("gh1", "s2", 128, 64, 1), ("gh2", "s1", 128, 128, 1), ("gh3", "s2", 128, 128, 1),
                                            ("gh4", "s1", 128, 128, 1), ("gh5", "t2", 128, 128, 0.5), ("gh6", "t2", 64, 128, 0.5)):
            add3(name, kind, cout, cin, gain)

    # -- layer helpers (each counts its forward FLOPs) -------------------------------------------------------------------
    trace2d = None        # a list: every 2D convolution call appends (cin, cout, k, stride, padding, dilation, batch, h, w) - tools/bench_conv2d_layers.py

    def _c2(self, x, name, relu=False, residual=None, chain_in=False, undilated=False, skip_out=False):
        """``relu="consumer"`` / ``chain_in=True``: see ops.Conv2dAuto - a ReLU layer whose ONLY consumer is the next convolution leaves
        its backward mask to that consumer's dgrad epilogue (no relu_backward pass over the tensor).  ``undilated``: x holds the four
        parity sub-images of the layer's input side by side in the batch (_parity_split): the dilation-2 layer is a dilation-1 layer on each"""
        w, b, s, p, d = self.w2[name]
        if undilated:
            assert d == 2 and p == 2 and s == 1
            p, d, name = 1, 1, name + "@parity"
        if DsgnShapedAdapter.trace2d is not None:
            DsgnShapedAdapter.trace2d.append((w.shape[1], w.shape[0], w.shape[2], s, p, d, x.shape[0], x.shape[2], x.shape[3]))
        if self.hip2d and not self.torch_ops and self.ops.conv2d_supported(x, w, s, p, d):
            # libadvengine's float32-MFMA 2D kernels: bias, skip connection and ReLU in the epilogue (no element-wise passes)
            if name not in self._p2:
                self._p2[name] = self.ops.Conv2dPrep(w, s, p, d)
            if self.hip2d == "auto":       # per layer shape and direction, whichever of {libadvengine, MIOpen} measured faster
                y = self.ops.Conv2dAuto.apply(x, self._p2[name], w, b, residual, relu, chain_in, skip_out)
                if skip_out:
                    self.flops_fwd += 2 * y[0].numel() * w.shape[1] * w.shape[2] * w.shape[3]
                    return y
            else:
                assert not chain_in and relu != "consumer" and not skip_out
                y = self.ops.Conv2d.apply(x, self._p2[name], b, residual, relu)
            self.flops_fwd += 2 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            return y
        assert not chain_in and relu != "consumer" and not skip_out, "chained ReLU masks / fused skip gradients need the Conv2dAuto path"
        y = F.conv2d(x, w, b, s, p, d)
        self.flops_fwd += 2 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y

    @staticmethod
    def _parity_split(x):
        """[B,C,H,W] -> [4B,C,H/2,W/2]: sub-image (row parity, column parity) of every map, (b, py, px) in the batch dimension"""
        b, c, h, w = x.shape
        return x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(4 * b, c, h // 2, w // 2)

    @staticmethod
    def _parity_merge(xs, b):
        _, c, h, w = xs.shape
        return xs.view(b, 2, 2, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(b, c, 2 * h, 2 * w)

    def _chain(self, *names):
        """can the ReLU masks between these consecutive layers be left to the consumers?  (all of them on the Conv2dAuto path)"""
        if self.hip2d != "auto" or self.torch_ops:
            return False
        for n in names:
            w, b, s, p, d = self.w2[n]
            if not (s == 1 and ((w.shape[2] == 1 and p == 0 and d == 1) or (w.shape[2] == 3 and d in (1, 2) and p == d))):
                return False
        return True

    def _ct2(self, x, name, relu=False, residual=None):
        w, b = self.wt2[name]
        self.flops_fwd += 2 * x.numel() * w.shape[1] * 9
        if not self.torch_ops and x.is_cuda:
            # the transposed convolution (with its bias) is MIOpen's; skip connection + ReLU are ONE pass behind it (ops.BiasAct) instead
            # of torch's add / relu and, in the backward, threshold + add: same float operations in the same order
            y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
            return self.ops.BiasAct.apply(y, None, relu, residual) if (relu or residual is not None) else y
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y

    def _c3(self, x, name, relu=False, residual=None, chain_in=False, skip_out=False):
        """``relu="consumer"`` / ``chain_in``: as _c2 - a ReLU layer whose only consumer is a stride-1 layer leaves its backward mask to that
        consumer's dgrad epilogue (ops.Conv3dK3 mask_input); only on the libadvengine path.  <round 4> stride-2 layers take both too, and
        ``skip_out``: -> (y, x_skip) - see ops.Conv3dK3S2"""
        e, ops = self.w3[name], self.ops
        kind, w, b, cout = e["kind"], e["w"], e["b"], e["cout"]
        if not self.mfma_conv:
            assert not chain_in and relu != "consumer" and not skip_out
        assert not skip_out or kind in ("s2", "s1")
        if kind == "t2":
            self.flops_fwd += 2 * x.numel() * cout * 27
        if self.mfma_conv:
            if kind == "s1":
                if cout < 4:                                   # the 32 -> 1 score layer: the narrow kernels, no bias
                    y = ops.Conv3dK3.apply(x, e["p"], e["pt"], cout, None, None, False, None, chain_in)      # (chain_in: its adjoint masks)
                    y = y + b.view(1, -1, 1, 1, 1)
                else:
                    y = ops.Conv3dK3.apply(x, e["p"], e["pt"], cout, None, b, relu, residual, chain_in, e["wino"], skip_out)
                    if skip_out:
                        self.flops_fwd += 2 * y[0].numel() * e["cin"] * 27
                        return y
            elif kind == "s2":
                y = ops.Conv3dK3S2.apply(x, e["p"], e["pt"], cout, b, relu, chain_in, skip_out)
                if skip_out:
                    self.flops_fwd += 2 * y[0].numel() * e["cin"] * 27
                    return y
            else:
                y = ops.ConvTranspose3dK3S2.apply(x, e["p"], e["pt"], cout, b, relu, residual)
        else:
            if kind == "t2":
                y = F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1)
            else:
                y = F.conv3d(x, w, b, stride=2 if kind == "s2" else 1, padding=1)
            if residual is not None:
                y = y + residual
            if relu:
                y = F.relu(y)
        if kind != "t2":
            self.flops_fwd += 2 * y.numel() * e["cin"] * 27
        return y

    # -- the graph ----------------------------------------------------------------------------------------------------------
    def features(self, img):
        # f0a (strided: torch's) -> f0b -> f0c -> the first block: f0b's ReLU mask is left to f0c's backward launch, f0c's to the first block's
        # first layer (which also takes the skip path's gradient: the mask then covers both) - where all of them are on the Conv2dAuto path
        first = self.blocks[0][0]
        bc = self._chain("f0b", "f0c") and img.is_cuda
        c1 = bc and self._chain("f0c", first + ".a", first + ".b") and not self.blocks[0][1] and img.is_cuda and \
            not (self.w2[first + ".a"][4] == 2 and self.w2[first + ".b"][4] == 2)
        x = self._c2(self._c2(img, "f0a", True), "f0b", "consumer" if bc else True)
        x = self._c2(x, "f0c", "consumer" if c1 else True, chain_in=bc)
        mask_first = c1
        outs = {}
        split = 0          # > 0: x is the four parity sub-images of a batch of `split` maps (the dilation-2 blocks run as dilation-1 blocks on them)
        for pre, proj, li in self.blocks:
            dil2 = self.w2[pre + ".a"][4] == 2 and self.w2[pre + ".b"][4] == 2 and not proj
            if dil2 and not split and self.hip2d == "auto" and not self.torch_ops and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                # a dilation-2 3x3 convolution only ever combines pixels of one (row, column) parity: on the four parity sub-images it
                # is a dilation-1 convolution with the same weights - which has a Winograd kernel (1.7x the dilated direct kernel's rate)
                split, x = x.shape[0], self._parity_split(x)
            elif split and not dil2:
                x, split = self._parity_merge(x, split), 0
            ch = self._chain(pre + ".a", pre + ".b")                                          # a's only consumer is b
            if ch and not proj and x.is_cuda:
                # identity block: layer a hands x on as the skip tensor and its backward adds the skip path's gradient in the dgrad
                # kernel's epilogue (ops.Conv2dAuto skip_out) - no element-wise addition by the autograd engine
                t, idt = self._c2(x, pre + ".a", "consumer", undilated=bool(split), skip_out=True, chain_in=mask_first)
                mask_first = False
            else:
                assert not mask_first, "f0c left its ReLU mask to a block that does not take it"
                idt = self._c2(x, pre + ".p") if proj else x
                t = self._c2(x, pre + ".a", "consumer" if ch else True, undilated=bool(split))
            x = self._c2(t, pre + ".b", residual=idt, chain_in=ch, undilated=bool(split))      # PSMNet's BasicBlock: no ReLU after the sum
            outs[li] = x
        if split:
            x = self._parity_merge(x, split)
            outs[li] = x
        l2, l4 = outs[2], outs[4]
        branches = []
        pooled = {}
        if l4.shape[2] >= 64 and l4.shape[3] >= 64:
            # the four pooling windows nest (8 | 16 | 32 | 64, same origin, floor): pool 8x8 once and halve three times - torch's avg_pool2d
            # with a 64 x 64 window has four output columns' worth of threads summing 4096 elements each (0.6 ms a call at 96 x 312)
            pooled[8] = F.avg_pool2d(l4, 8)
            for k in (16, 32, 64):
                pooled[k] = F.avg_pool2d(pooled[k // 2], 2)
        for k in (64, 32, 16, 8):
            if k in pooled:
                p = pooled[k]
            else:
                kh, kw = min(k, l4.shape[2]), min(k, l4.shape[3])
                p = F.avg_pool2d(l4, (kh, kw), stride=(kh, kw))
            br = self._c2(p, "spp%d" % k, True)
            branches.append(_BilinearUp.apply(br, tuple(l4.shape[2:])))
        cat = torch.cat([l2, l4] + branches, 1)
        lc = self._chain("last_a", "last_b") and cat.is_cuda
        return self._c2(self._c2(cat, "last_a", "consumer" if lc else True), "last_b", chain_in=lc)

    def _volume_net_impl(self, cost):
        ch = self.mfma_conv                 # chains: a ReLU output with ONE consumer, and that consumer a stride-1 layer on the main kernel
        if ch:
            # dres0b's output feeds dres1a AND dres1's skip connection: dres1a's backward takes both gradients and dres0b's ReLU mask over their
            # sum in its dgrad launch (ops.Conv3dK3 skip_out + mask_input) - no relu_backward pass and no autograd addition over the 184 MB volume
            c0 = self._c3(self._c3(cost, "dres0a", "consumer"), "dres0b", "consumer", chain_in=True)
            t, c0s = self._c3(c0, "dres1a", "consumer", chain_in=True, skip_out=True)
            c0 = self._c3(t, "dres1b", False, c0s, chain_in=True)
        else:
            c0 = self._c3(self._c3(cost, "dres0a", True), "dres0b", True)
            c0 = self._c3(self._c3(c0, "dres1a", True), "dres1b", False, c0)
        if ch:
            # the tensors a down-sampling layer reads also feed the matching up-sampling layer's skip connection: the down-sampling layer's
            # backward adds the skip path's gradient (and applies the producer's ReLU mask) in its own launch - ops.Conv3dK3S2 skip_out / mask_input
            t, c0s = self._c3(c0, "hg1", "consumer", skip_out=True)                          # c0 carries no ReLU
            pre = self._c3(t, "hg2", "consumer", chain_in=True)                               # hg2's mask: in hg3's backward
            t, pres = self._c3(pre, "hg3", "consumer", chain_in=True, skip_out=True)
            h = self._c3(t, "hg4", True, chain_in=True)
            post = self._c3(h, "hg5", True, pres)
            out = self._c3(post, "hg6", False, c0s)
        else:
            pre = self._c3(self._c3(c0, "hg1", True), "hg2", True)
            h = self._c3(self._c3(pre, "hg3", True), "hg4", True)
            post = self._c3(h, "hg5", True, pre)
            out = self._c3(post, "hg6", False, c0)
        # cls_a's ReLU mask: in cls_b's backward launch (the 1 -> 32 narrow-input kernel's epilogue), not a pass over the 184 MB volume
        score = self._c3(self._c3(out, "cls_a", "consumer" if ch else True), "cls_b", chain_in=ch)
        return score.squeeze(1), out

    def detection_maps(self, feat_vol, cost):
        ops = self.ops
        prob = torch.softmax(cost, dim=1)
        b = feat_vol.shape[0]
        vol = (feat_vol * prob[:, None]).contiguous()
        if self.torch_ops:
            gv = F.grid_sample(vol, self.gv_grid.expand(b, -1, -1, -1, -1), mode="bilinear", padding_mode="zeros", align_corners=True)
        else:
            grid, plan = self._gv(b)
            gv = ops.GridSample3d.apply(vol, grid, plan)                                                  # [B,32,Zg,Yg,Xg]
        ch = self.mfma_conv
        if ch:      # as in _volume_net_impl: the skip tensors' gradients and ReLU masks meet in the down-sampling layers' backward launches
            g1 = self._c3(gv, "gv1", "consumer")
            t, g1s = self._c3(g1, "gh1", "consumer", chain_in=True, skip_out=True)
            pre = self._c3(t, "gh2", "consumer", chain_in=True)
            t, pres = self._c3(pre, "gh3", "consumer", chain_in=True, skip_out=True)
            h = self._c3(t, "gh4", True, chain_in=True)
            post = self._c3(h, "gh5", True, pres)
            g = self._c3(post, "gh6", "consumer", g1s)                     # its only consumer is the bird's-eye-view fold, whose backward masks
        else:
            g1 = self._c3(gv, "gv1", True)
            pre = self._c3(self._c3(g1, "gh1", True), "gh2", True)
            h = self._c3(self._c3(pre, "gh3", True), "gh4", True)
            post = self._c3(h, "gh5", True, pre)
            g = self._c3(post, "gh6", True, g1)
        bb, c, zg, yg, xg = g.shape
        if self.torch_ops or not g.is_cuda:
            bev = F.avg_pool3d(g, (1, self.ypool, 1)).permute(0, 1, 3, 2, 4).reshape(bb, c * (yg // self.ypool), zg, xg)
        else:       # the same values, one pass each way (and gh6's ReLU backward inside the fold's, where gh6 left it to us)
            bev = ops.BevFold.apply(g, self.ypool, bool(ch))
        b0 = self._c2(bev, "bev_a", True)
        pre2 = self._c2(self._c2(b0, "bh1", True), "bh2", True)
        h2 = self._c2(self._c2(pre2, "bh3", True), "bh4", True)
        post2 = self._ct2(h2, "bh5", True, pre2)
        x = self._ct2(post2, "bh6", True, b0)
        ct = rt = x
        chain = self._chain("ct0", "ct1", "head_cls")
        for i in range(4):                      # the class tower is a chain (each layer's only consumer is the next one; ct3 -> head_cls);
            ct = self._c2(ct, "ct%d" % i, "consumer" if chain else True, chain_in=chain and i > 0)      # x itself feeds both towers: no chain into ct0
            rt = self._c2(rt, "rt%d" % i, "consumer" if (chain and i < 3) else True, chain_in=chain and i > 0)   # rt3 feeds two heads: it masks itself
        return self._c2(ct, "head_cls", chain_in=chain) + self.cls_bias, self._c2(rt, "head_reg"), self._c2(rt, "head_ctr")

    def forward_all(self, imgL, imgR):
        b = imgL.shape[0]
        f = self.features(torch.cat([imgL, imgR], 0))                      # both eyes through the shared extractor as one batch
        fl, fr = f.split(b, 0)                                             # one split node (contiguous halves): its backward is one concatenation
        cost = self.ops.PsvBuildLerp.apply(fl.contiguous(), fr.contiguous(), self.shifts(b))
        score, feat = self._volume_net_impl(cost)
        if self.torch_ops:
            up = F.interpolate(score[:, None], size=self.up_size, mode="trilinear", align_corners=False)[:, 0]
            depth = (torch.softmax(up, 1) * self.depth_up.view(1, -1, 1, 1)).sum(1)
        else:
            depth = self.ops.DepthRegress.apply(score.contiguous(), self.depth_up, self.up_size, False)
        return depth, self.detection_maps(feat, score)

    def detection_loss(self, maps, targets):
        if not self.torch_ops:
            return super().detection_loss(maps, targets)
        cls, reg, ctr = maps
        t = targets
        logits = cls.permute(0, 2, 3, 1).reshape(-1)
        tt = t["cls"].float()
        p = torch.sigmoid(logits)
        focal = -(tt * 0.25 * (1 - p) ** 2 * F.logsigmoid(logits) + (1 - tt) * 0.75 * p ** 2 * F.logsigmoid(-logits)).sum()
        npos = t["npos"]
        return focal / npos + F.smooth_l1_loss(reg.reshape(-1).index_select(0, t["posr_idx"]), t["reg_pos"], reduction="sum") / npos + \
            F.binary_cross_entropy_with_logits(ctr.reshape(-1).index_select(0, t["pos_idx"]), t["ctr_pos"], reduction="sum") / npos

    def flops_per_step(self, x, extra):
        """2 x MACs of every convolution of ONE detector step (forward + backward w.r.t. the images = 2 x the forward count)"""
        self.flops_fwd = 0
        self.loss_and_grad(x, extra)
        return 2 * self.flops_fwd
