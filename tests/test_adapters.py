"""The objective glue of the upstream-model adapters (attack/DSGN/pgd_attack.py:300-336 and
attack/Stereo-RCNN/pgd_attack.py:151-174) exercised with stand-in model objects that have the upstream
call signatures, and the KITTI folder reader on synthetic PNGs.  CPU only."""
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

from eval_driving_safety_amd import adapters, data


class _FakeDsgn(torch.nn.Module):
    """same call signature and output dict as StereoNet (pgd_attack.py:215-222,308)"""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.5))
        self.calls = []

    def forward(self, imgL, imgR, fu, baseline, proj, calibs_Proj_R=None):
        self.calls.append((fu, baseline, proj, calibs_Proj_R))
        depth = 10 + self.w * (imgL.mean(dim=1) - imgR.mean(dim=1))          # [B,H,W]
        return {"depth_preds": depth,        # eval mode: a [B,H,W] tensor that the script iterates over (pgd_attack.py:311)
                 "bbox_cls": imgL.sum(), "bbox_reg": imgR.sum(), "bbox_centerness": imgL.mean()}


class _FakeRpnLoss:
    def __init__(self, cfg):
        pass

    def __call__(self, cls, reg, ctr, targets, calib, calib_R, ious=None, labels_map=None):
        total = 0.01 * cls + 0.02 * reg + ctr
        return total, cls, reg, ctr


def test_dsgn_adapter_objective_and_gradient():
    torch.manual_seed(0)
    B, H, W = 1, 6, 8
    x = torch.randn(2 * B, 3, H, W)
    disp = torch.rand(B, H, W) * 50
    cfg = types.SimpleNamespace(PlaneSweepVolume=True, loss_disp=True, RPN3D_ENABLE=True, min_depth=2.0, max_depth=40.4)
    extra = types.SimpleNamespace(calibs_fu=torch.tensor([721.5]), calibs_baseline=torch.tensor([0.54]), calibs_Proj=torch.zeros(1, 3, 4),
                                  calibs_Proj_R=torch.ones(1, 3, 4), disp_true=disp, targets=None, calib=None, calib_R=None,
                                  ious=None, labels_map=None)
    model = _FakeDsgn()
    ad = adapters.DsgnAdapter(model, cfg, _FakeRpnLoss)
    loss, grad = ad.loss_and_grad(x, extra)
    assert model.calls[0][3] is extra.calibs_Proj_R and not x.requires_grad and x.grad is None
    # the same objective written out directly
    xr = x.clone().requires_grad_(True)
    imgL, imgR = xr[:B], xr[B:]
    depth = 10 + 0.5 * (imgL.mean(dim=1) - imgR.mean(dim=1))
    mask = (disp > 2.0) & (disp <= 40.4)
    want = 1.0 * F.smooth_l1_loss(depth[mask[0]][None] if False else depth[0][mask[0]], disp[mask]) \
        + 0.01 * imgL.sum() + 0.02 * imgR.sum() + imgL.mean()
    want.backward()
    assert torch.allclose(loss, want.detach(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(grad, xr.grad, rtol=1e-5, atol=1e-7)
    assert model.w.grad is None                          # the detector's weights are constants: no weight gradients by default
    model2 = _FakeDsgn()
    loss2, grad2 = adapters.DsgnAdapter(model2, cfg, _FakeRpnLoss, freeze=False).loss_and_grad(x, extra)
    assert model2.w.grad is not None                     # freeze=False: the reference's behaviour (model.zero_grad(), then backward)
    assert torch.equal(grad2, grad) and torch.equal(loss2, loss)      # the image gradient does not depend on it


class _FakeSrcnn(torch.nn.Module):
    """nine inputs -> fifteen outputs, the six losses at positions 8..13 (stereo_rcnn.py:324-326)"""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.ones(1))

    def forward(self, l, r, info, gl, gr, gm, gdo, gk, nb):
        losses = [(l * l).mean(dim=(1, 2, 3)), (r * r).mean(dim=(1, 2, 3)), (l * r).mean(dim=(1, 2, 3)), l.abs().mean(dim=(1, 2, 3)),
                  r.abs().mean(dim=(1, 2, 3)), (l - r).pow(2).mean(dim=(1, 2, 3))]
        return tuple([None] * 8 + [v * self.p for v in losses] + [None])


def test_stereo_rcnn_adapter_uncertainty_weighting():
    torch.manual_seed(1)
    x = torch.randn(2, 3, 5, 7)
    u = torch.tensor([0.1, -0.2, 0.3, 0.0, 0.5, -0.4])
    extra = types.SimpleNamespace(im_info=None, gt_boxes_left=None, gt_boxes_right=None, gt_boxes_merge=None, gt_dim_orien=None,
                                  gt_kpts=None, num_boxes=None)
    loss, grad = adapters.StereoRcnnAdapter(_FakeSrcnn(), u).loss_and_grad(x, extra)
    xr = x.clone().requires_grad_(True)
    l, r = xr[:1], xr[1:]
    terms = [(l * l).mean(), (r * r).mean(), (l * r).mean(), l.abs().mean(), r.abs().mean(), (l - r).pow(2).mean()]
    want = sum(t * torch.exp(-u[k]) + u[k] for k, t in enumerate(terms))      # pgd_attack.py:165-171
    want.backward()
    assert torch.allclose(loss, want.detach(), rtol=1e-6) and torch.allclose(grad, xr.grad, rtol=1e-5, atol=1e-7)


def test_toy_adapter_is_deterministic_and_nontrivial():
    a = adapters.ToyStereoAdapter(torch.device("cpu"), seed=3)
    x = torch.randn(2, 3, 32, 48)
    l1, g1 = a.loss_and_grad(x.clone())
    l2, g2 = adapters.ToyStereoAdapter(torch.device("cpu"), seed=3).loss_and_grad(x.clone())
    assert torch.equal(g1, g2) and float(g1.abs().sum()) > 0 and float(g1[1].abs().sum()) > 0 and l1 == l2


def test_kitti_folder_reader(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    for eye in ("image_2", "image_3"):
        os.makedirs(tmp_path / eye)
        for name in ("000005", "000009"):
            Image.fromarray(rs.randint(0, 256, (375, 1242, 3)).astype(np.uint8)).save(str(tmp_path / eye / (name + ".png")))
    (tmp_path / "val.txt").write_text("000005\n000009\n")
    batches = list(data.KittiFolder(str(tmp_path), str(tmp_path / "val.txt"), batch=2))
    assert len(batches) == 1
    b = batches[0]
    assert tuple(b.imgL.shape) == (2, 3, 384, 1248) and b.names == ["000005", "000009"] and b.sizes == [(1242, 375)] * 2
    u8 = np.array(Image.open(str(tmp_path / "image_2" / "000005.png")))
    want = ((u8[0, 0].astype(np.float32) / np.float32(255) - np.float32(data.DSGN_MEAN)) / np.float32(data.DSGN_STD))
    assert np.allclose(b.imgL[0, :, 0, 0].numpy(), want, atol=1e-6)
    assert float(b.imgL[:, :, 375:, :].abs().max()) == 0 and float(b.imgL[:, :, :, 1242:].abs().max()) == 0
