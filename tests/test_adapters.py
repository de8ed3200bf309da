"""The objective glue of the upstream-model adapters (attack/DSGN/pgd_attack.py:300-336 and
attack/Stereo-RCNN/pgd_attack.py:151-174) pinned against the reference's own statements, executed by
tests/golden/make_golden.py around stand-in model objects with the upstream call signatures (tests/golden/stub_models.py), and the KITTI folder reader on synthetic PNGs.  CPU only."""
import os
import types

import numpy as np
import pytest
import torch

import stub_models
import synth
from eval_driving_safety_amd import adapters, data


def _same(got, want, what):
    """bit for bit on the CPU family the fixture was generated on; another SIMD width may round oneDNN's convolutions of the stand-in
    detector differently - then (and only then) 1e-6 relative, with a warning naming the case"""
    got = got.detach().numpy() if hasattr(got, "detach") else np.asarray(got)
    if got.tobytes() == want.tobytes():
        return
    import warnings
    assert got.shape == want.shape and np.allclose(got, want, rtol=1e-6, atol=1e-7 * float(np.abs(want).max())), what
    warnings.warn("%s: equal to 1e-6 but not bit for bit (different CPU kernels than the fixture's host?)" % what)


def _dsgn_inputs(h, w):
    """the inputs tests/golden/make_golden.py:objective_cases() drew (same seeds, same order of draws)"""
    gen = torch.Generator().manual_seed(41)
    disp_true = torch.rand((1, h, w), generator=gen) * 50.0
    targets = (torch.randn((h, w), generator=gen),)
    x = torch.from_numpy(np.concatenate([synth.dsgn_normalised(51, h, w), synth.dsgn_normalised(52, h, w)]))
    extra = types.SimpleNamespace(calibs_fu=torch.tensor([721.5377]), calibs_baseline=torch.tensor([0.54]),
                                  calibs_Proj=torch.arange(12, dtype=torch.float64).view(1, 3, 4) / 10,
                                  calibs_Proj_R=torch.ones(1, 3, 4, dtype=torch.float64), disp_true=disp_true, targets=targets,
                                  calib=None, calib_R=None, ious=0.3, labels_map=None)
    return x, extra


@pytest.mark.parametrize("name", ["dsgn_both", "dsgn_depth_only", "dsgn_rpn_only"])
def test_dsgn_adapter_objective_equals_the_reference_statements(name, golden, golden_index):
    """a4: adapters.DsgnAdapter around the stand-in detector against what attack/DSGN/pgd_attack.py:269-270,301-336 - EXECUTED by
    make_golden.py around the same stand-in - left in ``loss`` and ``imgL.grad`` / ``imgR.grad``.  Same torch-CPU operators in the same
    order: compared bit for bit."""
    m, g = golden_index["objectives"][name], golden("objectives")
    x, extra = _dsgn_inputs(m["h"], m["w"])
    cfg = types.SimpleNamespace(PlaneSweepVolume=True, loss_disp=m["loss_disp"], RPN3D_ENABLE=m["RPN3D_ENABLE"], min_depth=2.0, max_depth=40.4,
                                stub_gain=0.7)
    model = stub_models.StubDsgn(5).eval()
    for freeze in (True, False):                   # freeze=False is the reference's behaviour; the image gradient does not depend on it
        loss, grad = adapters.DsgnAdapter(model, cfg, stub_models.StubRpn3dLoss, freeze=freeze).loss_and_grad(x, extra)
        assert not x.requires_grad and x.grad is None
        _same(loss, g[name + "_loss"], name + " loss")
        _same(grad[:1], g[name + "_gradL"], name + " left gradient")
        _same(grad[1:], g[name + "_gradR"], name + " right gradient")
    assert float(np.abs(g[name + "_gradL"]).sum()) > 0 and float(np.abs(g[name + "_gradR"]).sum()) > 0


def test_dsgn_adapter_freezes_the_detector_by_default():
    x, extra = _dsgn_inputs(10, 14)
    cfg = types.SimpleNamespace(PlaneSweepVolume=True, loss_disp=True, RPN3D_ENABLE=True, min_depth=2.0, max_depth=40.4, stub_gain=0.7)
    model = stub_models.StubDsgn(5).eval()
    adapters.DsgnAdapter(model, cfg, stub_models.StubRpn3dLoss).loss_and_grad(x, extra)
    assert all(p.grad is None for p in model.parameters())          # the detector's weights are constants: no weight gradients
    model2 = stub_models.StubDsgn(5).eval()
    adapters.DsgnAdapter(model2, cfg, stub_models.StubRpn3dLoss, freeze=False).loss_and_grad(x, extra)
    assert all(p.grad is not None for p in model2.parameters())     # the reference: model.zero_grad(), then backward into them too


def test_stereo_rcnn_adapter_objective_equals_the_reference_statements(golden, golden_index):
    """a14: adapters.StereoRcnnAdapter against attack/Stereo-RCNN/pgd_attack.py:153-174 executed around the same stand-in network:
    the fifteen outputs unpacked, six ``.mean() * exp(-u_k) + u_k`` terms in the script's order of additions, backward."""
    m, g = golden_index["objectives"]["srcnn"], golden("objectives")
    h, w = m["h"], m["w"]
    x = torch.from_numpy(np.concatenate([synth.srcnn_meansub(61, h, w), synth.srcnn_meansub(62, h, w)]))
    extra = types.SimpleNamespace(im_info=torch.tensor([[float(h), float(w), 1.6]]), gt_boxes_left=torch.full((1, 30, 5), 0.25),
                                  gt_boxes_right=torch.zeros(1, 30, 5), gt_boxes_merge=torch.zeros(1, 30, 5), gt_dim_orien=torch.zeros(1, 30, 5),
                                  gt_kpts=torch.zeros(1, 30, 6), num_boxes=torch.tensor([1]))
    u = torch.from_numpy(g["srcnn_uncert"].copy())
    loss, grad = adapters.StereoRcnnAdapter(stub_models.StubStereoRcnn(6).eval(), u).loss_and_grad(x, extra)
    _same(loss, g["srcnn_loss"], "srcnn loss")
    _same(grad[:1], g["srcnn_gradL"], "srcnn left gradient")
    _same(grad[1:], g["srcnn_gradR"], "srcnn right gradient")
    assert float(np.abs(g["srcnn_gradL"]).sum()) > 0 and float(np.abs(g["srcnn_gradR"]).sum()) > 0


def test_stereo_rcnn_adapter_graph_hooks_on_the_host():
    """what attacks.PgdAttack(graph=True) asks a Stereo R-CNN adapter: capturable only when the detector says so (the layer-list
    surrogates' static forward, opted in); a label signature only for device tensors of such a detector; clone / copy of the labels"""
    from eval_driving_safety_amd import surrogates
    model = surrogates.StereoRcnnR101(seed=0, rois_per_image=16, blocks=(1, 1, 1, 1)).eval()
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6))
    assert not net.graph_safe
    model.allow_graph_capture = True
    assert net.graph_safe
    model.static_shapes = False
    assert not net.graph_safe
    model.static_shapes = True
    extra = types.SimpleNamespace(im_info=torch.tensor([[600.0, 1987.0, 1.6]]), gt_boxes_left=torch.zeros(1, 30, 5), gt_boxes_right=torch.zeros(1, 30, 5),
                                  gt_boxes_merge=torch.zeros(1, 30, 5), gt_dim_orien=torch.zeros(1, 30, 5), gt_kpts=torch.zeros(1, 30, 6),
                                  num_boxes=torch.tensor([2]), note="kept by reference")
    assert net.graph_extra_signature(extra) is None                       # CPU labels: no capture to share
    assert model._host_values(extra.im_info, 2) == [600.0, 1987.0] and model._host_values(extra.num_boxes, 1) == [2.0]
    twin = net.graph_clone_extra(extra)
    assert twin.note == "kept by reference" and twin.gt_kpts is not extra.gt_kpts and torch.equal(twin.gt_kpts, extra.gt_kpts)
    extra.gt_boxes_left[0, 0] = torch.tensor([1.0, 2.0, 3.0, 4.0, 1.0])
    net.graph_copy_extra(twin, extra)
    assert torch.equal(twin.gt_boxes_left, extra.gt_boxes_left)
    assert not adapters.StereoRcnnAdapter(types.SimpleNamespace(), torch.zeros(6), freeze=False).graph_safe       # an upstream-like model says nothing: not capturable


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
