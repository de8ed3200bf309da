"""The Stereo R-CNN-shaped detector (surrogates.StereoRcnnShaped): the consumer of ops.RoIAlign (forward + deterministic
backward) and ops.nms inside the attack loop, checked against the same network with a plain-torch RoIAlign / NMS, and the
``--model shaped`` CLIs."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _roi_align_torch(feat, rois, pooled, scale):
    """legacy (aligned=False) RoIAlign with adaptive sampling, plain differentiable torch - the floating-point reference"""
    _, c, h, w = feat.shape
    outs = []
    for r in rois:
        b = int(r[0])
        sw, sh, ew, eh = r[1] * scale, r[2] * scale, r[3] * scale, r[4] * scale
        rw, rh = torch.clamp(ew - sw, min=1.0), torch.clamp(eh - sh, min=1.0)
        bw, bh = rw / pooled, rh / pooled
        gw, gh = int(torch.ceil(rw / pooled)), int(torch.ceil(rh / pooled))
        ys = sh + (torch.arange(pooled * gh, device=feat.device).float() + 0.5) * bh / gh
        xs = sw + (torch.arange(pooled * gw, device=feat.device).float() + 0.5) * bw / gw
        # the kernel computes start + ph*bin + (iy+.5)*bin/grid; the same value up to float32 rounding

        def axis(v, size):
            valid = ~((v < -1.0) | (v > size))
            v = v.clamp(min=0.0)
            lo = v.floor().long()
            top = lo >= size - 1
            lo = torch.where(top, torch.full_like(lo, size - 1), lo)
            hi = torch.where(top, lo, lo + 1)
            v = torch.where(top, lo.float(), v)
            frac = v - lo.float()
            return lo, hi, 1.0 - frac, frac, valid

        yl, yh, hy, ly, vy = axis(ys, h)
        xl, xh, hx, lx, vx = axis(xs, w)
        f = feat[b]
        val = (f[:, yl][:, :, xl] * (hy[:, None] * hx[None, :]) + f[:, yl][:, :, xh] * (hy[:, None] * lx[None, :]) +
               f[:, yh][:, :, xl] * (ly[:, None] * hx[None, :]) + f[:, yh][:, :, xh] * (ly[:, None] * lx[None, :]))
        val = val * (vy[:, None] & vx[None, :]).float()
        outs.append(val.view(c, pooled, gh, pooled, gw).mean(dim=(2, 4)))
    return torch.stack(outs)


def _nms_torch(boxes, scores, thresh):
    from oracle import oracle_np as O
    return torch.from_numpy(O.nms(boxes.cpu().numpy(), thresh)).to(boxes.device)


def _pair(dev, seed=0):
    from eval_driving_safety_amd import data, surrogates
    batch = next(iter(data.SyntheticStereo(1, "srcnn", 1, seed=seed)))
    extra = surrogates.synthetic_srcnn_extra(batch, dev)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    return batch, extra, x


def test_shaped_detector_runs_on_the_roi_kernels_and_matches_the_torch_formulation():
    from eval_driving_safety_amd import adapters, surrogates
    dev = torch.device("cuda", 0)
    _, extra, x = _pair(dev)
    model = surrogates.StereoRcnnShaped(seed=3).to(dev).eval()
    out = model(x[:1], x[1:], extra.im_info, extra.gt_boxes_left, extra.gt_boxes_right, extra.gt_boxes_merge, extra.gt_dim_orien,
                extra.gt_kpts, extra.num_boxes)
    assert len(out) == 15
    r = out[0].shape[1]
    assert out[0].shape == (1, r, 5) and out[2].shape == (1, r, 2) and out[3].shape == (1, r, 12) and out[4].shape == (1, r, 10)
    assert out[5].shape == (1, r, 4 * 28) and out[6].shape == (1, r, 28) and 2 <= r <= 65
    assert all(torch.isfinite(t).all() for t in out[8:14]) and int(out[14][0]) == 1          # the ground-truth roi is foreground
    u = torch.tensor([0.1, -0.2, 0.3, 0.0, 0.5, -0.4], device=dev)
    loss, grad = adapters.StereoRcnnAdapter(model, u).loss_and_grad(x.clone(), extra)
    assert float(grad[0].abs().sum()) > 0 and float(grad[1].abs().sum()) > 0
    # the same network with a plain-torch RoIAlign and the oracle's NMS: same proposals, loss and gradient within 1e-4
    ref = surrogates.StereoRcnnShaped(seed=3, roi_align=_roi_align_torch, nms=_nms_torch).to(dev).eval()
    ref.load_state_dict(model.state_dict())
    loss_r, grad_r = adapters.StereoRcnnAdapter(ref, u).loss_and_grad(x.clone(), extra)
    assert abs(float(loss) - float(loss_r)) <= 1e-4 * abs(float(loss_r))
    assert float((grad - grad_r).abs().max()) <= 1e-4 * float(grad_r.abs().max())
    # the gradient really flows through RoIAlign: with the RPN terms switched off it is still non-zero
    class NoRpn(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, *a):
            o = list(self.m(*a))
            o[8], o[9] = o[8] * 0, o[9] * 0
            return tuple(o)

    _, g2 = adapters.StereoRcnnAdapter(NoRpn(model), u).loss_and_grad(x.clone(), extra)
    assert float(g2.abs().sum()) > 0


def _run(mod, argv, cwd):
    out = subprocess.run([sys.executable, "-m", "eval_driving_safety_amd.cli." + mod] + argv, cwd=cwd, env=dict(os.environ, PYTHONPATH=ROOT),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    return out.stdout


def test_shaped_detectors_through_the_clis(tmp_path):
    from PIL import Image
    out = _run("srcnn_pgd_attack", ["--model", "shaped", "--synthetic", "1", "--iter", "2", "--eps", "0.03"], str(tmp_path))
    assert "attacked 1 stereo pairs" in out
    a = np.array(Image.open(str(tmp_path / "stereo_rcnn_pgd_iters_0" / "image_2" / "000000.png")))
    b = np.array(Image.open(str(tmp_path / "stereo_rcnn_pgd_iters_2" / "image_2" / "000000.png")))
    assert a.shape == (600, 1987, 3) and 1 <= np.abs(a.astype(int) - b.astype(int)).max() <= 3
    out = _run("srcnn_patch_attack", ["--model", "shaped", "--synthetic", "1", "--iter", "1", "--epochs", "1", "--pos_seed", "2"], str(tmp_path))
    p = np.load(str(tmp_path / "stereo_rcnn_patch_ratio_0.1" / "epoch1" / "patch.npy"))
    assert p.shape == (1, 3, 61, 61) and np.abs(p).max() > 0
    out = _run("dsgn_pgd_attack", ["--model", "shaped", "--synthetic", "1", "-btest", "1", "-d", "0", "--iter", "2", "--eps", "0.03"], str(tmp_path))
    a = np.array(Image.open(str(tmp_path / "dsgn_pgd_iters_0" / "image_3" / "000000.png")))
    b = np.array(Image.open(str(tmp_path / "dsgn_pgd_iters_2" / "image_3" / "000000.png")))
    assert a.shape == (375, 1242, 3) and 1 <= np.abs(a.astype(int) - b.astype(int)).max() <= 3


@pytest.mark.parametrize("reference_on_gpu", [False, True])
def test_device_import_gives_the_same_png_files(tmp_path, reference_on_gpu):
    """data.KittiFolder(as_u8=True) (8-bit upload, the loader transform on the GPU: ops.import_u8) through PgdAttack writes byte-identical
    attacked PNGs to the host-transform run - also with the reference-on-GPU step arithmetic (the import itself always uses the
    plain space), and with a frame larger than the network frame (clipped, and so is its reported size).  A library path: the CLI
    flag was removed in round 3 because the folder benchmark measured it slower (profiles/r02_folder_attack_device_import.jsonl)."""
    from PIL import Image
    import synth
    from eval_driving_safety_amd import adapters, attacks, data
    data_dir = tmp_path / "kitti"
    for eye in ("image_2", "image_3"):
        os.makedirs(str(data_dir / eye))
    frames = (("000001", (375, 1242)), ("000002", (370, 1224)), ("000003", (390, 1250)))
    for i, (name, (h, w)) in enumerate(frames):
        left = synth.u8_image(90 + i, h, w)
        Image.fromarray(left).save(str(data_dir / "image_2" / (name + ".png")))
        Image.fromarray(np.roll(left, -17, axis=1)).save(str(data_dir / "image_3" / (name + ".png")))
    (data_dir / "val.txt").write_text("".join(n + "\n" for n, _ in frames[:2]))
    (data_dir / "big.txt").write_text("000003\n")
    dev = torch.device("cuda", 0)
    for tag, as_u8 in (("host", False), ("dev", True)):
        atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 3, out_root=str(tmp_path / tag), device=dev, reference_on_gpu=reference_on_gpu)
        atk.run(data.KittiFolder(str(data_dir), str(data_dir / "val.txt"), 2, workers=2, as_u8=as_u8), adapters.ToyStereoAdapter(dev, seed=1))
    for k in (0, 3):
        for eye in ("image_2", "image_3"):
            for name in ("000001", "000002"):
                a = np.array(Image.open(str(tmp_path / "host" / ("dsgn_pgd_iters_%d" % k) / eye / (name + ".png"))))
                b = np.array(Image.open(str(tmp_path / "dev" / ("dsgn_pgd_iters_%d" % k) / eye / (name + ".png"))))
                assert a.shape == b.shape and np.array_equal(a, b), (k, eye, name)
    if not reference_on_gpu:
        atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 1, out_root=str(tmp_path / "big"), device=dev)
        atk.run(data.KittiFolder(str(data_dir), str(data_dir / "big.txt"), 1, as_u8=True), adapters.ToyStereoAdapter(dev, seed=1))
        assert np.array(Image.open(str(tmp_path / "big" / "dsgn_pgd_iters_1" / "image_2" / "000003.png"))).shape == (384, 1248, 3)


def test_own_pipeline_attack_detect_labels_scenario(tmp_path):
    """no upstream checkout anywhere: PNG folder -> dsgn_pgd_attack --model shaped -> dsgn_predict_and_save_pgd --model shaped on
    the clean and on the attacked folder -> KITTI label files -> the consumer's parse (evaluation/convert_scenarios.py rules)"""
    from PIL import Image
    from eval_driving_safety_amd import pixelio
    import synth
    data_dir = tmp_path / "kitti"
    for eye in ("image_2", "image_3"):
        os.makedirs(str(data_dir / eye))
    for i, name in enumerate(("000004", "000009")):
        left = synth.u8_image(50 + i, 375, 1242)
        Image.fromarray(left).save(str(data_dir / "image_2" / (name + ".png")))
        Image.fromarray(np.roll(left, -20, axis=1)).save(str(data_dir / "image_3" / (name + ".png")))
    (data_dir / "val.txt").write_text("000004\n000009\n")
    common = ["--model", "shaped", "--split_file", str(data_dir / "val.txt"), "-btest", "1", "-d", "0", "--out_root", str(tmp_path)]
    _run("dsgn_pgd_attack", common + ["--data_path", str(data_dir), "--iter", "3", "--eps", "0.03"], str(tmp_path))
    attacked = tmp_path / "dsgn_pgd_iters_3"
    assert sorted(os.listdir(str(attacked / "image_2"))) == ["000004.png", "000009.png"]
    _run("dsgn_predict_and_save_pgd", common + ["--data_path", str(data_dir), "--tag", "_clean"], str(tmp_path))
    _run("dsgn_predict_and_save_pgd", common + ["--data_path", str(attacked), "--iter", "3", "--alpha", "0.0039"], str(tmp_path))
    for folder in ("kitti_output_clean", "kitti_output_iter3_alpha0.0039"):
        files = sorted(os.listdir(str(tmp_path / folder)))
        assert files == ["000004.txt", "000009.txt"], (folder, files)
        label = pixelio.load_label(str(tmp_path / folder / "000004.txt"))
        assert len(label) >= 1 and all(item[0] == "Car" and 2.0 <= item[6][2] <= 41.0 for item in label)
        obstacles = pixelio.scenario_obstacles(label)
        assert len(obstacles) == len(label) and all(o["length"] > 3.0 and o["width"] > 1.0 for o in obstacles)


def test_r101_backbone_in_hip_graphs_gives_the_same_loss_and_gradient():
    """StereoRcnnR101.use_graph: the backbone + FPN forward and backward replayed from two hipGraphs (torch.cuda.make_graphed_callables)
    - the same kernels in the same order as the eager step, so the same loss and image gradient"""
    from eval_driving_safety_amd import adapters, surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=2, rois_per_image=32, blocks=(1, 1, 2, 1)).to(dev).eval()
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn((2, 3, 160, 320), generator=gen) * 40).to(dev)
    left = torch.zeros((1, 30, 5), device=dev)
    left[:, 0] = torch.tensor([100.0, 50.0, 220.0, 120.0, 1.0], device=dev)
    right = left.clone()
    right[:, 0, 0] -= 10
    right[:, 0, 2] -= 10
    do = torch.zeros((1, 30, 5), device=dev)
    kp = torch.zeros((1, 30, 6), device=dev)
    kp[:, 0] = torch.tensor([150.0, 1, 0, 110, 210, 0], device=dev)
    import types
    extra = types.SimpleNamespace(im_info=torch.tensor([[160.0, 320.0, 1.0]], device=dev), gt_boxes_left=left, gt_boxes_right=right,
                                  gt_boxes_merge=left.clone(), gt_dim_orien=do, gt_kpts=kp, num_boxes=torch.tensor([1], device=dev))
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
    net.loss_and_grad(x.clone(), extra)                       # warm-up: solver searches
    loss_e, grad_e = net.loss_and_grad(x.clone(), extra)
    model.use_graph = True
    net.loss_and_grad(x.clone(), extra)                       # captures
    loss_g, grad_g = net.loss_and_grad(x.clone(), extra)
    assert float(loss_g) == float(loss_e) or abs(float(loss_g) - float(loss_e)) <= 1e-5 * abs(float(loss_e))
    assert float((grad_g - grad_e).abs().max()) <= 1e-4 * float(grad_e.abs().max())


def test_r101_step_on_libadvengine_convolutions_matches_miopen():
    """FoldedConv.impl = "auto" (ops.Conv2dAuto: libadvengine's 1x1 / 3x3 kernels where they measure faster, the ReLU masks of each
    bottleneck's conv1 / conv2 left to their consumer's dgrad epilogue) and "hip" (always libadvengine) against "miopen" (torch's
    operators throughout): the same loss and image gradient within float32 summation-order differences"""
    import types
    from eval_driving_safety_amd import adapters, surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=5, rois_per_image=32, blocks=(2, 1, 2, 1)).to(dev).eval()
    gen = torch.Generator().manual_seed(6)
    x = (torch.randn((2, 3, 192, 352), generator=gen) * 40).to(dev)
    left = torch.zeros((1, 30, 5), device=dev)
    left[:, 0] = torch.tensor([100.0, 50.0, 240.0, 140.0, 1.0], device=dev)
    right = left.clone()
    right[:, 0, 0] -= 10
    right[:, 0, 2] -= 10
    kp = torch.zeros((1, 30, 6), device=dev)
    kp[:, 0] = torch.tensor([150.0, 1, 0, 110, 230, 0], device=dev)
    extra = types.SimpleNamespace(im_info=torch.tensor([[192.0, 352.0, 1.0]], device=dev), gt_boxes_left=left, gt_boxes_right=right,
                                  gt_boxes_merge=left.clone(), gt_dim_orien=torch.zeros((1, 30, 5), device=dev), gt_kpts=kp,
                                  num_boxes=torch.tensor([1], device=dev))
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
    out = {}
    try:
        for impl in ("miopen", "hip", "auto"):
            surrogates.FoldedConv.impl = impl
            net.loss_and_grad(x.clone(), extra)
            out[impl] = net.loss_and_grad(x.clone(), extra)
    finally:
        surrogates.FoldedConv.impl = "miopen"
    ref_loss, ref_grad = out["miopen"]
    for impl in ("hip", "auto"):
        loss, grad = out[impl]
        assert abs(float(loss) - float(ref_loss)) <= 2e-4 * abs(float(ref_loss)), impl
        scale = float(ref_grad.abs().max())
        assert float((grad - ref_grad).abs().max()) <= 2e-2 * scale, impl          # float32 through ~40 layers, different summation orders
        big = ref_grad.abs() > 2e-2 * scale                                        # what the PGD step consumes: the signs
        assert float((torch.sign(grad[big]) == torch.sign(ref_grad[big])).float().mean()) > 0.995, impl


def test_r101_merged_rpn_heads_give_the_separate_layers_outputs():
    """rpn_heads on the Conv2dAuto path runs the class and the regression layer as one 1x1 layer to 3 + 18 channels: the forward's bits
    are those of the two layers; the gradient w.r.t. the map agrees to float32 rounding (21 products summed in one chain instead of 3 + 18);
    a changed weight (checkpoint load) is picked up"""
    from eval_driving_safety_amd import surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=2, rois_per_image=32, blocks=(1, 1, 1, 1)).to(dev).eval()
    gen = torch.Generator().manual_seed(1)
    both = torch.relu(torch.randn((1, 1024, 38, 125), generator=gen)).to(dev).requires_grad_(True)
    try:
        surrogates.FoldedConv.impl = "auto"
        for attempt in range(2):
            s, d = model.rpn_heads(both)
            s2, d2 = model.rpn_scores(both), model.rpn_deltas(both)
            assert torch.equal(s, s2) and torch.equal(d, d2)
            gs, gd = torch.randn(s.shape, generator=gen).to(dev), torch.randn(d.shape, generator=gen).to(dev)
            g1, = torch.autograd.grad((s * gs).sum() + (d * gd).sum(), both)
            g2, = torch.autograd.grad((s2 * gs).sum() + (d2 * gd).sum(), both)
            assert float((g1 - g2).abs().max()) <= 1e-5 * float(g2.abs().max())
            with torch.no_grad():
                model.rpn_reg.weight.mul_(1.5)                  # the merged copy follows (the layer's own prepared weights are dropped by hand,
            model.rpn_reg._prep = None                          # as checkpoints.load_stereo_rcnn does)
    finally:
        surrogates.FoldedConv.impl = "miopen"


def test_r101_step_is_reproducible_bit_for_bit():
    """the R101 layer-list step twice from the same input: the same bytes.  What makes that hold: RoIAlign's backward is this
    package's ordered one (no atomics), the FPN / keypoint bilinear up-samplings go through adapters._BilinearUp (torch's own backward
    scatters with atomicAdd), the level gather is a permutation (one addend per element), and MIOpen is asked for deterministic
    solvers.  torch.use_deterministic_algorithms(True) makes torch raise on any operator it knows to be order-dependent."""
    import types
    from eval_driving_safety_amd import adapters, surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=8, rois_per_image=64, blocks=(1, 1, 2, 1)).to(dev).eval()
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn((2, 3, 192, 352), generator=gen) * 40).to(dev)
    left = torch.zeros((1, 30, 5), device=dev)
    left[:, 0] = torch.tensor([100.0, 50.0, 240.0, 140.0, 1.0], device=dev)
    right = left.clone()
    right[:, 0, 0] -= 10
    right[:, 0, 2] -= 10
    kp = torch.zeros((1, 30, 6), device=dev)
    kp[:, 0] = torch.tensor([150.0, 1, 0, 110, 230, 0], device=dev)
    extra = types.SimpleNamespace(im_info=torch.tensor([[192.0, 352.0, 1.0]], device=dev), gt_boxes_left=left, gt_boxes_right=right,
                                  gt_boxes_merge=left.clone(), gt_dim_orien=torch.zeros((1, 30, 5), device=dev), gt_kpts=kp,
                                  num_boxes=torch.tensor([1], device=dev))
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
    old = os.environ.get("CUBLAS_WORKSPACE_CONFIG")
    os.environ.setdefault("CUBLAS_WORKSPACE_CONFIG", ":4096:8")
    try:
        surrogates.FoldedConv.impl = "auto"
        torch.use_deterministic_algorithms(True)
        net.loss_and_grad(x.clone(), extra)                   # MIOpen answers a shape's first call with a fallback solver
        runs = [net.loss_and_grad(x.clone(), extra) for _ in range(3)]
    finally:
        torch.use_deterministic_algorithms(False)
        surrogates.FoldedConv.impl = "miopen"
        if old is None:
            os.environ.pop("CUBLAS_WORKSPACE_CONFIG", None)
    for loss, grad in runs[1:]:
        assert float(loss) == float(runs[0][0])
        assert torch.equal(grad, runs[0][1])
    assert float(runs[0][1].abs().max()) > 0


def test_static_forward_equals_the_compacting_forward():
    """StereoRcnnShaped._forward_static (no host read-back: masks and padded index lists where forward() compacts) against forward() with
    ``static_shapes = False``: the same rois and labels to the bit, the same outputs / loss terms / image gradient up to the float32
    summation order of the masked sums - with ground truth, with two boxes, and without any"""
    import types
    from eval_driving_safety_amd import adapters, surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=3, rois_per_image=96, blocks=(1, 1, 2, 1)).to(dev).eval()
    gen = torch.Generator().manual_seed(12)
    x = (torch.randn((2, 3, 192, 352), generator=gen) * 40).to(dev)
    for n_gt in (1, 2, 0):
        left = torch.zeros((1, 30, 5), device=dev)
        left[:, 0] = torch.tensor([100.0, 50.0, 240.0, 140.0, 1.0], device=dev)
        left[:, 1] = torch.tensor([20.0, 90.0, 90.0, 150.0, 1.0], device=dev)
        right = left.clone()
        right[:, :2, 0] -= 10
        right[:, :2, 2] -= 10
        kp = torch.zeros((1, 30, 6), device=dev)
        kp[:, 0] = torch.tensor([150.0, 1, 0, 110, 230, 0], device=dev)
        kp[:, 1] = torch.tensor([40.0, 1, 0, 25, 85, 0], device=dev)
        do = torch.randn((1, 30, 5), generator=gen).to(dev)
        extra = types.SimpleNamespace(im_info=torch.tensor([[192.0, 352.0, 1.0]], device=dev), gt_boxes_left=left, gt_boxes_right=right,
                                      gt_boxes_merge=left.clone(), gt_dim_orien=do, gt_kpts=kp, num_boxes=torch.tensor([n_gt], device=dev))
        net = adapters.StereoRcnnAdapter(model, torch.tensor([0.1, -0.2, 0.3, 0.0, 0.2, -0.1], device=dev))
        out = {}
        # (MIOpen computes the strided layers: deterministic solvers and a warm-up call, or the two forwards differ in their last bits by themselves)
        net.loss_and_grad(x.clone(), extra)
        for static in (False, True):
            model.static_shapes = static
            assert model._static_ok(x) == static
            with torch.no_grad():
                res = model(x[:1], x[1:], extra.im_info, left, right, left, do, kp, extra.num_boxes)
            out[static] = (res, net.loss_and_grad(x.clone(), extra))
        model.static_shapes = True
        (ra, (la, ga)), (rb, (lb, gb)) = out[False], out[True]
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[14], rb[14]), n_gt      # rois left / right, labels
        for k in range(2, 14):
            scale = max(float(ra[k].abs().max()), 1e-6)
            assert float((ra[k] - rb[k]).abs().max()) <= 2e-5 * scale, (n_gt, k)
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), n_gt
        assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max()), n_gt


def test_host_value_cache_follows_the_tensor():
    """StereoRcnnShaped._host_values remembers im_info / num_boxes per tensor object (no read-back per step) - and must notice an in-place
    write (version) and a ``.data`` swap (pointer), as patch_attack.py:188 does with num_boxes"""
    from eval_driving_safety_amd import surrogates
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=1, rois_per_image=32, blocks=(1, 1, 1, 1)).to(dev).eval()
    nb = torch.tensor([3], device=dev)
    assert model._host_values(nb, 1) == [3.0] and model._host_values(nb, 1) is model._host_values(nb, 1)      # cached list object
    nb.fill_(2)
    assert model._host_values(nb, 1) == [2.0]
    nb.data = torch.tensor([5], device=dev)
    assert model._host_values(nb, 1) == [5.0]
    nb.data = torch.tensor(1)                                   # the reference's own statement: a CPU scalar from then on
    assert model._host_values(nb, 1) == [1.0]
    info = torch.tensor([[600.0, 1987.0, 1.0]], device=dev)
    assert model._host_values(info, 2) == [600.0, 1987.0]


_GRAPH_WORKER = r"""
import sys, torch
sys.path.insert(0, %r)
from eval_driving_safety_amd import adapters, attacks, data, surrogates
dev = torch.device("cuda", 0)
model = surrogates.StereoRcnnR101(seed=4, rois_per_image=64, blocks=(1, 1, 2, 1)).to(dev).eval()
net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
assert not net.graph_safe                # opt-in
model.allow_graph_capture = True
assert net.graph_safe
batch = next(iter(data.SyntheticStereo(1, "srcnn", batch=1, seed=3)))
batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
surrogates.FoldedConv.impl = "auto"
eager = attacks.PgdAttack("srcnn", 1.0, 0.03, 4, save=False, device=dev)
xe = eager.run_batch(batch, net).clone()
le = [float(v) for v in eager.last_losses]
atk = attacks.PgdAttack("srcnn", 1.0, 0.03, 4, save=False, device=dev, graph=True)
outs = []
for _ in range(3):                      # capture, then the capture reused twice
    outs.append((atk.run_batch(batch, net).clone(), [float(v) for v in atk.last_losses]))
# another label set with the same host constants (one box, same image size) shares the capture: its tensors are copied into the
# captured ones; a label set with two boxes needs a capture of its own
import copy
b2 = copy.copy(batch)
b2.extra = surrogates.synthetic_srcnn_extra(batch, dev)
b2.extra.gt_boxes_left[:, 0, :4] += torch.tensor([-60.0, 20.0, -40.0, 30.0], device=dev)
b2.extra.gt_boxes_right[:, 0, :4] += torch.tensor([-60.0, 20.0, -40.0, 30.0], device=dev)
b2.extra.gt_boxes_merge.copy_(b2.extra.gt_boxes_left)
keep_first = batch.extra.gt_boxes_left.clone()
x2e, x2g = eager.run_batch(b2, net).clone(), atk.run_batch(b2, net).clone()
reused_after_b2 = atk.graph_captures_reused
b3 = copy.copy(batch)
b3.extra = surrogates.synthetic_srcnn_extra(batch, dev)
b3.extra.gt_boxes_left[:, 1] = torch.tensor([200.0, 280.0, 420.0, 400.0, 1.0], device=dev)
b3.extra.gt_boxes_right[:, 1] = torch.tensor([170.0, 280.0, 390.0, 400.0, 1.0], device=dev)
b3.extra.gt_boxes_merge.copy_(b3.extra.gt_boxes_left)
b3.extra.num_boxes.fill_(2)
x3e, x3g = eager.run_batch(b3, net).clone(), atk.run_batch(b3, net).clone()
torch.cuda.synchronize()
assert reused_after_b2 == 3 and atk.graph_captures_reused == 3 and len(atk._graph_caches) == 2
assert torch.equal(x2e, x2g) and not torch.equal(x2e, xe) and torch.equal(x3e, x3g)
assert torch.equal(batch.extra.gt_boxes_left, keep_first)          # the caller's labels were not written to
for xg, lg in outs:
    assert lg == le and torch.equal(xg, xe), (lg, le)
assert le[-1] != le[0]
model.static_shapes = False
assert not net.graph_safe
print("GRAPH-REUSE-OK")
"""


def test_static_iteration_replays_from_a_hipgraph_and_the_capture_is_reused(tmp_path):
    """one whole attack iteration through the R101-shaped detector (forward, losses, backward, fused PGD step) captured once, replayed, and
    reused for two more batches (attacks.PgdAttack(graph=True)): the eager loop's perturbed pair and losses, byte for byte.  Runs in a
    child process: a fault inside a replayed hipGraph aborts the process that launched it, and this torch / ROCm stack has produced such
    faults on the way here (surrogates.StereoRcnnShaped.graph_capturable).  A child killed by a signal FAILS the test - a fault could as
    well be an out-of-bounds access of one of this package's kernels inside the capture - unless ADV_ALLOW_GRAPH_FAULT=1 says the machine is
    known to fault on replays; a child that finishes must have found everything equal."""
    script = tmp_path / "graph_worker.py"
    script.write_text(_GRAPH_WORKER % ROOT)
    res = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if (res.returncode < 0 or "hardware exception" in res.stderr) and os.environ.get("ADV_ALLOW_GRAPH_FAULT") == "1":
        pytest.xfail("the hipGraph replay faulted on this machine (signal %d): %s" % (res.returncode, res.stderr[-200:]))
    assert res.returncode == 0 and "GRAPH-REUSE-OK" in res.stdout, res.stderr[-2000:]


def test_cli_layerlist_models_and_graph_flag(tmp_path):
    """`--model layerlist`: the attack CLIs on the random-weight networks with the upstream layer lists (what bench.py's end-to-end legs
    measure) - DSGN with one PGD iteration captured in a hipGraph (`--graph`), Stereo R-CNN's ResNet-101-FPN eagerly"""
    from PIL import Image
    out = _run("dsgn_pgd_attack", ["--model", "layerlist", "--graph", "--synthetic", "1", "-btest", "1", "-d", "0", "--iter", "2", "--eps", "0.03"], str(tmp_path))
    assert "rank 0 attacked 1 stereo pairs" in out
    a = np.array(Image.open(str(tmp_path / "dsgn_pgd_iters_0" / "image_2" / "000000.png")))
    b = np.array(Image.open(str(tmp_path / "dsgn_pgd_iters_2" / "image_2" / "000000.png")))
    assert a.shape == (375, 1242, 3) and 1 <= np.abs(a.astype(int) - b.astype(int)).max() <= 3
    out = _run("srcnn_pgd_attack", ["--model", "layerlist", "--synthetic", "1", "-d", "0", "--iter", "1", "--eps", "0.03"], str(tmp_path))
    assert "rank 0 attacked 1 stereo pairs" in out
    assert np.array(Image.open(str(tmp_path / "stereo_rcnn_pgd_iters_1" / "image_3" / "000000.png"))).shape == (600, 1987, 3)
