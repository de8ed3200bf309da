"""The attack drivers on the real HIP path (toy detector on the GPU supplies gradients; the oracle
replays the same gradients on the host).  Bit-exact."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import synth
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Recorder:
    """wraps an adapter and keeps the gradients it returned"""

    def __init__(self, inner):
        self.inner, self.grads = inner, []

    def loss_and_grad(self, x, extra=None):
        loss, g = self.inner.loss_and_grad(x, extra)
        self.grads.append(g.detach().cpu().numpy().copy())
        return loss, g

    def inject_fake_target(self, *a):
        pass


def test_pgd_driver_batch_of_pairs_on_hip(tmp_path):
    from PIL import Image
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    batch = next(iter(data.SyntheticStereo(3, "dsgn", batch=3, seed=2)))
    rec = _Recorder(adapters.ToyStereoAdapter(dev, seed=1))
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 4, out_root=str(tmp_path), device=dev)
    x = atk.run_batch(batch, rec)
    atk.close()
    xm = torch.cat([batch.imgL, batch.imgR]).numpy().copy()
    clean = O.denormalize(xm)
    for k in range(4):
        xm = O.pgd_step_norm01(xm, rec.grads[k], clean, 1 / 255, 0.03)
    assert x.cpu().numpy().tobytes() == xm.tobytes()
    # the loader's zero-padded 375x1242 images take the 8-bit-index path, every image of the batch
    assert atk.last_clean_index is not None and atk.last_clean_index.verified() == [True] * 6
    got = np.array(Image.open(os.path.join(str(tmp_path), "dsgn_pgd_iters_4", "image_3", "000001.png")).convert("RGB"))
    assert got.shape == (375, 1242, 3)
    assert np.array_equal(got, O.tensor2im_u8(xm[3 + 1], 375, 1242))
    got0 = np.array(Image.open(os.path.join(str(tmp_path), "dsgn_pgd_iters_0", "image_2", "000000.png")).convert("RGB"))
    assert np.array_equal(got0, O.tensor2im_u8(batch.imgL[0].numpy(), 375, 1242))


def test_pgd_driver_one_foreign_image_keeps_the_rest_on_the_index_path(tmp_path):
    """a pair whose left image has no 8-bit origin (here: re-scaled after loading) falls back to the float32 clean
    image on its own; the other images of the same launches stay indexed; everything equals the oracle"""
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    batch = next(iter(data.SyntheticStereo(3, "dsgn", batch=3, seed=4)))
    batch.imgL[1, :, :375, :1242] *= 0.999
    rec = _Recorder(adapters.ToyStereoAdapter(dev, seed=1))
    for in_place in (False, True):
        rec.grads = []
        atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 3, save=False, device=dev, in_place=in_place)
        x = atk.run_batch(batch, rec)
        assert atk.last_clean_index.verified() == [True, False, True, True, True, True]
        xm = torch.cat([batch.imgL, batch.imgR]).numpy().copy()
        clean = O.denormalize(xm)
        for k in range(3):
            xm = O.pgd_step_norm01(xm, rec.grads[k], clean, 1 / 255, 0.03)
        assert x.cpu().numpy().tobytes() == xm.tobytes()


@pytest.mark.parametrize("kind", ["dsgn", "srcnn"])
def test_patch_trainer_reference_sequence_on_hip(kind, tmp_path):
    from eval_driving_safety_amd import adapters, attacks, data, patchgeom
    dev = torch.device("cuda", 0)
    H, W = patchgeom.DSGN_SHAPE if kind == "dsgn" else patchgeom.SRCNN_SHAPE
    batches = list(data.SyntheticStereo(2, kind, batch=1, seed=3))
    rec = _Recorder(adapters.ToyStereoAdapter(dev, seed=2))
    eps = 8 / 255 if kind == "dsgn" else 0.1
    ratio = 0.2 if kind == "dsgn" else 0.1
    tr = attacks.PatchTrainer(kind, ratio, eps, 2, 1, out_root=str(tmp_path), seed=11, device=dev)
    tr.ALPHA = 1e3
    patch = tr.train(lambda: batches, rec)
    rng = random.Random(11)
    D, r = O.init_patch_dims(H if kind == "dsgn" else 600, ratio)
    p = np.zeros((1, 3, D, D), np.float32)
    lo, hi = (None, None) if kind == "dsgn" else (O.SRCNN_LO, O.SRCNN_HI)
    gi = 0
    for b in batches:
        cl, cr = O.round_mask_centers(rng, H, W, r)
        gacc = None
        for it in range(2):
            g = rec.grads[gi]
            gi += 1
            gacc = g if gacc is None else gacc + g
            p = O.patch_update(p, gacc[0:1], gacc[1:2], cl[0], cl[1], cr[1], r, eps, lo=lo, hi=hi)
    assert patch.cpu().numpy().tobytes() == p.tobytes()
    assert np.abs(p).max() > 0


def test_patch_trainer_batched_rule_on_hip(tmp_path):
    """B = 3 pairs per round against one snapshot: summed deltas, applied once"""
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    batch = next(iter(data.SyntheticStereo(3, "dsgn", batch=3, seed=5)))
    rec = _Recorder(adapters.ToyStereoAdapter(dev, seed=3))
    tr = attacks.PatchTrainer("dsgn", 0.2, 8 / 255, 2, 1, out_root=str(tmp_path), seed=4, device=dev)
    patch = tr.train(lambda: [batch], rec)
    rng = random.Random(4)
    D, r = O.init_patch_dims(384, 0.2)
    cs = [O.round_mask_centers(rng, 384, 1248, r) for _ in range(3)]
    p = np.zeros((1, 3, D, D), np.float32)
    gacc = None
    for it in range(2):
        gacc = rec.grads[it] if gacc is None else gacc + rec.grads[it]
        total = None
        for i, (cl, cr) in enumerate(cs):
            d = O.patch_delta(gacc[i:i + 1], gacc[3 + i:4 + i], cl[0], cl[1], cr[1], r, 8 / 255)
            total = d if total is None else total + d
        p = O.patch_apply_delta(p, total)
    assert patch.cpu().numpy().tobytes() == p.tobytes()


def test_cli_toy_run_writes_the_reference_layout(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "eval_driving_safety_amd.cli.dsgn_pgd_attack", "--model", "toy", "--synthetic", "2",
           "-btest", "1", "-d", "0", "--iter", "2", "--eps", "0.03", "--out_root", str(tmp_path)]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    for k in range(3):
        for folder in ("image_2", "image_3"):
            assert sorted(os.listdir(os.path.join(str(tmp_path), "dsgn_pgd_iters_%d" % k, folder))) == ["000000.png", "000001.png"]
    cmd = [sys.executable, "-m", "eval_driving_safety_amd.cli.srcnn_patch_attack", "--model", "toy", "--synthetic", "2",
           "--iter", "1", "--epochs", "1", "--out_root", str(tmp_path), "--pos_seed", "1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    p = np.load(os.path.join(str(tmp_path), "stereo_rcnn_patch_ratio_0.1", "epoch1", "patch.npy"))
    assert p.shape == (1, 3, 61, 61) and p.dtype == np.float32 and np.abs(p).max() > 0
    assert "Average loss for epoch1" in out.stdout


@pytest.mark.parametrize("script,prefix", [("dsgn_pgd_attack", "dsgn_pgd_iters_"), ("srcnn_pgd_attack", "stereo_rcnn_pgd_iters_")])      # (the second: 36 s - the graph that leaves more layers to MIOpen)
def test_a_cli_run_twice_in_fresh_processes_writes_the_same_png_bytes(tmp_path, script, prefix):
    """README "Reproducibility", for the product path: the command-line script, run twice in two fresh processes on the layer-list detector
    (route table + the layers left to MIOpen under determinism.py's flags and warm-up, set by the drivers - nothing is set here), writes
    byte-identical attacked images for every iterate.  The first process may populate MIOpen's user caches for the second: the bytes must
    not depend on that."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    digests = []
    for run in ("a", "b"):
        root = tmp_path / run
        cmd = [sys.executable, "-m", "eval_driving_safety_amd.cli." + script, "--model", "layerlist", "--synthetic", "1", "--iter", "3", "--eps", "0.03",
               "--out_root", str(root)] + (["-btest", "1", "-d", "0"] if script.startswith("dsgn") else [])
        out = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:]
        files = {}
        for k in range(4):
            for eye in ("image_2", "image_3"):
                d = os.path.join(str(root), prefix + str(k), eye)
                for name in sorted(os.listdir(d)):
                    files[(k, eye, name)] = open(os.path.join(d, name), "rb").read()
        assert len(files) == 8
        digests.append(files)
    assert digests[0].keys() == digests[1].keys()
    differ = [k for k in digests[0] if digests[0][k] != digests[1][k]]
    assert not differ, differ
    assert digests[0][(0, "image_2", sorted(n for (k, e, n) in digests[0] if k == 0)[0])] != digests[0][(3, "image_2", sorted(n for (k, e, n) in digests[0] if k == 3)[0])]


def test_attack_folders_feed_detect_under_attack(tmp_path):
    """the file contract end to end: PGD folders -> swapped in as image_2/image_3 (attack/DSGN/README.md:30,69) ->
    detect-under-attack -> KITTI label files -> the consumer's parser (evaluation/convert_scenarios.py:52-95)"""
    from PIL import Image
    from eval_driving_safety_amd import adapters, attacks, data, pixelio
    dev = torch.device("cuda", 0)
    src = data.SyntheticStereo(2, "dsgn", batch=2, seed=9, first_index=40)
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 2, out_root=str(tmp_path), device=dev)
    final = None
    for batch in src:
        final = atk.run_batch(batch, adapters.ToyStereoAdapter(dev, seed=5)).cpu().numpy()
    atk.close()
    attacked = os.path.join(str(tmp_path), "dsgn_pgd_iters_2")
    (tmp_path / "val.txt").write_text("000040\n000041\n")
    seen = []

    class Det:
        def detect(self, x, extra):
            seen.append(x.cpu().numpy())
            return [[(2, np.float32([100, 120, 180, 190]), np.float32(0.8), np.float32([2.0, 1.6, 25.0]), (1.5, 1.7, 4.1, 4.0))]
                    for _ in range(x.shape[0] // 2)]

    label_dir = pixelio.dsgn_label_dir(str(tmp_path / "ckpt" / "finetune_53.tar"), pixelio.dsgn_tag("", 2, 1 / 255))
    dua = attacks.DetectUnderAttack("dsgn", "pgd", label_dir, device=dev)
    assert dua.run(data.KittiFolder(attacked, str(tmp_path / "val.txt"), batch=2), Det()) == 2
    # what the detector saw is the 8-bit file content, re-normalised by the loader
    png = np.array(Image.open(os.path.join(attacked, "image_3", "000041.png")).convert("RGB"))
    assert np.array_equal(png, O.tensor2im_u8(final[3], 375, 1242))
    redo = data.dsgn_transform(torch.from_numpy(np.ascontiguousarray(png.transpose(2, 0, 1)))).numpy()
    assert np.array_equal(seen[0][3], redo)
    assert np.abs(seen[0][3][:, :375, :1242] - final[3][:, :375, :1242]).max() < 1.2 / 255 / 0.224   # one 8-bit quantum
    label = pixelio.load_label(os.path.join(label_dir, "000041.txt"))
    assert label[0][0] == "Car" and label[0][5] == [1.5, 1.7, 4.1]
    obs = pixelio.scenario_obstacles(label)
    assert len(obs) == 1 and obs[0]["position"] == [25.0, -2.0] and abs(obs[0]["orientation"] - (0.5 * np.pi - (4.0 - 2 * np.pi))) < 1e-12
    assert label_dir.endswith("kitti_output_iter2_alpha%s" % str(1 / 255))


def test_baseline_config1_fgsm_four_pairs(tmp_path):
    """BASELINE.json configs[0]: 1-step FGSM, eps = alpha = 8/255, DSGN pixel space, 4 KITTI-shaped pairs - the HIP
    path against the CPU oracle fed with the same gradients; every pixel moved by exactly +-8/255 or hit the range"""
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    batch = next(iter(data.SyntheticStereo(4, "dsgn", batch=4, seed=21)))
    rec = _Recorder(adapters.ToyStereoAdapter(dev, seed=8))
    atk = attacks.PgdAttack("dsgn", 8 / 255, 8 / 255, 1, out_root=str(tmp_path), device=dev)
    x = atk.run_batch(batch, rec).cpu().numpy()
    atk.close()
    x0 = torch.cat([batch.imgL, batch.imgR]).numpy()
    clean = O.denormalize(x0)
    want = O.pgd_step_norm01(x0, rec.grads[0], clean, 8 / 255, 8 / 255)
    assert x.tobytes() == want.tobytes()
    moved = np.abs(O.denormalize(x) - clean)
    g = rec.grads[0]
    inner = (clean > 8 / 255 + 1e-3) & (clean < 1 - 8 / 255 - 1e-3) & (g != 0)
    assert np.allclose(moved[inner], 8 / 255, atol=2e-7)
    assert sorted(os.listdir(str(tmp_path))) == ["dsgn_pgd_iters_0", "dsgn_pgd_iters_1"]
    assert len(os.listdir(os.path.join(str(tmp_path), "dsgn_pgd_iters_1", "image_2"))) == 4


@pytest.mark.slow        # 60 s of MIOpen compilation on a fresh box; tests/test_surrogates.py compares the same two routes on the R101 step
def test_surrogate_detector_gradients_agree_between_mfma_and_miopen_convs():
    """the DSGN-shaped surrogate with its 3D convolutions on libadvengine's MFMA kernel vs on torch / MIOpen: same
    loss and image gradient up to float32 summation order; and a 3-step attack raises the loss"""
    import types
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    batch = next(iter(data.SyntheticStereo(1, "dsgn", batch=1, seed=4)))
    gen = torch.Generator().manual_seed(2)
    gt = torch.rand((1, 384, 1248), generator=gen) * 38.4 + 2.0
    gt = torch.where(torch.rand((1, 384, 1248), generator=gen) < 0.05, gt, torch.zeros(()))
    extra = types.SimpleNamespace(disp_true=gt.to(dev))
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    la, ga = adapters.PsvStereoAdapter(dev, seed=0, mfma_conv=True).loss_and_grad(x.clone(), extra)
    lb, gb = adapters.PsvStereoAdapter(dev, seed=0, mfma_conv=False).loss_and_grad(x.clone(), extra)
    assert abs(float(la) - float(lb)) <= 1e-4 * abs(float(lb))
    # ReLU gates sitting at ~0 flip between the two float32 summation orders, so compare in the L2 sense
    rel = float((ga - gb).norm() / gb.norm())
    assert rel <= 2e-2, rel
    assert float(ga[1].abs().sum()) > 0                      # the right eye receives gradient through the cost volume
    batch.extra = extra
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 3, save=False, device=dev)
    atk.run_batch(batch, adapters.PsvStereoAdapter(dev, seed=0))
    assert float(atk.last_losses[-1]) > float(atk.last_losses[0])


@pytest.mark.parametrize("detector", ["toy", "dsgn_shaped", "dsgn_layer_list"])
def test_pgd_iteration_captured_in_a_hip_graph_equals_the_eager_loop(tmp_path, detector):
    """PgdAttack(graph=True): ONE iteration - detector forward + loss + backward + adv_pgd_step_indexed_f32 with its 8-bit export -
    captured with torch's stream capture and replayed N times, against the eager loop: the same final iterate, the same losses, the
    same PNG files, bit for bit.  The DSGN-shaped detectors put every convolution / cost-volume / depth-regression / grid-sample /
    focal-loss entry point of libadvengine (forward and backward) inside the capture."""
    from eval_driving_safety_amd import adapters, attacks, data
    dev = torch.device("cuda", 0)
    if detector == "toy":
        batch = next(iter(data.SyntheticStereo(2, "dsgn", batch=2, seed=5)))
        make = lambda: adapters.ToyStereoAdapter(dev, seed=1)                                  # noqa: E731
    else:
        hw = (96, 160)
        gen = torch.Generator().manual_seed(11)
        left = torch.randn((1, 3) + hw, generator=gen)
        batch = data.StereoBatch(left, torch.roll(left, shifts=-6, dims=3) + 0.05 * torch.randn((1, 3) + hw, generator=gen), ["000000"], None)
        kw = dict(seed=3, image_hw=hw, cu=80.0, cv=44.0, fu=180.0)
        if detector == "dsgn_shaped":
            make = lambda: adapters.PsvStereoAdapter(dev, hourglass=True, dsgn_head=True, **kw)  # noqa: E731
        else:
            make = lambda: adapters.DsgnShapedAdapter(dev, **kw)                                # noqa: E731
        batch.extra = make().synthetic_extra(batch, seed=2)
    n = 5
    outs = {}
    # (MIOpen may pick split-K solvers that accumulate with atomics for some 2D shapes: two runs of the same eager loop then differ in
    #  the last bits - ask it for deterministic solvers, so that what is compared is the capture, not MIOpen's schedule)
    for mode in ("eager", "graph"):
        root = tmp_path / mode
        atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, n, out_root=str(root), device=dev, graph=(mode == "graph"), save=(detector == "toy"))
        x = atk.run_batch(data.StereoBatch(batch.imgL.clone(), batch.imgR.clone(), batch.names, batch.sizes, batch.extra), make())
        atk.close()
        outs[mode] = (x.clone(), [float(v) for v in atk.last_losses])
    assert torch.equal(outs["eager"][0], outs["graph"][0]), "iterate after %d captured iterations" % n
    assert outs["eager"][1] == outs["graph"][1] and len(outs["graph"][1]) == n
    if detector == "toy":
        for k in range(n + 1):
            for eye in ("image_2", "image_3"):
                for name in batch.names:
                    a = open(str(tmp_path / "eager" / ("dsgn_pgd_iters_%d" % k) / eye / (name + ".png")), "rb").read()
                    b = open(str(tmp_path / "graph" / ("dsgn_pgd_iters_%d" % k) / eye / (name + ".png")), "rb").read()
                    assert a == b, (k, eye, name)


def test_roi_path_and_2d_convolution_entry_points_are_capturable():
    """include/advengine.h: "enqueue-only ... safe to capture in a hipGraph" for the entry points the DSGN-shaped capture does not
    reach: RoIAlign forward / backward (tile lists + gather), NMS (its memset included), dense alignment, the 1x1 convolution GEMM"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    feat = torch.tensor(rs.randn(1, 16, 38, 125).astype(np.float32), device=dev)
    rois = torch.tensor(np.float32([[0, 40, 30, 400, 300], [0, 900, 100, 1500, 500], [0, 0, 0, 1986, 599], [0, 5, 5, 9, 9]]), device=dev)
    boxes = torch.tensor(np.float32([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10], [21, 21, 29, 29]]), device=dev)
    scores = torch.tensor(np.float32([0.9, 0.8, 0.7, 0.6, 0.5]), device=dev)
    g = torch.tensor(rs.randn(4, 16, 7, 7).astype(np.float32), device=dev)
    x2 = torch.tensor(rs.randn(2, 24, 19, 63).astype(np.float32), device=dev)
    prep = ops.Conv2dPrep(torch.tensor((rs.randn(40, 24, 1, 1) * 0.2).astype(np.float32), device=dev))

    def work():
        out = ops.roi_align(feat, rois, 7, 1 / 16.0, 0)
        gf = ops.roi_align_bwd(g, rois, feat.shape, 1 / 16.0, 0)
        keep, count = ops.nms_padded(boxes, scores, 0.5)
        y = ops.conv2d(x2, prep, relu=True)
        gx = ops.conv2d_dgrad(y, prep, mask=x2)
        return out, gf, keep, count, y, gx

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        want = [t.clone() for t in work()]
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = work()
    for t in got:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert got[2][:int(got[3])].tolist() == [0, 2]
