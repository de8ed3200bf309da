"""CPU tests of everything around the kernels: the C-ABI surface, the host geometry/RNG logic against
the reference-generated golden index, the file surface, and the attack drivers run end-to-end with the
oracle standing in for the HIP ops (tests only - product code has no such path)."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

import _oracle_ops
import synth
from oracle import oracle_np as O
from eval_driving_safety_amd import _lib, adapters, attacks, data, patchgeom, pixelio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "advengine.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(adv_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 15
    lib = _lib.load()                         # works without a GPU: nothing is launched
    for name in declared:
        assert hasattr(lib, name), "libadvengine.so lacks %s" % name
    assert sorted(_lib.EXPORTED) == declared, "ctypes binding and header disagree"
    assert lib.adv_abi_version() == _lib.ABI_VERSION == 11
    assert lib.adv_strerror(-22) == b"invalid argument"


def test_shipped_library_reads_no_environment_variable():
    """include/advengine.h: "no environment variable is read" - the A/B route switches exist in the -DADV_TEST_HOOKS build only"""
    import subprocess
    assert _lib.load().adv_build_has_test_hooks() == 0
    with _lib.using(_lib.HOOKS_LIB_PATH) as hooks:
        assert hooks.adv_build_has_test_hooks() == 1
    assert _lib.load().adv_build_has_test_hooks() == 0, "using() must restore the shipped library"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "getenv" not in undefined, "libadvengine.so imports getenv"
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", _lib.HOOKS_LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    for name in os.listdir(os.path.join(ROOT, "eval_driving_safety_amd")):        # nothing in the package opens the hooks build
        if name.endswith(".py") and name != "_lib.py":
            assert "HOOKS_LIB_PATH" not in open(os.path.join(ROOT, "eval_driving_safety_amd", name)).read(), name


def test_header_is_valid_c99(tmp_path):
    """the boundary is a C ABI: the header must compile as plain C, and a C program must link against the library"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "advengine.h"\n#include <stdio.h>\nint main(void) { adv_space_t s; adv_space_dsgn(&s); '
                   'printf("%d %d %.3f\\n", adv_abi_version(), s.kind, s.scale[0]); return adv_pgd_step_f32(0, 0, 0, 0, 0, 1, 2, 2, &s, '
                   '0.1f, 0.1f, 2, 2, 0, 0, 0) == ADV_EINVAL ? 0 : 1; }\n')
    exe = tmp_path / "t"
    lib_dir = os.path.join(ROOT, "eval_driving_safety_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-l:libadvengine.so", "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["11", "0", "0.229"]


def test_space_constants_are_the_reference_constants():
    lib = _lib.load()
    s = _lib.AdvSpace()
    lib.adv_space_dsgn(ctypes.byref(s))
    assert s.kind == _lib.ADV_SPACE_AFFINE
    assert [np.float32(v) for v in s.scale] == [np.float32(v) for v in O.DSGN_STD]
    assert [np.float32(v) for v in s.shift] == [np.float32(v) for v in O.DSGN_MEAN]
    lib.adv_space_srcnn(ctypes.byref(s))
    assert s.kind == _lib.ADV_SPACE_IDENTITY
    assert [np.float32(v) for v in s.lo] == list(O.SRCNN_LO) and [np.float32(v) for v in s.hi] == list(O.SRCNN_HI)
    assert list(s.export_add) == list(O.SRCNN_PIXEL_MEANS)


def test_argument_errors_without_a_gpu():
    """validation happens before any launch, so the error paths are testable on CPU"""
    lib = _lib.load()
    s = _lib.AdvSpace()
    lib.adv_space_dsgn(ctypes.byref(s))
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.adv_pgd_step_f32(None, p, p, p, None, 1, 2, 2, ctypes.byref(s), 0.1, 0.1, 2, 2, 0, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_pgd_step_f32(p, p, p, p, None, 1, 2, 2, ctypes.byref(s), 0.1, -0.1, 2, 2, 0, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_pgd_step_f32(p, p, p, p, None, 0, 2, 2, ctypes.byref(s), 0.1, 0.1, 2, 2, 0, 0, None) == _lib.ADV_EINVAL
    odd = ctypes.c_void_p(p.value + 2)
    assert lib.adv_pgd_step_f32(odd, p, p, p, None, 1, 2, 2, ctypes.byref(s), 0.1, 0.1, 2, 2, 0, 0, None) == _lib.ADV_EALIGN
    assert lib.adv_patch_paste_f32(p, p, 4, 4, 3, 0, 1, 1, None) == _lib.ADV_EINVAL      # window leaves the image
    assert lib.adv_patch_paste_f32(p, p, 4, 4, 4, 2, 2, 1, None) == _lib.ADV_EINVAL      # d != 2r+1
    lib.adv_space_srcnn(ctypes.byref(s))
    assert lib.adv_denormalize_f32(p, p, 1, 2, 2, ctypes.byref(s), None) == _lib.ADV_EINVAL
    # the detector-side entry points validate before launching too
    assert lib.adv_conv3d_k3_f32(p, p, p, 1, 6, 8, 2, 2, 4, 0, None) == _lib.ADV_EINVAL               # Cin neither 1..3 nor a multiple of 4
    assert lib.adv_conv3d_k3_f32(p, None, p, 1, 4, 8, 2, 2, 4, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_conv3d_k3_s2_stage_channels(p, 64, 312) == 2 and lib.adv_conv3d_k3_s2_stage_channels(p, 32, 312) == 2      # host-side query, no launch
    assert lib.adv_conv3d_k3_s2_stage_channels(p, 64, 78) == 4 and lib.adv_conv3d_k3_s2_stage_channels(odd, 64, 312) == 4
    q = ctypes.c_void_p(0x2000)
    assert lib.adv_conv3d_k3_ex_f32(p, p, None, q, q, 1, 4, 8, 2, 2, 4, 1, 0, (1 << 27) - 1, None, 0, None, None, None, None) == _lib.ADV_EINVAL  # residual is y
    assert lib.adv_conv3d_k3_ex_f32(p, p, None, None, q, 1, 4, 8, 2, 2, 4, 3, 0, (1 << 27) - 1, None, 0, None, None, None, None) == _lib.ADV_EINVAL  # stride 3
    assert lib.adv_depth_regress_f32(None, p, p, None, 1, 2, 2, 2, 4, 4, 4, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_depth_regress_f32(p, p, p, None, 1, 2, 2, 2, 0, 4, 4, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_depth_regress_f32(odd, p, p, None, 1, 2, 2, 2, 4, 4, 4, 0, None) == _lib.ADV_EALIGN
    assert lib.adv_depth_regress_bwd_f32(p, p, p, p, p, None, p, 1, 2, 2, 2, 4, 4, 4, 0, None) == _lib.ADV_EINVAL   # no workspace
    assert lib.adv_grid_sample3d_f32(p, None, p, 1, 1, 2, 2, 2, 2, 2, 2, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_grid_sample3d_plan_bytes(1, 2, 3, 4, 5, 6, 7) == 4 * ((24 + 1) + 24 + 2 + 2 * 8 * 210)
    assert lib.adv_grid_sample3d_plan_bytes(1, 2048, 2048, 2048, 2, 2, 2) == 0                          # beyond a 32-bit plan
    assert lib.adv_grid_sample3d_plan_f32(p, None, 1, 2, 2, 2, 2, 2, 2, 0, None) == _lib.ADV_EINVAL
    assert lib.adv_grid_sample3d_bwd_f32(p, p, p, 1, 0, 2, 2, 2, 2, 2, 2, None) == _lib.ADV_EINVAL
    assert lib.adv_sigmoid_focal_loss_f32(p, p, None, None, 4, 1, 2.0, 0.25, None) == _lib.ADV_EINVAL   # nothing to write
    assert lib.adv_sigmoid_focal_loss_f32(p, p, p, None, 0, 1, 2.0, 0.25, None) == _lib.ADV_OK          # empty input: nothing to do
    assert lib.adv_roi_align_fwd_f32(p, p, p, 1, 1, 2, 2, 0, 2, 2, 0.25, 2, None) == _lib.ADV_OK        # zero rois


def test_ops_refuse_cpu_tensors():
    from eval_driving_safety_amd import ops
    x = torch.zeros((1, 3, 4, 4))
    with pytest.raises(TypeError):
        ops.pgd_step(x, x, x, ops.Space.dsgn(), 0.1, 0.1)


# ------------------------------------------------------------------------------------ geometry / RNG
def test_init_patch_dims_against_reference(golden_index):
    for row in golden_index["masks"]["init_patch"]:
        short = 384 if row["model"] == "dsgn" else 600
        assert patchgeom.init_patch_dims(short, row["ratio"]) == (row["patch_dim"], row["radius"]), row


def test_center_sampler_reproduces_the_reference_stream(golden_index):
    for row in golden_index["masks"]["centers"]:
        h, w = patchgeom.DSGN_SHAPE if row["model"] == "dsgn" else patchgeom.SRCNN_SHAPE
        random.seed(row["seed"])
        cl, cr = patchgeom.CenterSampler(h, w, row["radius"], row["atk_mode"], rng=random).draw()
        assert cl == row["center_l"] and cr == row["center_r"], row
        cl2, _ = patchgeom.CenterSampler(h, w, row["radius"], row["atk_mode"], seed=row["seed"]).draw()
        assert cl2 == row["center_l"]                       # random.Random(seed) == random.seed(seed)
    with pytest.raises(Exception):
        patchgeom.CenterSampler(384, 1248, 38, "sideways")


def test_fake_targets():
    bbox, box3d = torch.ones((3, 4)), torch.ones((3, 7))
    patchgeom.inject_fake_target_dsgn(bbox, box3d)
    assert torch.allclose(bbox[0], torch.tensor(patchgeom.DSGN_FAKE_BBOX)) and not bbox[1:].any()
    assert torch.allclose(box3d[0], torch.tensor(patchgeom.DSGN_FAKE_BOX3D)) and not box3d[1:].any()
    gl, gr, gm = (torch.ones((1, 30, 5)) for _ in range(3))
    assert patchgeom.inject_fake_target_srcnn(gl, gr, gm, [300, 700], [300, 636], 30) == 1
    assert gl[0, 0].tolist() == [670, 270, 730, 330, 0] and gr[0, 0].tolist() == [606, 270, 666, 330, 0]
    assert torch.equal(gm, gl) and not gl[0, 1:].any()


# ------------------------------------------------------------------------------------ file surface
def test_kitti_label_text_matches_reference(golden_index, tmp_path):
    L = golden_index["label"]
    dets = []
    for i in range(len(L["labels"])):
        x = np.float32(L["corners"][i]).reshape(8, 3)
        center = ((((x[0] + x[4]) + (x[1] + x[5])) + (x[2] + x[6])) + (x[3] + x[7])) / np.float32(8)
        dets.append((L["labels"][i], np.float32(L["bbox"][i]), np.float32(L["scores"][i]), center, L["dims"][i]))
    pixelio.write_kitti_labels(str(tmp_path), L["image_index"], dets)
    text = open(os.path.join(str(tmp_path), "000042.txt")).read()
    assert text == L["text"]
    # the consumer's parse (evaluation/convert_scenarios.py:74-93): split on ' ', fields 0..14
    for line in text.strip().split("\n"):
        f = line.split(" ")
        assert len(f) == 16 and f[0] in ("Car", "Pedestrian", "Cyclist")
        [float(v) for v in f[1:]]


def test_kitti_output_of_upstream_boxlists_matches_reference(golden_index, tmp_path):
    """the BoxList-shaped predictions of the upstream post-processor through pixelio.kitti_output: the same text the
    reference's own kitti_output wrote for them (get_dimensions is upstream code: stubbed identically in both)"""
    L = golden_index["label"]
    state = {"i": 0}

    def get_dimensions(c):
        assert tuple(c.shape) == (3, 8)
        d = L["dims"][state["i"]]
        state["i"] += 1
        return d[0], d[1], d[2], d[3]

    class Pred:
        bbox = torch.tensor(L["bbox"], dtype=torch.float32)
        f = {"labels": torch.tensor(L["labels"]), "scores": torch.tensor(L["scores"], dtype=torch.float32),
             "box_corner3d": torch.tensor(L["corners"], dtype=torch.float32)}

        def get_field(self, k):
            return self.f[k]

        def has_field(self, k):
            return k in self.f

    logged = []
    pixelio.kitti_output([Pred()], [L["image_index"]], str(tmp_path), get_dimensions, log=logged.append)
    assert open(os.path.join(str(tmp_path), "000042.txt")).read() == L["text"] and logged == ["Wrote 42"]


def test_consumer_view_of_label_files(golden_index, tmp_path):
    """load_label + the obstacle mapping against what the reference's own consumer code produced"""
    S = golden_index["scenario"]
    path = tmp_path / "000042.txt"
    path.write_text(S["text"])
    label = pixelio.load_label(str(path))
    assert label == S["label"]
    obs = pixelio.scenario_obstacles(label)
    assert len(obs) == len(S["obstacles"]) == 5
    for got, want in zip(obs, S["obstacles"]):
        assert got["width"] == want["width"] and got["length"] == want["length"]
        assert got["position"] == want["position"] and got["orientation"] == want["orientation"]
    assert all(-np.pi <= 0.5 * np.pi - o["orientation"] <= np.pi for o in obs)


def test_patch_files(tmp_path):
    d0 = pixelio.patch_dir("dsgn", 0.2, 0, str(tmp_path))
    assert d0.endswith(os.path.join("dsgn_patch_ratio_0.2", "epoch0"))
    p, existed = pixelio.load_or_init_patch(d0, 77)
    assert not existed and p.shape == (1, 3, 77, 77) and p.dtype == np.float32 and not p.any()
    saved = np.load(os.path.join(d0, "patch.npy"))
    assert saved.shape == (1, 3, 77, 77) and saved.dtype == np.float32
    pixelio.save_patch(d0, np.full((3, 77, 77), 0.5, np.float32))
    p, existed = pixelio.load_or_init_patch(d0, 77)
    assert existed and p.shape == (1, 3, 77, 77) and float(p.mean()) == 0.5
    with pytest.raises(ValueError):
        pixelio.load_or_init_patch(d0, 61)


def test_patch_resize_bilinear(tmp_path):
    """cross-model transfer branch of init_patch (attack/DSGN/patch_attack.py:220-227); unpinned vs cv2,
    checked for the properties of OpenCV's INTER_LINEAR: identity at equal size, exact on affine ramps
    in the interior, edge clamp, separability."""
    p = synth.patch_init(3, 61, -100, 100)
    assert pixelio.resize_patch_bilinear(p, 61).tobytes() == p.tobytes()
    yy, xx = np.mgrid[:61, :61].astype(np.float32)
    ramp = np.stack([2 * xx + 1, 3 * yy - 2, xx + yy])[None]
    out = pixelio.resize_patch_bilinear(ramp, 77)
    assert out.shape == (1, 3, 77, 77) and out.dtype == np.float32
    d = np.arange(77)
    f = (d + 0.5) * (61 / 77) - 0.5
    inner = (f > 0) & (f < 60)
    np.testing.assert_allclose(out[0, 0][:, inner], np.broadcast_to(2 * f[inner] + 1, (77, inner.sum())), rtol=0, atol=2e-4)
    np.testing.assert_allclose(out[0, 1][inner, :], np.broadcast_to((3 * f[inner] - 2)[:, None], (inner.sum(), 77)), rtol=0, atol=2e-4)
    assert out[0, 0][0, 0] == ramp[0, 0, 0, 0] and out[0, 0][0, -1] == ramp[0, 0, 0, -1]          # clamped edges
    d0 = pixelio.patch_dir("dsgn", 0.2, 0, str(tmp_path))
    pixelio.save_patch(d0, p)
    got, existed = pixelio.load_or_init_patch(d0, 77, allow_resize=True)
    assert existed and got.shape == (1, 3, 77, 77)
    with pytest.raises(ValueError):
        pixelio.load_or_init_patch(d0, 77)


# ------------------------------------------------------------------------------------ drivers on CPU
class _CpuToy(adapters.ToyStereoAdapter):
    def __init__(self, seed=0):
        super().__init__(torch.device("cpu"), seed=seed)


def _small_batch(n_pairs, h=48, w=64, seed=0, sizes=True):
    ls = [torch.from_numpy(synth.dsgn_normalised(seed + 2 * i, h, w)[0]) for i in range(n_pairs)]
    rs = [torch.from_numpy(synth.dsgn_normalised(seed + 2 * i + 1, h, w)[0]) for i in range(n_pairs)]
    names = ["%06d" % (7 + i) for i in range(n_pairs)]
    return attacks.StereoBatch(torch.stack(ls), torch.stack(rs), names, [(w - 3, h - 2)] * n_pairs if sizes else None)


def test_pgd_driver_files_and_iterates(tmp_path):
    from PIL import Image
    batch = _small_batch(2)
    toy = _CpuToy()
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, 3, out_root=str(tmp_path), ops=_oracle_ops, device=torch.device("cpu"))
    x = atk.run_batch(batch, toy)
    atk.close()
    # manual loop with the oracle
    xm = torch.cat([batch.imgL, batch.imgR]).numpy().copy()
    clean = O.denormalize(xm)
    want_png = {0: xm.copy()}
    for k in range(3):
        _, g = toy.loss_and_grad(torch.from_numpy(xm.copy()))
        xm = O.pgd_step_norm01(xm, g.numpy(), clean, 1 / 255, 0.03)
        want_png[k + 1] = xm.copy()
    assert x.numpy().tobytes() == xm.tobytes()
    for k in range(4):
        for eye, folder in ((0, "image_2"), (1, "image_3")):
            for i, name in enumerate(batch.names):
                path = os.path.join(str(tmp_path), "dsgn_pgd_iters_%d" % k, folder, name + ".png")
                got = np.array(Image.open(path).convert("RGB"))
                assert got.shape == (46, 61, 3)
                assert np.array_equal(got, O.tensor2im_u8(want_png[k][eye * 2 + i], 46, 61)), path
    assert sorted(os.listdir(str(tmp_path))) == ["dsgn_pgd_iters_%d" % k for k in range(4)]


def test_srcnn_pgd_driver_scales_eps_and_writes_bgr(tmp_path):
    from PIL import Image
    h, w = 40, 56
    l = torch.from_numpy(synth.srcnn_meansub(1, h, w))
    r = torch.from_numpy(synth.srcnn_meansub(2, h, w))
    batch = attacks.StereoBatch(l, r, ["000011.png"], None)
    toy = _CpuToy(seed=1)
    atk = attacks.PgdAttack("srcnn", 1.0, 0.03, 2, out_root=str(tmp_path), ops=_oracle_ops, device=torch.device("cpu"))
    assert atk.eps == 255 * 0.03                                            # pgd_attack.py:57
    x = atk.run_batch(batch, toy)
    atk.close()
    xm = torch.cat([l, r]).numpy().copy()
    clean = xm.copy()
    for k in range(2):
        _, g = toy.loss_and_grad(torch.from_numpy(xm.copy()))
        xm = O.pgd_step_meansub255(xm, g.numpy(), clean, 1.0, 255 * 0.03)
    assert x.numpy().tobytes() == xm.tobytes()
    got = np.array(Image.open(os.path.join(str(tmp_path), "stereo_rcnn_pgd_iters_2", "image_2", "000011.png")).convert("RGB"))
    assert np.array_equal(got[:, :, ::-1], O.srcnn_export_u8(xm[0]))       # file is RGB, tensor is BGR


def test_patch_trainer_reproduces_the_reference_sequence(tmp_path):
    """world 1, batch 1: paste -> fwd/bwd -> update with the gradient ACCUMULATING over the inner
    iterations, sequentially over images and epochs; positions from the seeded stream."""
    H, W = patchgeom.DSGN_SHAPE
    pairs = [(synth.dsgn_normalised(50 + 2 * i, H, W), synth.dsgn_normalised(51 + 2 * i, H, W)) for i in range(2)]

    def factory():
        return [attacks.StereoBatch(torch.from_numpy(l.copy()), torch.from_numpy(r.copy()), ["%06d" % i], [(1242, 375)])
                for i, (l, r) in enumerate(pairs)]

    toy = _CpuToy(seed=2)
    tr = attacks.PatchTrainer("dsgn", 0.2, 8 / 255, 2, 2, out_root=str(tmp_path), seed=5, ops=_oracle_ops, device=torch.device("cpu"))
    patch = tr.train(factory, toy)
    # manual
    rng = random.Random(5)
    D, r = O.init_patch_dims(384, 0.2)
    p = np.zeros((1, 3, D, D), np.float32)
    for epoch in range(2):
        for l, rr in pairs:
            cl, cr = O.round_mask_centers(rng, H, W, r)
            x = np.concatenate([l, rr]).copy()
            gacc = None
            for it in range(2):
                x[0:1] = O.patch_paste(x[0:1], p, cl[0], cl[1], r)
                x[1:2] = O.patch_paste(x[1:2], p, cr[0], cr[1], r)
                _, g = toy.loss_and_grad(torch.from_numpy(x.copy()))
                gacc = g.numpy() if gacc is None else gacc + g.numpy()
                p = O.patch_update(p, gacc[0:1], gacc[1:2], cl[0], cl[1], cr[1], r, 8 / 255)
    assert patch.numpy().tobytes() == p.tobytes()
    assert np.abs(p).max() > 0
    saved = np.load(os.path.join(str(tmp_path), "dsgn_patch_ratio_0.2", "epoch2", "patch.npy"))
    assert saved.shape == (1, 3, 77, 77) and saved.tobytes() == p.tobytes()
    assert os.path.exists(os.path.join(str(tmp_path), "dsgn_patch_ratio_0.2", "epoch0", "patch.npy"))


def test_patch_trainer_batched_sum_and_average(tmp_path):
    """B = 2 pairs per round against one snapshot: deltas summed (default) or averaged (average=True);
    accumulate_grad=False uses only the current iteration's gradient"""
    H, W = patchgeom.DSGN_SHAPE
    l = np.concatenate([synth.dsgn_normalised(80, H, W), synth.dsgn_normalised(82, H, W)])
    r = np.concatenate([synth.dsgn_normalised(81, H, W), synth.dsgn_normalised(83, H, W)])
    toy = _CpuToy(seed=6)
    D, rad = O.init_patch_dims(384, 0.2)
    results = {}
    for mode, kw in (("sum", {}), ("avg", {"average": True}), ("noacc", {"accumulate_grad": False})):
        out = tmp_path / mode
        tr = attacks.PatchTrainer("dsgn", 0.2, 8 / 255, 2, 1, out_root=str(out), seed=2, ops=_oracle_ops, device=torch.device("cpu"), **kw)
        batch = attacks.StereoBatch(torch.from_numpy(l.copy()), torch.from_numpy(r.copy()), ["000000", "000001"], [(1242, 375)] * 2)
        results[mode] = tr.train(lambda: [batch], toy).numpy().copy()
    rng = random.Random(2)
    cs = [O.round_mask_centers(rng, H, W, rad) for _ in range(2)]
    for mode in ("sum", "avg", "noacc"):
        p = np.zeros((1, 3, D, D), np.float32)
        x = np.concatenate([l, r]).copy()
        gacc = None
        for it in range(2):
            for i, (cl, cr) in enumerate(cs):
                x[i:i + 1] = O.patch_paste(x[i:i + 1], p, cl[0], cl[1], rad)
                x[2 + i:3 + i] = O.patch_paste(x[2 + i:3 + i], p, cr[0], cr[1], rad)
            _, g = toy.loss_and_grad(torch.from_numpy(x.copy()))
            g = g.numpy()
            gacc = g if (gacc is None or mode == "noacc") else gacc + g
            total = None
            for i, (cl, cr) in enumerate(cs):
                d = O.patch_delta(gacc[i:i + 1], gacc[2 + i:3 + i], cl[0], cl[1], cr[1], rad, 8 / 255)
                total = d if total is None else total + d
            if mode == "avg":
                total = total / np.float32(2)
            p = O.patch_apply_delta(p, total)
        assert results[mode].tobytes() == p.tobytes(), mode
    assert results["sum"].tobytes() != results["avg"].tobytes()


def test_patch_trainer_skips_wrong_shapes(tmp_path):
    b = _small_batch(1, sizes=False)
    tr = attacks.PatchTrainer("dsgn", 0.2, 8 / 255, 1, 1, out_root=str(tmp_path), seed=1, ops=_oracle_ops, device=torch.device("cpu"))
    patch = tr.train(lambda: [b], _CpuToy())
    assert not patch.numpy().any()                                   # patch_attack.py:318-320: skipped


def test_detect_under_attack_patch_mode(tmp_path):
    """paste at atk_mode positions, then labels; the consumer's parse of evaluation/convert_scenarios.py"""
    H, W = patchgeom.DSGN_SHAPE
    patch = torch.from_numpy(synth.patch_init(1, 77))
    l, r = synth.dsgn_normalised(90, H, W), synth.dsgn_normalised(91, H, W)
    seen = {}

    class Det:
        def detect(self, x, extra):
            seen["x"] = x.clone()
            return [[(2, np.float32([10, 20, 30, 40]), np.float32(0.9), np.float32([1.0, 1.5, 20.0]), (1.5, 1.6, 3.9, -1.57)),
                     (1, np.float32([1, 2, 3, 4]), np.float32(0.5), np.float32([-2.0, 1.2, 9.0]), (1.7, 0.6, 0.8, 0.3))]]

    dua = attacks.DetectUnderAttack("dsgn", "patch", str(tmp_path / "kitti_output_x"), patch=patch, atk_mode="sp_left",
                                    seed=3, ops=_oracle_ops, device=torch.device("cpu"))
    small = _small_batch(1, sizes=False)                                  # wrong shape: skipped
    big = attacks.StereoBatch(torch.from_numpy(l), torch.from_numpy(r), ["000123"], [(1242, 375)])
    assert dua.run([small, big], Det()) == 1
    (name, cl, cr), = dua.positions
    rng = random.Random(3)
    wl, wr = O.round_mask_centers(rng, H, W, 38, "sp_left")
    assert (cl, cr) == (wl, wr) and int(W * 0.2) <= cl[1] <= int(W * 0.4)
    assert seen["x"][0:1].numpy().tobytes() == O.patch_paste(l, patch.numpy(), cl[0], cl[1], 38).tobytes()
    assert seen["x"][1:2].numpy().tobytes() == O.patch_paste(r, patch.numpy(), cr[0], cr[1], 38).tobytes()
    lines = open(os.path.join(str(tmp_path), "kitti_output_x", "000123.txt")).read().strip().split("\n")
    assert [ln.split(" ")[0] for ln in lines] == ["Car", "Pedestrian"] and all(len(ln.split(" ")) == 16 for ln in lines)
    with pytest.raises(Exception):
        attacks.DetectUnderAttack("dsgn", "patch", str(tmp_path), patch=None, ops=_oracle_ops)


def test_result_folder_names():
    assert pixelio.dsgn_tag("", 4, 0.00392) == "_iter4_alpha0.00392"
    assert pixelio.dsgn_tag("", ratio=0.2, epochs=80) == "_ratio0.2_epochs80"
    assert pixelio.dsgn_label_dir("./outputs/temp/DSGN_car_pretrained/finetune_53.tar", "_ratio0.2_epochs80") == \
        "./outputs/temp/DSGN_car_pretrained/kitti_output_ratio0.2_epochs80"
    assert pixelio.srcnn_result_dir(1, 1) == "result_stereo_rcnn_pgd_1_1"
    assert pixelio.srcnn_result_dir(ratio=0.1, epochs=40) == "result_stereo_rcnn_ratio_0.1/epoch40"
    info = pixelio.srcnn_im_info_prescaled(np.float32([[600, 1987, 1.0]]))
    assert info[0][2] == np.float32(600 / 375) and info[0][0] == 600


# ------------------------------------------------------------------------------------ CLI surface
def test_cli_defaults_match_the_reference_scripts():
    from eval_driving_safety_amd.cli import dsgn_pgd_attack, dsgn_patch_attack, srcnn_pgd_attack, srcnn_patch_attack
    a = dsgn_pgd_attack.build_parser().parse_args(["-btest", "1", "-d", "0"])
    assert (a.iter, a.alpha, a.eps, a.seed, a.split_file, a.data_path) == (4, 1.0 / 255, 0.3, 1, "./data/kitti/val.txt", "./data/kitti/training")
    a = dsgn_patch_attack.build_parser().parse_args([])
    assert (a.iter, a.eps, a.epochs, a.ratio) == (2, 8.0 / 255, 80, 0.2)
    a = srcnn_pgd_attack.build_parser().parse_args([])
    assert (a.iter, a.alpha, a.eps) == (4, 1.0, 0.3)
    a = srcnn_patch_attack.build_parser().parse_args([])
    assert (a.iter, a.eps, a.epochs, a.ratio) == (2, 0.1, 40, 0.1)


def test_synthetic_source_shapes():
    b = next(iter(data.SyntheticStereo(2, "dsgn", batch=2)))
    assert tuple(b.imgL.shape) == (2, 3, 384, 1248) and b.sizes == [(1242, 375)] * 2 and b.names == ["000000", "000001"]
    assert float(b.imgL[:, :, 375:].abs().max()) == 0.0          # zero padding after normalisation
    b = next(iter(data.SyntheticStereo(1, "srcnn")))
    assert tuple(b.imgL.shape) == (1, 3, 600, 1987) and b.sizes is None


def test_depth_statistics_match_the_reference(golden_index):
    """error_estimating / depth_error_estimating / project_disp_to_depth* (attack/DSGN/predict_and_save_pgd.py:202-247,
    304-329) against the values the reference's own functions produced for the same seeded inputs"""
    import hashlib
    import types
    import torch
    from eval_driving_safety_amd import depthstats as D
    G = golden_index["depth_stats"]
    pred, gt = synth.depth_stats_inputs(G["seed"])
    tp, tg = torch.from_numpy(pred), torch.from_numpy(gt)
    calib = types.SimpleNamespace(P=np.array(G["P"]), f_u=G["f_u"])
    calib_R = types.SimpleNamespace(P=np.array(G["P_R"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    def same(got, want):
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert (a != a and b != b) or a == b, (got, want)

    same(D.error_estimating(tp, tg), G["error_estimating"])                      # an image without valid pixels: NaN, as the reference
    same(D.error_estimating(tp, tg, maxdisp=50), G["error_estimating_maxdisp50"])
    same(D.error_estimating(tp[[0, 2]], tg[[0, 2]]), G["error_estimating_valid_images"])
    same(D.depth_error_estimating(tp, tg, max_depth=G["max_depth"], depth_disp=True), G["depth_error_depth"])
    disp = torch.from_numpy(np.abs(pred) + 1)
    same(D.depth_error_estimating(disp, tg, depth_disp=False, calib_batch=[calib] * 3, calib_R_batch=[calib_R] * 3), G["depth_error_disp"])
    assert sha(D.project_disp_to_depth_map(G["f_u"], pred[0].copy(), 0.54, True)) == G["depth_map_depth"]
    assert sha(D.project_disp_to_depth_map(G["f_u"], pred[2].copy(), 0.532, False)) == G["depth_map_disp"]
    pts = D.project_disp_to_points(G["f_u"], pred[0].copy(), 0.54, True)
    pts = pts[(pts[:, 0] >= 0) & (pts[:, 2] < 30.)]           # the reference's filter after its (here: identity) velo transform
    assert pts.shape[0] == G["cloud_rows"] and sha(pts) == G["cloud"]


def test_ops_surface_is_complete():
    """every host-side operator the drivers, CLIs and INTEGRATION.md name is there (importing ops needs no GPU)"""
    from eval_driving_safety_amd import ops
    names = ["Space", "denormalize", "normalize", "CleanIndex", "can_index_clean", "denormalize_indexed", "import_u8", "alloc_u8", "pgd_step", "export_u8",
             "disc_mask", "patch_paste", "patch_paste_batch", "patch_update", "patch_delta_batch", "patch_apply",
             "psv_build", "psv_build_bwd", "PsvBuild", "psv_build_lerp", "psv_build_lerp_bwd", "PsvBuildLerp",
             "roi_align", "roi_align_bwd", "RoIAlign", "nms",
             "conv3d_k3_prep", "conv3d_k3", "conv3d_k3_s2", "conv_transpose3d_k3_s2_prep", "conv_transpose3d_k3_s2",
             "Conv3dK3", "Conv3dK3S2", "ConvTranspose3dK3S2",
             "dense_align_cost", "dense_align_argmin", "dense_align_search", "box_depth_offsets", "dense_align",
             "depth_regress", "depth_regress_bwd", "DepthRegress", "grid_sample3d", "GridSamplePlan", "grid_sample3d_bwd", "GridSample3d",
             "sigmoid_focal_loss", "SigmoidFocalLoss", "relu_backward", "stem_pool", "stem_pool_bwd", "StemPool", "conv_wino4", "ConvWino4Prep",
             "conv2d_supported", "Conv2dPrep", "conv2d", "conv2d_dgrad", "Conv2d", "Conv2dAuto", "bias_act_", "nms_padded"]
    missing = [n for n in names if not hasattr(ops, n)]
    assert not missing, missing
    # and every exported C symbol is reachable from some operator
    src = "".join(open(p).read() for p in ops.SOURCES)            # the package's modules, one per kernel family
    unused = [s for s in _lib.SIGNATURES if s not in src]
    assert not unused, unused


# ------------------------------------------------------------------------------------ command-line surface
CLI_COUNTERPARTS = {
    "attack/DSGN/pgd_attack.py": "dsgn_pgd_attack", "attack/DSGN/patch_attack.py": "dsgn_patch_attack",
    "attack/DSGN/predict_and_save_pgd.py": "dsgn_predict_and_save_pgd", "attack/DSGN/predict_and_save_patch.py": "dsgn_predict_and_save_patch",
    "attack/Stereo-RCNN/pgd_attack.py": "srcnn_pgd_attack", "attack/Stereo-RCNN/patch_attack.py": "srcnn_patch_attack",
    "attack/Stereo-RCNN/predict_and_save_pgd.py": "srcnn_predict_and_save_pgd",
    "attack/Stereo-RCNN/predict_and_save_patch.py": "srcnn_predict_and_save_patch",
}


@pytest.mark.parametrize("rel", sorted(CLI_COUNTERPARTS))
def test_cli_accepts_every_reference_flag_with_the_reference_default(rel, golden_index):
    """every add_argument of the reference script (option strings, default value, type, store_true) - read off the reference's
    parser set-up by tests/golden/make_golden.py - exists in the counterpart's parser with the same meaning"""
    import importlib
    want = golden_index["cli_flags"][rel]
    parser = importlib.import_module("eval_driving_safety_amd.cli." + CLI_COUNTERPARTS[rel]).build_parser()
    by_opt = {o: a for a in parser._actions for o in a.option_strings}
    assert want, rel
    for flag in want:
        acts = {id(by_opt[o]): by_opt[o] for o in flag["options"] if o in by_opt}
        assert len(acts) == 1 and all(o in by_opt for o in flag["options"]), (rel, flag["options"])
        (act,) = acts.values()
        if flag.get("action") == "store_true":
            assert type(act).__name__ == "_StoreTrueAction" and act.default is False, (rel, flag)
            continue
        assert act.default == flag.get("default"), (rel, flag, act.default)
        if "type" in flag:
            assert act.type is not None and act.type.__name__ == flag["type"], (rel, flag)
        if "dest" in flag:
            assert act.dest == flag["dest"], (rel, flag)
    # and the defaults parse: the reference's no-argument invocation is accepted as it stands
    parser.parse_args([])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """a launcher that started 2 ranks for `--gpus 4` must not produce a line labelled n_gpus 2: exit 2 before any GPU call"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_parent_relays_the_childs_failure():
    """no GPU here: the self-launched ranks fail, and the parent must exit non-zero without printing a JSON line"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "2"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
