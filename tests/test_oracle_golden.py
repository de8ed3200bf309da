"""The numpy oracle against golden vectors produced by executing the reference's own
statements (tests/golden/make_golden.py).  Bit-exact: arrays are compared as raw bytes."""
import hashlib
import random

import numpy as np
import pytest

import synth
from oracle import oracle_np as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero(a.view(np.uint8).reshape(-1) != b.view(np.uint8).reshape(-1))
        raise AssertionError("bit mismatch, first differing byte %d of %d" % (bad[0], a.nbytes))


DSGN_PGD = ["dsgn_pgd_default", "dsgn_pgd_fgsm", "dsgn_pgd_cfg2", "dsgn_pgd_specials", "dsgn_pgd_ragged", "dsgn_pgd_padded"]
SRCNN_PGD = ["srcnn_pgd_default", "srcnn_pgd_cfg3", "srcnn_pgd_specials", "srcnn_pgd_zero_eps", "srcnn_pgd_zero_eps_tiny_alpha"]


@pytest.mark.parametrize("name", DSGN_PGD)
def test_dsgn_pgd_steps(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    same_bits(O.denormalize(g["x0L"]), g["cleanL"])
    same_bits(O.denormalize(g["x0R"]), g["cleanR"])
    same_bits(O.tensor2im_u8(g["x0L"][0], m["crop_h"], m["crop_w"]), g["u8L_0"])
    for eye in "LR":
        x, clean = g["x0" + eye], g["clean" + eye]
        for k in range(m["n_iter"]):
            x = O.pgd_step_norm01(x, g["g%s_%d" % (eye, k)], clean, m["alpha"], m["eps"])
            same_bits(x, g["x%s_%d" % (eye, k + 1)])
            same_bits(O.tensor2im_u8(x[0], m["crop_h"], m["crop_w"]), g["u8%s_%d" % (eye, k + 1)])
            assert sha(x) == m["digests"]["x%s_%d" % (eye, k + 1)]


@pytest.mark.parametrize("case", ["dsgn_pgd_fullsize", "dsgn_pgd_fullsize_padded"])
def test_dsgn_pgd_fullsize_digests(case, golden_index):
    m = golden_index["cases"][case]
    for eye, off in (("L", 0), ("R", 1)):
        if m.get("padded"):     # a 375x1242 image zero-padded in normalised space to the 384x1248 network size
            x = synth.dsgn_padded(m["seed"] + off, m["crop_h"], m["crop_w"], m["h"], m["w"])
        else:
            x = synth.dsgn_normalised(m["seed"] + off, m["h"], m["w"])
        clean = O.denormalize(x)
        for k in range(m["n_iter"]):
            g = synth.gradient(1000 * m["seed"] + 2 * k + off, x.shape, m["grad_scale"])
            x = O.pgd_step_norm01(x, g, clean, m["alpha"], m["eps"])
            assert sha(x) == m["digests"]["x%s_%d" % (eye, k + 1)]
            assert sha(O.tensor2im_u8(x[0], m["crop_h"], m["crop_w"])) == m["digests"]["u8%s_%d" % (eye, k + 1)]


@pytest.mark.parametrize("name", SRCNN_PGD)
def test_srcnn_pgd_steps(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    for eye in "LR":
        x = g["x0" + eye]
        clean = x.copy()
        for k in range(m["n_iter"]):
            x = O.pgd_step_meansub255(x, g["g%s_%d" % (eye, k)], clean, m["alpha"], m["eps"])
            same_bits(x, g["x%s_%d" % (eye, k + 1)])
            same_bits(O.srcnn_hwc_plus_means(x[0]), g["hwc%s_%d" % (eye, k + 1)])


def test_srcnn_pgd_fullsize_digests(golden_index):
    m = golden_index["cases"]["srcnn_pgd_fullsize"]
    x = synth.srcnn_meansub(m["seed"], m["h"], m["w"])
    clean = x.copy()
    for k in range(m["n_iter"]):
        g = synth.gradient(2000 * m["seed"] + 2 * k, x.shape, m["grad_scale"])
        x = O.pgd_step_meansub255(x, g, clean, m["alpha"], m["eps"])
        assert sha(x) == m["digests"]["xL_%d" % (k + 1)]
        assert sha(O.srcnn_hwc_plus_means(x[0])) == m["digests"]["hwcL_%d" % (k + 1)]


PATCH = ["dsgn_patch_default", "dsgn_patch_zero", "dsgn_patch_100px", "srcnn_patch_default"]


@pytest.mark.parametrize("name", PATCH)
def test_patch_paste_and_update(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    dsgn = m["model"] == "dsgn"
    H, W, r = m["H"], m["W"], m["radius"]
    assert O.init_patch_dims(384 if dsgn else 600, m["ratio"]) == (m["patch_dim"], r)
    random.seed(m["seed"])
    cl, cr = O.round_mask_centers(random, H, W, r)
    assert cl == m["center_l"] and cr == m["center_r"]
    cy, cxl, cxr = cl[0], cl[1], cr[1]
    ml, mr = O.disc_mask(H, W, cy, cxl, r), O.disc_mask(H, W, cy, cxr, r)
    assert sha(ml) == m["digests"]["mask_l"] and sha(mr) == m["digests"]["mask_r"]
    assert int(ml.sum()) == m["mask_area"]
    mk = synth.dsgn_normalised if dsgn else synth.srcnn_meansub
    xL, xR = mk(m["seed"] + 10, H, W), mk(m["seed"] + 11, H, W)
    patch = g["patch_0"]
    gaccL = gaccR = None
    lo, hi = (None, None) if dsgn else (O.SRCNN_LO, O.SRCNN_HI)
    for k in range(m["iters"]):
        xL, xR = O.patch_paste(xL, patch, cy, cxl, r), O.patch_paste(xR, patch, cy, cxr, r)
        assert sha(xL) == m["digests"]["pastedL_%d" % k]
        assert sha(xR) == m["digests"]["pastedR_%d" % k]
        gl = synth.gradient(3000 * m["seed"] + 2 * k, xL.shape, m["grad_scale"])
        gr = synth.gradient(3000 * m["seed"] + 2 * k + 1, xR.shape, m["grad_scale"])
        gaccL = gl if gaccL is None else gaccL + gl      # autograd accumulation across inner iterations
        gaccR = gr if gaccR is None else gaccR + gr
        assert sha(gaccL) == m["digests"]["gradL_%d" % k]
        patch = O.patch_update(patch, gaccL, gaccR, cy, cxl, cxr, r, m["eps"], lo=lo, hi=hi)
        same_bits(patch, g["patch_%d" % (k + 1)])


def test_init_patch_dims(golden_index):
    for row in golden_index["masks"]["init_patch"]:
        short = 384 if row["model"] == "dsgn" else 600
        assert O.init_patch_dims(short, row["ratio"]) == (row["patch_dim"], row["radius"]), row


def test_round_mask_centers_and_masks(golden_index):
    for row in golden_index["masks"]["centers"]:
        h, w = (384, 1248) if row["model"] == "dsgn" else (600, 1987)
        random.seed(row["seed"])
        cl, cr = O.round_mask_centers(random, h, w, row["radius"], row["atk_mode"])
        assert cl == row["center_l"] and cr == row["center_r"], row
        assert row["mask_shape"] == [1, 3, h, w] and row["mask_dtype"] == "float32"
        m = O.disc_mask(h, w, cl[0], cl[1], row["radius"])
        assert int(m.sum()) == row["area"] and sha(m) == row["mask_l"]
        assert sha(O.disc_mask(h, w, cr[0], cr[1], row["radius"])) == row["mask_r"]


def test_disc_mask_equals_integer_test():
    # the kernels use dy*dy + dx*dx <= r*r in integers; the reference uses float64 sqrt
    for r in (1, 2, 30, 38, 49, 50, 127):
        d = 2 * r + 1
        m = O.disc_mask(d + 4, d + 6, r + 2, r + 3, r)
        yy, xx = np.mgrid[:d + 4, :d + 6]
        ref = ((yy - (r + 2)) ** 2 + (xx - (r + 3)) ** 2 <= r * r).astype(np.float32)
        assert np.array_equal(m, ref)


def test_kitti_label_text(golden_index):
    L = golden_index["label"]
    text = ""
    for i in range(len(L["labels"])):
        text += O.kitti_label_line(L["labels"][i], np.float32(L["bbox"][i]), np.float32(L["scores"][i]),
                                   np.float32(L["corners"][i]), L["dims"][i])
    assert text == L["text"]
