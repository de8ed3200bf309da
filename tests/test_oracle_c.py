"""The C oracle (oracle/oracle.c, the bench's cpu_baseline) against the same reference-generated
golden vectors, bit for bit."""
import numpy as np
import pytest

import synth
from test_oracle_golden import DSGN_PGD, SRCNN_PGD, PATCH, same_bits, sha

C = pytest.importorskip("oracle.oracle_c", reason="make -C oracle first (build() does it)")
from oracle import oracle_np as O  # noqa: E402


@pytest.mark.parametrize("name", DSGN_PGD)
def test_c_dsgn_pgd(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    same_bits(C.denormalize(g["x0L"]), g["cleanL"])
    same_bits(C.normalize(g["cleanL"]), O.normalize(g["cleanL"]))
    for eye in "LR":
        x, clean = g["x0" + eye], g["clean" + eye]
        for k in range(m["n_iter"]):
            x = C.pgd_step_norm01(x, g["g%s_%d" % (eye, k)], clean, m["alpha"], m["eps"])
            same_bits(x, g["x%s_%d" % (eye, k + 1)])
            same_bits(C.tensor2im_u8(x[0], m["crop_h"], m["crop_w"]), g["u8%s_%d" % (eye, k + 1)])


@pytest.mark.parametrize("name", SRCNN_PGD)
def test_c_srcnn_pgd(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    for eye in "LR":
        x = g["x0" + eye]
        clean = x.copy()
        for k in range(m["n_iter"]):
            x = C.pgd_step_meansub255(x, g["g%s_%d" % (eye, k)], clean, m["alpha"], m["eps"])
            same_bits(x, g["x%s_%d" % (eye, k + 1)])
            same_bits(C.srcnn_hwc_plus_means(x[0]), g["hwc%s_%d" % (eye, k + 1)])
            same_bits(C.srcnn_export_u8(x[0]), O.srcnn_export_u8(x[0]))


@pytest.mark.parametrize("name", PATCH)
def test_c_patch(name, golden, golden_index):
    g, m = golden(name), golden_index["cases"][name]
    dsgn = m["model"] == "dsgn"
    H, W, r = m["H"], m["W"], m["radius"]
    cy, cxl, cxr = m["center_l"][0], m["center_l"][1], m["center_r"][1]
    mk = synth.dsgn_normalised if dsgn else synth.srcnn_meansub
    xL = mk(m["seed"] + 10, H, W)
    patch = g["patch_0"]
    lo, hi = (None, None) if dsgn else (O.SRCNN_LO, O.SRCNN_HI)
    gaccL = gaccR = None
    for k in range(m["iters"]):
        xL = C.patch_paste(xL, patch, cy, cxl, r)
        assert sha(xL) == m["digests"]["pastedL_%d" % k]
        gl = synth.gradient(3000 * m["seed"] + 2 * k, xL.shape, m["grad_scale"])
        gr = synth.gradient(3000 * m["seed"] + 2 * k + 1, xL.shape, m["grad_scale"])
        gaccL = gl if gaccL is None else gaccL + gl
        gaccR = gr if gaccR is None else gaccR + gr
        patch = C.patch_update(patch, gaccL, gaccR, cy, cxl, cxr, r, m["eps"], lo=lo, hi=hi)
        same_bits(patch, g["patch_%d" % (k + 1)])


def test_c_disc_masks(golden_index):
    for row in golden_index["masks"]["centers"][:8]:
        h, w = (384, 1248) if row["model"] == "dsgn" else (600, 1987)
        assert sha(C.disc_mask(h, w, row["center_l"][0], row["center_l"][1], row["radius"])) == row["mask_l"]
