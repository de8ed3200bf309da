#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of the test-suite: run by hand / by gpurun).  Random shapes through the HIP
kernels against the oracle, bit for bit: convolution (plain / bias+ReLU / masked / strided / transposed, every alternative code path),
grid_sample3d forward + gather backward, bilinear up-sampling + gather adjoint, RoIAlign forward + backward, cost volume, PGD steps with and without the 8-bit index.
usage: python tools/fuzz_gpu.py [--cases 150] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from eval_driving_safety_amd import ops  # noqa: E402
from oracle import oracle_c as C  # noqa: E402
from oracle import oracle_np as O  # noqa: E402
import synth  # noqa: E402


def same(a, b, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    if a.tobytes() != np.ascontiguousarray(b).tobytes():
        bad = int((a.view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)).sum()) if a.shape == b.shape else -1
        raise AssertionError("%s: %d elements differ" % (what, bad))


def conv_case(rs, dev, big=False):
    b = int(rs.randint(1, 3))
    cin = 4 * int(rs.randint(1, 5))
    cout = int(rs.choice([1, 3, 8, 12, 32, 33, 40, 64]))
    d, h = int(rs.randint(1, 9)), int(rs.randint(1, 30))
    w = int(rs.choice([4, 8, 31, 32, 33, 36, 40, 64, 78, 100]))
    if big:      # more tiles than resident workgroups: the persistent walk with several tiles per workgroup
        cin, cout = 4 * int(rs.randint(1, 3)), int(rs.choice([32, 40, 64]))
        d, h, w = int(rs.randint(20, 40)), int(rs.randint(40, 80)), int(rs.choice([156, 160, 200, 310, 312]))
    x = rs.randn(b, cin, d, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3, 3) * 0.1).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32) if rs.rand() < 0.5 else None
    relu = bool(rs.rand() < 0.5)
    mask = int(rs.randint(1, 1 << 27)) if rs.rand() < 0.4 else (1 << 27) - 1
    sw = rs.choice(["", "", "ADV_CONV_NO_DMA", "ADV_CONV_TH4", "ADV_CONV_TH8", "ADV_CONV_TH44", "ADV_CONV_ONE_TILE_PER_WG", "ADV_CONV_GENERIC", "ADV_CONV_T_CLASS_TILES",
                    "ADV_CONV_CLASS_LAUNCHES"])
    skip = bool(rs.rand() < 0.4)
    env = {"ADV_CONV_TH4": ("ADV_CONV_TH", "4"), "ADV_CONV_TH8": ("ADV_CONV_TH", "8"), "ADV_CONV_TH44": ("ADV_CONV_TH", "44")}.get(sw, (sw, "1") if sw else None)
    import contextlib
    from eval_driving_safety_amd import _lib
    with contextlib.ExitStack() as stack:        # a switched case runs on the -DADV_TEST_HOOKS build, an unswitched one on the shipped library
        if env:
            stack.enter_context(_lib.using(_lib.HOOKS_LIB_PATH))
            os.environ[env[0]] = env[1]
            stack.callback(os.environ.pop, env[0])
        tx, tw = torch.tensor(x, device=dev), torch.tensor(wt, device=dev)
        tb = None if bias is None else torch.tensor(bias, device=dev)
        def plus_skip(conv, sk):        # the epilogue's order: accumulate, + bias, + skip, max
            if sk is None:
                return conv(relu)
            r = conv(False) + sk
            return np.maximum(r, np.float32(0)) if relu else r
        sk = rs.randn(b, cout, d, h, w).astype(np.float32) if skip else None
        y = ops._conv3d_ex(tx, ops.conv3d_k3_prep(tw), cout, 1, relu, tb, mask, residual=None if sk is None else torch.tensor(sk, device=dev))
        same(y, plus_skip(lambda r: C.conv3d_k3_ex(x, wt, bias=bias, relu=r, tap_mask=mask), sk),
             "conv %s" % ((b, cin, cout, d, h, w, relu, hex(mask), sw, skip),))
        if rs.rand() < 0.4:
            ys = ops.conv3d_k3_s2(tx, ops.conv3d_k3_s2_prep(tw), cout, relu=relu, bias=tb, route="s2d")
            same(ys, C.conv3d_k3_s2(x, wt, bias=bias, relu=relu), "strided conv %s" % ((b, cin, cout, d, h, w, sw),))
            yd = ops.conv3d_k3_s2(tx, ops.conv3d_k3_prep(tw), cout, relu=relu, bias=tb)       # direct: matrix kernel (2-channel stages) or scalar staging (4)
            same(yd, C.conv3d_k3_ex(x, wt, bias=bias, stride=2, relu=relu, chunk=ops.conv3d_k3_s2_stage_channels(tx, cout)),
                 "direct strided conv %s" % ((b, cin, cout, d, h, w, sw),))
        if rs.rand() < 0.4:
            wtt = (rs.randn(cin, cout, 3, 3, 3) * 0.1).astype(np.float32)
            sk2 = rs.randn(b, cout, 2 * d, 2 * h, 2 * w).astype(np.float32) if skip else None
            yt = ops.conv_transpose3d_k3_s2(tx, ops.conv_transpose3d_k3_s2_prep(torch.tensor(wtt, device=dev)), cout, relu=relu, bias=tb,
                                            residual=None if sk2 is None else torch.tensor(sk2, device=dev))
            same(yt, plus_skip(lambda r: C.conv_transpose3d_k3_s2(x, wtt, bias=bias, relu=r), sk2), "transposed conv %s" % ((b, cin, cout, d, h, w, sw, skip),))


def conv2d_case(rs, dev):
    """csrc/conv2d.hip: 1x1 and 3x3 (dilation 1 / 2) layers of random shape, every tile shape, random epilogue (bias, skip connection,
    ReLU, mask), forward and backward w.r.t. the input, against the oracle's fmaf chain bit for bit"""
    k = int(rs.choice([1, 3]))
    dil = int(rs.choice([1, 2])) if k == 3 else 1
    b = int(rs.randint(1, 4))
    cin, cout = int(rs.choice([1, 3, 8, 13, 16, 24, 64, 130])), int(rs.choice([1, 6, 18, 32, 33, 64, 70, 128]))
    h, w = int(rs.randint(1, 40)), int(rs.choice([1, 2, 5, 31, 32, 33, 41, 63, 64, 97]))
    if b * min(cin, cout) * h * w < 4:          # the kernels load whole float4s: tensors of fewer than four floats are refused (ADV_EINVAL)
        h = 4
    x = rs.randn(b, cin, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, k, k) * (1.0 / (cin * k * k)) ** 0.5).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32) if rs.rand() < 0.6 else None
    res = rs.randn(b, cout, h, w).astype(np.float32) if rs.rand() < 0.5 else None
    mask = rs.randn(b, cout, h, w).astype(np.float32) if rs.rand() < 0.3 else None
    relu = bool(rs.rand() < 0.5)
    pad = dil if k == 3 else 0
    chunk = 8 if k == 3 else 16
    tile = int(rs.randint(-1, 6 if k == 1 else 3))      # 1x1: -1 .. 5 (all six shapes), 3x3: -1 .. 2
    t = lambda a: None if a is None else torch.tensor(a, device=dev)       # noqa: E731
    prep = ops.Conv2dPrep(t(wt), 1, pad, dil)
    y = ops.conv2d(t(x), prep, t(bias), t(res), relu, t(mask), tile=tile)
    same(y, C.conv2d(x, wt, bias, res, mask, padding=pad, dilation=dil, relu=relu, chunk=chunk), "conv2d %s" % ((k, dil, b, cin, cout, h, w, relu, tile),))
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gres = rs.randn(b, cin, h, w).astype(np.float32) if rs.rand() < 0.5 else None
    gmask = x if rs.rand() < 0.5 else None
    gx = ops.conv2d_dgrad(t(g), prep, residual=t(gres), mask=t(gmask), tile=tile)
    same(gx, C.conv2d(g, wt, residual=gres, mask=gmask, padding=pad, dilation=dil, transpose=True, chunk=chunk), "conv2d dgrad %s" % ((k, dil, b, cin, cout, h, w, tile),))
    if prep.has_wino:       # csrc/wino2d.hip against its own restatement, both tile shapes
        wtile = int(rs.randint(-1, 8))
        y = ops.conv2d(t(x), prep, t(bias), t(res), relu, t(mask), tile=wtile, wino=True)
        same(y, C.conv2d_wino(x, wt, bias, res, mask, relu=relu), "conv2d wino %s" % ((b, cin, cout, h, w, relu, wtile),))
        gx = ops.conv2d_dgrad(t(g), prep, residual=t(gres), mask=t(gmask), tile=wtile, wino=True)
        same(gx, C.conv2d_wino(g, wt, residual=gres, mask=gmask, transpose=True), "conv2d wino dgrad %s" % ((b, cin, cout, h, w, wtile),))


def wino3d_case(rs, dev):
    """csrc/wino2d.hip on 3x3x3 layers: random shapes (one plane, two planes, odd maps, channels around the stage / block sizes), every
    workgroup shape, random epilogue, forward and backward w.r.t. the input, against the restatement bit for bit"""
    b = int(rs.randint(1, 3))
    cin, cout = int(rs.choice([1, 3, 4, 8, 13, 32, 40, 64])), int(rs.choice([4, 6, 18, 32, 33, 64, 70]))
    d, h, w = int(rs.randint(1, 5)), int(rs.randint(1, 30)), int(rs.choice([1, 2, 5, 23, 24, 31, 32, 33, 49, 65]))
    if b * min(cin, cout) * d * h * w < 4:
        h = 4
    x = rs.randn(b, cin, d, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3, 3) * (1.0 / (27 * cin)) ** 0.5).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32) if rs.rand() < 0.6 else None
    res = rs.randn(b, cout, d, h, w).astype(np.float32) if rs.rand() < 0.5 else None
    mask = rs.randn(b, cout, d, h, w).astype(np.float32) if rs.rand() < 0.3 else None
    relu = bool(rs.rand() < 0.5)
    tile = int(rs.randint(-1, 8))
    t = lambda a: None if a is None else torch.tensor(a, device=dev)       # noqa: E731
    prep = ops.Conv3dWinoPrep(t(wt))
    same(ops.conv3d_wino(t(x), prep, t(bias), t(res), relu, t(mask), tile=tile), C.conv3d_wino(x, wt, bias, res, mask, relu=relu),
         "conv3d wino %s" % ((b, cin, cout, d, h, w, relu, tile),))
    g = rs.randn(b, cout, d, h, w).astype(np.float32)
    gres = rs.randn(b, cin, d, h, w).astype(np.float32) if rs.rand() < 0.5 else None
    gmask = x if rs.rand() < 0.5 else None
    same(ops.conv3d_wino_dgrad(t(g), prep, residual=t(gres), mask=t(gmask), tile=tile), C.conv3d_wino(g, wt, residual=gres, mask=gmask, transpose=True),
         "conv3d wino dgrad %s" % ((b, cin, cout, d, h, w, tile),))


def wino4_case(rs, dev):
    """csrc/wino4.hip (Winograd F(4x4,3x3)) on 3x3 and 3x3x3 layers of random shape - odd maps, one-pixel maps, channels around the stage /
    block sizes, every workgroup shape incl. the two-images-per-tile one, random epilogue - forward and backward w.r.t. the input, against
    its restatement (oracle.c orc_conv_wino4) bit for bit"""
    three_d = bool(rs.rand() < 0.4)
    b = int(rs.randint(1, 4))
    cin, cout = int(rs.choice([1, 3, 4, 8, 13, 32, 40, 64, 100])), int(rs.choice([1, 4, 6, 18, 32, 33, 64, 70, 128]))
    d = int(rs.randint(1, 4)) if three_d else 1
    h, w = int(rs.randint(1, 36)), int(rs.choice([1, 2, 5, 14, 15, 16, 31, 32, 33, 49, 65, 90]))
    if b * min(cin, cout) * d * h * w < 4:          # (tensors of fewer than four floats are refused)
        h = 4
    shape = (b, cin, d, h, w) if three_d else (b, cin, h, w)
    oshape = (b, cout) + shape[2:]
    x = rs.randn(*shape).astype(np.float32)
    wt = (rs.randn(*((cout, cin) + ((3, 3, 3) if three_d else (3, 3)))) * (1.0 / (cin * (27 if three_d else 9))) ** 0.5).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32) if rs.rand() < 0.6 else None
    res = rs.randn(*oshape).astype(np.float32) if rs.rand() < 0.5 else None
    mask = rs.randn(*oshape).astype(np.float32) if rs.rand() < 0.3 else None
    relu = bool(rs.rand() < 0.5)
    pair_ok = (not three_d) and w <= 15 and cin % 8 == 0
    tile = int(rs.choice([-1, 0, 1, 2, 3] + ([4] if pair_ok else [])))
    t = lambda a: None if a is None else torch.tensor(a, device=dev)       # noqa: E731
    prep = ops.ConvWino4Prep(t(wt))
    same(ops.conv_wino4(t(x), prep, t(bias), t(res), relu, t(mask), tile=tile), C.conv_wino4(x, wt, bias, res, mask, relu=relu),
         "wino4 %s" % ((three_d, b, cin, cout, d, h, w, relu, tile),))
    g = rs.randn(*oshape).astype(np.float32)
    gres = rs.randn(*shape).astype(np.float32) if rs.rand() < 0.5 else None
    gmask = x if rs.rand() < 0.5 else None
    dtile = tile if not (tile == 4 and cout % 8 != 0) else -1          # (the image-pair shape needs the contraction's channels in whole stages: eights)
    same(ops.conv_wino4_dgrad(t(g), prep, residual=t(gres), mask=t(gmask), tile=dtile), C.conv_wino4(g, wt, residual=gres, mask=gmask, transpose=True),
         "wino4 dgrad %s" % ((three_d, b, cin, cout, d, h, w, dtile),))
    if not three_d:      # the K-split launch: parts of the contraction on their own workgroups, added in order (its own order: oracle chunk=)
        ktile, parts = int(rs.choice([0, 1, 2, 3])), int(rs.choice([2, 3, 5]))
        same(ops.conv_wino4(t(x), prep, t(bias), t(res), relu, t(mask), tile=ktile, splits=parts),
             C.conv_wino4(x, wt, bias, res, mask, relu=relu, chunk=ops.conv_wino4_ksplit_chunk(cin, ktile, parts)), "wino4 k-split %s" % ((b, cin, cout, h, w, relu, ktile, parts),))
        same(ops.conv_wino4_dgrad(t(g), prep, residual=t(gres), mask=t(gmask), tile=ktile, splits=parts),
             C.conv_wino4(g, wt, residual=gres, mask=gmask, transpose=True, chunk=ops.conv_wino4_ksplit_chunk(cout, ktile, parts)), "wino4 k-split dgrad %s" % ((b, cin, cout, h, w, ktile, parts),))


def boxes_case(rs, dev):
    """csrc/boxes.hip: IoU rows + maxima and the stable size partition / roi sampling against the oracle / the tensor formulation (exact)"""
    n, m = int(rs.randint(1, 3000)), int(rs.randint(1, 20))
    def boxes(k):
        x1, y1 = rs.rand(k) * 1800, rs.rand(k) * 550
        return np.stack([x1, y1, x1 + rs.rand(k) * 300 + 1, y1 + rs.rand(k) * 200 + 1], 1).astype(np.float32)
    a, g = boxes(n), boxes(m)
    ta, tg = torch.tensor(a, device=dev), torch.tensor(g, device=dev)
    iou, best, arg = ops.box_iou_rows(ta, tg)
    want = O.box_iou(a, g)
    same(iou, want, "box_iou %s" % ((n, m),))
    assert np.array_equal(arg.cpu().numpy(), want.argmax(1)) and best.cpu().numpy().tobytes() == want.max(1).tobytes(), "box_iou maxima %s" % ((n, m),)
    right = torch.tensor(boxes(n), device=dev)
    big = torch.tensor((rs.rand(n) < rs.rand()).astype(np.int64), device=dev)
    pl, pr, nvalid = ops.box_partition_stereo(ta, right, big)
    perm = torch.argsort(1 - big, stable=True) if int(big.sum()) > 0 else torch.arange(n, device=dev)
    assert torch.equal(pl, ta[perm]) and torch.equal(pr, right[perm]) and int(nvalid) == (int(big.sum()) or n), "box_partition %s" % (n,)


def grid_case(rs, dev):
    b, c = int(rs.randint(1, 3)), int(rs.randint(1, 20))
    dims = tuple(int(v) for v in rs.randint(1, 12, 3))
    out = tuple(int(v) for v in rs.randint(1, 12, 3))
    align = bool(rs.rand() < 0.5)
    vol = rs.randn(b, c, *dims).astype(np.float32)
    grid = (rs.rand(b, *out, 3) * 2.6 - 1.3).astype(np.float32)
    tv, tg = torch.tensor(vol, device=dev), torch.tensor(grid, device=dev)
    same(ops.grid_sample3d(tv, tg, align), O.grid_sample3d(vol, grid, align), "grid_sample3d %s" % ((b, c, dims, out, align),))
    g = rs.randn(b, c, *out).astype(np.float32)
    plan = ops.GridSamplePlan(tg, dims, align)
    same(ops.grid_sample3d_bwd(torch.tensor(g, device=dev), plan), O.grid_sample3d_bwd(g, grid, dims, align), "grid_sample3d bwd %s" % ((b, c, dims, out, align),))


def resize_case(rs, dev):
    """bilinear up-sampling (any size pair, also equal sizes and down-sampling) and its gather adjoint"""
    lead = (int(rs.randint(1, 3)), int(rs.randint(1, 5)))
    h, w = int(rs.randint(1, 14)), int(rs.randint(1, 20))
    mode = int(rs.randint(4))
    if mode == 0:        # the pyramid's ragged ~2x steps
        ho, wo = 2 * h - int(rs.randint(2)), 2 * w - int(rs.randint(2))
    elif mode == 1:      # large ratios: more candidates per axis than the kernel keeps in registers
        ho, wo = h * int(rs.randint(3, 9)) + int(rs.randint(3)), w * int(rs.randint(3, 7)) + int(rs.randint(3))
    else:                # anything, down-sampling included
        ho, wo = int(rs.randint(1, 40)), int(rs.randint(1, 50))
    ho, wo = max(ho, 1), max(wo, 1)
    x = rs.randn(*lead, h, w).astype(np.float32)
    g = rs.randn(*lead, ho, wo).astype(np.float32)
    same(ops.bilinear_up(torch.tensor(x, device=dev), (ho, wo)), O.bilinear_up(x, (ho, wo)), "bilinear_up %s" % ((lead, h, w, ho, wo),))
    same(ops.bilinear_up_bwd(torch.tensor(g, device=dev), (h, w)), O.bilinear_up_bwd(g, (h, w)), "bilinear_up bwd %s" % ((lead, h, w, ho, wo),))


def roi_case(rs, dev):
    b, c = int(rs.randint(1, 3)), int(rs.randint(1, 40))
    h, w = int(rs.randint(2, 40)), int(rs.randint(2, 60))
    n = int(rs.randint(1, 30))
    scale = float(rs.choice([0.25, 0.125, 0.0625]))
    pooled = (int(rs.randint(1, 8)), int(rs.randint(1, 8)))
    sr = int(rs.choice([0, 1, 2, 3]))
    feat = rs.randn(b, c, h, w).astype(np.float32)
    x1, y1 = rs.rand(n) * w / scale * 0.9 - 5, rs.rand(n) * h / scale * 0.9 - 5
    rois = np.stack([rs.randint(0, b, n), x1, y1, x1 + rs.rand(n) * w / scale * 0.6 + 1, y1 + rs.rand(n) * h / scale * 0.6 + 1], 1).astype(np.float32)
    tf, tr = torch.tensor(feat, device=dev), torch.tensor(rois, device=dev)
    same(ops.roi_align(tf, tr, pooled, scale, sr), O.roi_align(feat, rois, pooled, scale, sr), "roi_align %s" % ((b, c, h, w, n, pooled, scale, sr),))
    g = rs.randn(n, c, *pooled).astype(np.float32)
    same(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, scale, sr), O.roi_align_bwd_ordered(g, rois, feat.shape, scale, sr, segments=ops.roi_align_bwd_segments(len(rois))),
         "roi_align bwd %s" % ((b, c, h, w, n, pooled, scale, sr),))


def depth_case(rs, dev):
    b, d, h, w = int(rs.randint(1, 3)), int(rs.randint(1, 10)), int(rs.randint(1, 12)), int(rs.randint(1, 14))
    out = (int(rs.randint(1, 30)), int(rs.randint(1, 30)), int(rs.randint(1, 40)))
    align = bool(rs.rand() < 0.5)
    cost = (rs.randn(b, d, h, w) * 2).astype(np.float32)
    zv = np.sort(rs.rand(out[0]) * 40 + 2).astype(np.float32)
    tc, tz = torch.tensor(cost, device=dev), torch.tensor(zv, device=dev)
    depth, stats = ops.depth_regress(tc, tz, out, align, with_stats=True)
    want, _ = O.depth_regress(cost, zv, out, align)
    np.testing.assert_allclose(depth.cpu().numpy(), want, rtol=2e-5, atol=1e-3, err_msg="depth_regress %s" % ((b, d, h, w, out, align),))
    g = rs.randn(*want.shape).astype(np.float32)
    gc = ops.depth_regress_bwd(tc, tz, depth, stats, torch.tensor(g, device=dev), align)
    ref = O.depth_regress_bwd(cost, zv, g, out, align)
    np.testing.assert_allclose(gc.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())),
                               err_msg="depth_regress bwd %s" % ((b, d, h, w, out, align),))


def pgd_case(rs, dev):
    n = int(rs.randint(1, 7))
    if rs.rand() < 0.5:
        h, w = int(rs.randint(2, 20)), 4 * int(rs.randint(1, 20))
        vh, vw = int(rs.randint(1, h + 1)), int(rs.randint(1, w + 1))
        sp = ops.Space.dsgn()
        x0 = np.concatenate([synth.dsgn_padded(int(rs.randint(1 << 20)), vh, vw, h, w) for _ in range(n)])
        step, alpha, eps, valid = O.pgd_step_norm01, 1 / 255, 0.03, (vh, vw)
        clean_np = O.denormalize(x0)
    else:
        h = int(rs.randint(2, 20))
        w = int(rs.randint(3, 60))
        if (h * w) % 4:
            h *= 4
        sp = ops.Space.srcnn()
        x0 = np.concatenate([synth.srcnn_meansub(int(rs.randint(1 << 20)), h, w) for _ in range(n)])
        step, alpha, eps, valid = O.pgd_step_meansub255, 1.0, 7.65, None
        clean_np = x0
    if rs.rand() < 0.3:
        x0[0] = x0[0] * np.float32(0.9993)                        # one image that is not 8-bit derived
        clean_np = O.denormalize(x0) if sp.affine else x0
    g = synth.gradient(int(rs.randint(1 << 20)), x0.shape, 1.0)
    x = torch.tensor(x0, device=dev)
    clean, ci = ops.denormalize_indexed(x, sp, valid=valid)
    want = x0
    for _ in range(2):
        want = step(want, g, clean_np, alpha, eps)
        ops.pgd_step(x, torch.tensor(g, device=dev), clean, sp, alpha, eps, out=x, clean_index=ci)
    same(x, want, "pgd %s" % ((n, h, w, sp.affine),))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", type=int, default=0, help="additional convolution cases with thousands of tiles")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(args.seed)
    kinds = [conv_case] * 5 + [conv2d_case] * 4 + [wino3d_case] * 3 + [wino4_case] * 4 + [boxes_case] * 1 + [grid_case] * 2 + [pgd_case] * 3 + [roi_case] * 3 + [depth_case] * 2 + [resize_case] * 2
    counts = {}
    for i in range(args.cases):
        fn = kinds[int(rs.randint(len(kinds)))]
        fn(rs, dev)
        counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
    for i in range(args.big):
        conv_case(rs, dev, big=True)
        counts["conv_case(big)"] = counts.get("conv_case(big)", 0) + 1
    torch.cuda.synchronize()
    print("fuzz ok: %s (seed %d)" % (", ".join("%s x%d" % kv for kv in sorted(counts.items())), args.seed))


if __name__ == "__main__":
    main()
