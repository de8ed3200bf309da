#!/usr/bin/env python3
"""The kernels VERDICT r3 asked counters for, each launched alone on a named layer shape - the target of tools/gpu_profile_layers.sh
(rocprofv3 --pmc passes + a kernel trace; the program itself directly after ``--``).  Every case runs WARM + REPS launches of ONE kernel
(weight preparation and workspace kernels have other names and are filtered by the summariser); the launch order is written to
``--manifest`` so that tools/summarize_pmc_layers.py can attribute dispatches to cases.

cases: conv_wino on six 2D shapes and two 3D shapes, conv_wino4 on five 2D and three 3D shapes, conv2d_1x1_mfma, conv2d_3x3_mfma (dilation 2), conv3d_k3_s2_mfma and
convt3d_k3_s2_mfma on the hourglass shapes, roi_align_bwd_tab on ResNet-101-FPN-like proposals."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eval_driving_safety_amd import ops  # noqa: E402

WARM, REPS = 2, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest", default="gpurun_out/pmc_layers_manifest.json")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g)              # noqa: E731
    cases = []

    def case(name, kernel, flops, fn):
        if args.only and args.only not in name:
            return
        for _ in range(WARM + REPS):
            fn()
        torch.cuda.synchronize()
        cases.append({"case": name, "kernel": kernel, "launches": WARM + REPS, "warm": WARM, "direct_flops_per_launch": flops})

    # ---- 2D Winograd F(2x2,3x3), forward (bias + ReLU epilogue)
    for b, c, k, h, w in ((2, 64, 64, 96, 312), (2, 128, 128, 96, 312), (2, 256, 256, 38, 125), (2, 512, 512, 19, 63), (2, 256, 256, 150, 497), (1, 128, 128, 192, 304)):
        x, wt, bias = rnd(b, c, h, w), rnd(k, c, 3, 3) * 0.05, rnd(k)
        prep = ops.Conv2dPrep(wt, 1, 1, 1)
        prep.wino(False)
        case("wino2d %d->%d on [%d,%d,%d,%d]" % (c, k, b, c, h, w), "conv_wino", 2.0 * b * k * c * 9 * h * w, lambda: ops.conv2d(x, prep, bias, None, True, wino=True))
    # ---- 3D Winograd (plane transform, depth taps in the contraction)
    for c, k, d, h, w in ((32, 32, 48, 96, 312), (128, 128, 96, 10, 152)):
        x, wt, bias = rnd(1, c, d, h, w), rnd(k, c, 3, 3, 3) * 0.03, rnd(k)
        p3 = ops.Conv3dWinoPrep(wt)
        p3.u(False)
        case("wino3d %d->%d on [1,%d,%d,%d,%d]" % (c, k, c, d, h, w), "conv_wino", 2.0 * k * c * 27 * d * h * w, lambda: ops.conv3d_wino(x, p3, bias, None, True))
    # ---- Winograd F(4x4,3x3): 2D (one and two images per tile), 3D
    for b, c, k, h, w in ((2, 256, 256, 150, 497), (2, 128, 128, 96, 312), (2, 512, 512, 38, 125), (512, 256, 256, 14, 14), (1, 128, 128, 192, 304)):
        x, wt, bias = rnd(b, c, h, w), rnd(k, c, 3, 3) * 0.05, rnd(k)
        p4 = ops.ConvWino4Prep(wt)
        p4.u(False)
        case("wino4 2d %d->%d on [%d,%d,%d,%d]" % (c, k, b, c, h, w), "conv_wino4", 2.0 * b * k * c * 9 * h * w, lambda: ops.conv_wino4(x, p4, bias, None, True))
    for c, k, d, h, w in ((128, 128, 96, 10, 152), (64, 64, 24, 48, 156), (32, 32, 48, 96, 312)):
        x, wt, bias = rnd(1, c, d, h, w), rnd(k, c, 3, 3, 3) * 0.03, rnd(k)
        p4 = ops.ConvWino4Prep(wt)
        p4.u(False)
        case("wino4 3d %d->%d on [1,%d,%d,%d,%d]" % (c, k, c, d, h, w), "conv_wino4", 2.0 * k * c * 27 * d * h * w, lambda: ops.conv_wino4(x, p4, bias, None, True))
    # ---- direct 2D kernels
    for b, c, k, h, w in ((2, 256, 1024, 38, 125), (2, 64, 256, 150, 497), (2, 1024, 256, 38, 125)):
        x, wt, bias = rnd(b, c, h, w), rnd(k, c, 1, 1) * 0.05, rnd(k)
        prep = ops.Conv2dPrep(wt, 1, 0, 1)
        case("1x1 %d->%d on [%d,%d,%d,%d]" % (c, k, b, c, h, w), "conv2d_1x1_mfma", 2.0 * b * k * c * h * w, lambda: ops.conv2d(x, prep, bias, None, True))
    x, wt, bias = rnd(2, 128, 96, 312), rnd(128, 128, 3, 3) * 0.05, rnd(128)
    prep = ops.Conv2dPrep(wt, 1, 2, 2)
    case("3x3 dil2 128->128 on [2,128,96,312]", "conv2d_3x3_mfma", 2.0 * 2 * 128 * 128 * 9 * 96 * 312, lambda: ops.conv2d(x, prep, bias, None, True))
    # ---- strided / transposed 3D (the hourglass layers)
    for name, c, k, d, h, w in (("hg1", 32, 64, 48, 96, 312), ("gh1", 64, 128, 192, 20, 304), ("hg3", 64, 64, 24, 48, 156), ("gh3", 128, 128, 96, 10, 152)):
        x, wt, bias = rnd(1, c, d, h, w), rnd(k, c, 3, 3, 3) * 0.03, rnd(k)
        prep = ops.conv3d_k3_s2_prep(wt)
        od, oh, ow = (d + 1) // 2, (h + 1) // 2, (w + 1) // 2
        case("s2 %s %d->%d on [1,%d,%d,%d,%d]" % (name, c, k, c, d, h, w), "conv3d_k3_s2_mfma", 2.0 * k * c * 27 * od * oh * ow, lambda: ops.conv3d_k3_s2(x, prep, k, relu=True, bias=bias))
    for name, c, k, d, h, w in (("hg6", 64, 32, 24, 48, 156), ("gh6", 128, 64, 96, 10, 152), ("hg5", 64, 64, 12, 24, 78), ("gh5", 128, 128, 48, 5, 76)):
        x, wt, bias = rnd(1, c, d, h, w), rnd(c, k, 3, 3, 3) * 0.03, rnd(k)
        cls = ops.conv_transpose3d_k3_s2_prep(wt)
        # (a width that is not a multiple of 4 - 78 at 1/16 resolution - takes the register-staged variant of the same kernel)
        case("t2 %s %d->%d on [1,%d,%d,%d,%d]" % (name, c, k, c, d, h, w), "convt3d_k3_s2_mfma", 2.0 * k * c * 27 * d * h * w,
             lambda: ops.conv_transpose3d_k3_s2(x, cls, k, relu=True, bias=bias))
    # ---- RoIAlign backward: 512 proposals, 7x7 and 14x14, on a P2-sized map
    rs = torch.Generator().manual_seed(3)
    n = 512
    cx, cy = torch.rand(n, generator=rs) * 1900 + 40, torch.rand(n, generator=rs) * 520 + 40
    bw, bh = torch.rand(n, generator=rs) * 160 + 24, torch.rand(n, generator=rs) * 110 + 24
    rois = torch.stack([torch.zeros(n), cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).to(dev)
    for pooled in (7, 14):
        go = rnd(n, 256, pooled, pooled)
        case("roi_bwd 512 rois %dx%d C=256 on [1,256,150,497]" % (pooled, pooled), "roi_align_bwd_tab", 0.0, lambda: ops.roi_align_bwd(go, rois, (1, 256, 150, 497), 0.25, 0))
    os.makedirs(os.path.dirname(os.path.abspath(args.manifest)), exist_ok=True)
    if os.environ.get("PMC_WRITE_MANIFEST", "1") == "1":
        with open(args.manifest, "w") as f:
            json.dump({"warm": WARM, "reps": REPS, "cases": cases}, f, indent=1)
    print("ran %d cases" % len(cases))


if __name__ == "__main__":
    main()
