#!/usr/bin/env python3
"""The Winograd kernel's eight workgroup shapes (tile 0..7: 8x32 / 16x16 / 10x24 outputs x 64 or 32 output channels) on the 3x3 layer shapes of
both detector graphs, forward, one JSON line per layer: which shape is fastest where - the evidence behind pick_wino_tile (csrc/wino2d.hip).
Same bits whatever the shape (tests/test_conv2d.py)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402

LAYERS = [  # b, cin, cout, h, w
    (2, 32, 32, 192, 624), (2, 64, 64, 96, 312), (2, 128, 128, 96, 312), (8, 128, 128, 48, 156), (2, 64, 128, 96, 312), (2, 320, 128, 96, 312),
    (1, 128, 128, 192, 304), (1, 320, 128, 192, 304), (1, 256, 256, 96, 152), (1, 256, 256, 48, 76),
    (2, 64, 64, 150, 497), (2, 128, 128, 75, 249), (2, 256, 256, 38, 125), (2, 512, 512, 19, 63), (2, 256, 256, 150, 497), (2, 256, 256, 75, 249),
    (1, 256, 512, 150, 497), (1, 256, 512, 75, 249), (1, 256, 512, 38, 125), (1, 256, 512, 19, 63), (512, 256, 256, 14, 14),
]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for b, cin, cout, h, w in LAYERS:
        x = torch.randn((b, cin, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.05
        bias = torch.randn((cout,), device=dev, generator=g)
        prep = ops.Conv2dPrep(wt, 1, 1, 1)
        ms = {}
        for tile in (-1, 0, 1, 2, 3, 4, 5, 6, 7):
            ms["auto" if tile < 0 else str(tile)] = round(timed(lambda: ops.conv2d(x, prep, bias, None, True, tile=tile, wino=True)), 4)
        flops = 2.0 * b * cout * cin * 9 * h * w
        best = min((v, k) for k, v in ms.items() if k != "auto")
        print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), "ms_by_tile": ms, "best_tile": best[1], "best_ms": best[0],
                          "auto_over_best": round(ms["auto"] / best[0], 3), "direct_equiv_tflops_auto": round(flops / ms["auto"] / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
