#!/usr/bin/env python3
"""Does the 1x1 GEMM kernel pay for partial rounds of the chip?  256 -> 1024 and 1024 -> 256 at pixel counts that make the 64 x 64-tile launch
exactly 1, 1.16 (the R101 layer), 1.25, 1.5 and 2 rounds of 256 CUs x 8 workgroups.  One JSON line per case: us, us per round-equivalent."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_wino_tiles import timed  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for cin, cout in ((256, 1024), (1024, 256)):
        prep = ops.Conv2dPrep(torch.randn((cout, cin, 1, 1), device=dev, generator=g) * 0.05, 1, 0, 1)
        bias = torch.randn((cout,), device=dev, generator=g)
        for h, w in ((64, 128), (76, 125), (64, 160), (96, 128), (128, 128), (100, 128), (112, 128)):
            x = torch.randn((1, cin, h, w), device=dev, generator=g)
            row = {"layer": "1x1 %d->%d" % (cin, cout), "pixels": h * w}
            for tile in (3, 2):
                ms = timed(lambda: ops.conv2d(x, prep, bias, None, True, tile=tile))
                bm, bn = (64, 64) if tile == 3 else (64, 128)
                wgs = ((cout + bm - 1) // bm) * ((h * w + bn - 1) // bn)
                row["tile_%dx%d" % (bm, bn)] = {"us": round(ms * 1e3, 1), "workgroups": wgs, "tflops": round(2.0 * cin * cout * h * w / ms / 1e9, 1)}
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
