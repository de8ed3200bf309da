#!/usr/bin/env python3
"""A/B of the F(4x4,3x3) kernel's wave priorities (hooks build, ADV_WINO4_FLAGS = 0 / 1 / 2; same bits).  One JSON line per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route  # noqa: E402
from bench_wino_tiles import timed  # noqa: E402


FLAGS = tuple(os.environ.get("ADV_BENCH_FLAGS", "0,1,2").split(","))      # 0 = shipped; 1 / 2 = wave priorities; 8 = 3x3x3 layers on the plain grid


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for b, cin, cout, h, w in ((2, 256, 256, 150, 497), (2, 128, 128, 96, 312), (2, 320, 128, 96, 312), (512, 256, 256, 14, 14)):
        x = torch.randn((b, cin, h, w), device=dev, generator=g)
        prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.05)
        ms = {}
        for rep in range(2):
            for f in FLAGS:
                with hooks_route(ADV_WINO4_FLAGS=f):
                    ms.setdefault(f, []).append(round(timed(lambda: ops.conv_wino4(x, prep, None, None, True)), 4))
        print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), "ms_by_flags": ms}), flush=True)
    for c, d, h, w in ((128, 96, 10, 152), (32, 48, 96, 312), (64, 24, 48, 156), (64, 48, 96, 312)):
        x = torch.randn((1, c, d, h, w), device=dev, generator=g)
        prep = ops.ConvWino4Prep(torch.randn((c, c, 3, 3, 3), device=dev, generator=g) * 0.05)
        ms = {}
        for rep in range(2):
            for f in FLAGS:
                with hooks_route(ADV_WINO4_FLAGS=f):
                    ms.setdefault(f, []).append(round(timed(lambda: ops.conv_wino4(x, prep, None, None, True)), 4))
        print(json.dumps({"layer": "3D %d->%d on [%d,%d,%d]" % (c, c, d, h, w), "ms_by_flags": ms}), flush=True)


if __name__ == "__main__":
    main()
