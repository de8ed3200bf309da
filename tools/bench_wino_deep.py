#!/usr/bin/env python3
"""Deep staging A/B of the Winograd kernel (hooks build, ADV_WINO_DEEP=0 / 1; same bits): the 2D and 3D 3x3 layer shapes of both detector
graphs, forward, tile = auto.  One JSON line per layer.  usage: python tools/bench_wino_deep.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route  # noqa: E402
from bench_wino_tiles import LAYERS, timed  # noqa: E402

L3D = [(64, 32, 48, 96, 312), (32, 32, 48, 96, 312), (64, 64, 24, 48, 156), (32, 64, 192, 20, 304), (128, 128, 96, 10, 152), (128, 128, 48, 5, 76)]


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for b, cin, cout, h, w in LAYERS:
        x = torch.randn((b, cin, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.05
        prep = ops.Conv2dPrep(wt, 1, 1, 1)
        ms = {}
        for rep in range(2):
            for deep in ("0", "1"):
                with hooks_route(ADV_WINO_DEEP=deep):
                    ms.setdefault(deep, []).append(round(timed(lambda: ops.conv2d(x, prep, None, None, True, wino=True)), 4))
        print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), "stages": (cin + 7) // 8, "ms_deep_off": ms["0"], "ms_deep_on": ms["1"],
                          "on_over_off": round(min(ms["1"]) / min(ms["0"]), 3)}), flush=True)
    for cin, cout, d, h, w in L3D:
        x = torch.randn((1, cin, d, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=g) * 0.05
        prep = ops.Conv3dWinoPrep(wt)
        ms = {}
        for rep in range(2):
            for deep in ("0", "1"):
                with hooks_route(ADV_WINO_DEEP=deep):
                    ms.setdefault(deep, []).append(round(timed(lambda: ops.conv3d_wino(x, prep, None, relu=True)), 4))
        print(json.dumps({"layer": "3D %d->%d on [%d,%d,%d]" % (cin, cout, d, h, w), "ms_deep_off": ms["0"], "ms_deep_on": ms["1"],
                          "on_over_off": round(min(ms["1"]) / min(ms["0"]), 3)}), flush=True)


if __name__ == "__main__":
    main()
