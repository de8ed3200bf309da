#!/usr/bin/env python3
"""Winograd F(4x4,3x3) (csrc/wino4.hip) against F(2x2,3x3) (csrc/wino2d.hip) on the 3x3 / 3x3x3 layer shapes of both detector graphs, forward.
One JSON line per layer: ms and TFLOP/s in direct-convolution FLOPs.  usage: python tools/bench_wino4.py [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_wino_tiles import LAYERS, timed  # noqa: E402

L3D = [(64, 32, 48, 96, 312), (32, 32, 48, 96, 312), (64, 64, 24, 48, 156), (32, 64, 192, 20, 304), (128, 128, 96, 10, 152), (128, 128, 48, 5, 76)]


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    layers = LAYERS if "--quick" not in sys.argv else LAYERS[1:3] + LAYERS[14:15]
    for b, cin, cout, h, w in layers:
        x = torch.randn((b, cin, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.05
        bias = torch.randn((cout,), device=dev, generator=g)
        p2, p4 = ops.Conv2dPrep(wt, 1, 1, 1), ops.ConvWino4Prep(wt)
        flops = 2.0 * b * cout * cin * 9 * h * w
        m2 = timed(lambda: ops.conv2d(x, p2, bias, None, True, wino=True))
        m4 = {t: round(timed(lambda t=t: ops.conv_wino4(x, p4, bias, None, True, tile=t)), 4) for t in ((0, 1, 2, 3) + ((4,) if w <= 15 and b >= 2 else ()))}
        m4["auto"] = round(timed(lambda: ops.conv_wino4(x, p4, bias, None, True)), 4)
        err = float((ops.conv_wino4(x, p4, bias, None, True) - ops.conv2d(x, p2, bias, None, True, wino=True)).abs().max())
        print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), "wino2_ms": round(m2, 4), "wino4_ms_by_tile": m4,
                          "wino2_tflops": round(flops / m2 / 1e9, 1), "wino4_tflops": round(flops / min(m4.values()) / 1e9, 1),
                          "speedup": round(m2 / min(m4.values()), 3), "max_abs_diff": err}), flush=True)
    for cin, cout, d, h, w in L3D:
        x = torch.randn((1, cin, d, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=g) * 0.05
        p2, p4 = ops.Conv3dWinoPrep(wt), ops.ConvWino4Prep(wt)
        flops = 2.0 * cout * cin * 27 * d * h * w
        m2 = timed(lambda: ops.conv3d_wino(x, p2, None, relu=True))
        m4 = {t: round(timed(lambda t=t: ops.conv_wino4(x, p4, None, None, True, tile=t)), 4) for t in (0, 1, 2, 3)}
        print(json.dumps({"layer": "3D %d->%d on [%d,%d,%d]" % (cin, cout, d, h, w), "wino2_ms": round(m2, 4), "wino4_ms_by_tile": m4,
                          "wino2_tflops": round(flops / m2 / 1e9, 1), "wino4_tflops": round(flops / min(m4.values()) / 1e9, 1),
                          "speedup": round(m2 / min(m4.values()), 3)}), flush=True)


if __name__ == "__main__":
    main()
