#!/usr/bin/env python3
"""Where the Winograd kernel's time goes (csrc/wino2d.hip): the same layer with phases of the stage loop switched off in the
-DADV_TEST_HOOKS build (ADV_WINO_DBG bits: 1 input transform, 2 matrix instructions, 4 global fetches, 8 LDS commits, 16 barrier).
Results with a phase off are wrong by construction - this is a timing probe.  One JSON line per setting."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route, timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for (b, cin, cout, h, w) in ((2, 256, 256, 150, 497), (2, 64, 64, 96, 312)):
        x = torch.randn((b, cin, h, w), device=dev)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
        prep = ops.Conv2dPrep(wt, 1, 1, 1)
        flops = 2.0 * b * cin * cout * 9 * h * w
        for dbg, what in ((0, "full kernel"), (1, "no input transform"), (2, "no matrix instructions"), (4, "no global fetches"), (8, "no LDS commits"),
                          (16, "no barrier"), (3, "no transform, no matrix instructions"), (13, "only the matrix instructions + barrier"),
                          (29, "only the matrix instructions"), (31, "empty stage loop (prologue + epilogue)")):
            with hooks_route(ADV_WINO_DBG=str(dbg)):
                ms = timeit(lambda: ops.conv2d(x, prep, wino=True), reps=10)
            print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), "dbg": dbg, "what": what, "ms": round(ms, 4),
                              "direct_equiv_tflops": round(flops / ms / 1e9, 1)}), flush=True)


def main3d():
    """the 32 -> 32 layer of the cost-volume network (8 x 32 x 32-channel workgroups, stages of 4 q, three depth taps)"""
    dev = torch.device("cuda", 0)
    b, cin, cout, d, h, w = 1, 32, 32, 48, 96, 312
    x = torch.randn((b, cin, d, h, w), device=dev)
    wt = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
    prep = ops.Conv3dWinoPrep(wt)
    flops = 2.0 * b * cin * cout * 27 * d * h * w
    for dbg, what in ((0, "full kernel"), (1, "no input transform"), (2, "no matrix instructions"), (4, "no global fetches"), (8, "no LDS commits"),
                      (16, "no barrier"), (3, "no transform, no matrix instructions"), (13, "only the matrix instructions + barrier"),
                      (29, "only the matrix instructions"), (31, "empty stage loop (prologue + epilogue)")):
        with hooks_route(ADV_WINO_DBG=str(dbg)):
            ms = timeit(lambda: ops.conv3d_wino(x, prep, None, relu=False, tile=2), reps=10)
        print(json.dumps({"layer": "3D %d->%d on [%d,%d,%d,%d,%d]" % (cin, cout, b, cin, d, h, w), "dbg": dbg, "what": what, "ms": round(ms, 4),
                          "direct_equiv_tflops": round(flops / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    if "--3d" in sys.argv:
        main3d()
        sys.exit(0)
    main()
