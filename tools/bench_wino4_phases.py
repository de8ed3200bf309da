#!/usr/bin/env python3
"""Where the F(4x4,3x3) kernel's time goes (csrc/wino4.hip): the same layer with phases of the stage loop switched off in the
-DADV_TEST_HOOKS build (ADV_WINO4_DBG bits: 1 input transform, 2 matrix instructions, 4 input loads, 8 input commits, 16 the stage's
wait + barrier, 32 operand reads, 64 weight DMA).  Results with a phase off are wrong by construction - a timing probe.  One JSON line per setting."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route, timeit  # noqa: E402

WHAT = ((0, "full kernel"), (1, "no input transform"), (2, "no matrix instructions"), (4, "no input loads"), (64, "no weight DMA"), (8, "no input commits"),
        (12, "no input loads, no commits"), (16, "no wait + barrier"), (32, "no operand reads"), (93, "matrix instructions + operand reads only"),
        (125, "matrix instructions only"), (127, "empty stage loop (prologue + epilogue)"))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    b, cin, cout, h, w = 2, 256, 256, 150, 497
    x = torch.randn((b, cin, h, w), device=dev)
    prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3), device=dev) * 0.02)
    flops = 2.0 * b * cin * cout * 9 * h * w
    for dbg, what in WHAT:
        with hooks_route(ADV_WINO4_DBG=str(dbg)):
            ms = timeit(lambda: ops.conv_wino4(x, prep, tile=1), reps=10)
        print(json.dumps({"layer": "%d->%d on [%d,%d,%d,%d] tile 1" % (cin, cout, b, cin, h, w), "dbg": dbg, "what": what, "ms": round(ms, 4), "direct_equiv_tflops": round(flops / ms / 1e9, 1)}), flush=True)
    cin, cout, d, h, w = 32, 32, 48, 96, 312
    x = torch.randn((1, cin, d, h, w), device=dev)
    prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05)
    flops = 2.0 * cin * cout * 27 * d * h * w
    for dbg, what in WHAT:
        with hooks_route(ADV_WINO4_DBG=str(dbg)):
            ms = timeit(lambda: ops.conv_wino4(x, prep, tile=3), reps=10)
        print(json.dumps({"layer": "3D %d->%d on [%d,%d,%d] tile 3" % (cin, cout, d, h, w), "dbg": dbg, "what": what, "ms": round(ms, 4), "direct_equiv_tflops": round(flops / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
