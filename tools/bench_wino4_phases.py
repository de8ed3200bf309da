#!/usr/bin/env python3
"""Where the F(4x4,3x3) kernel's time goes (csrc/wino4.hip): the same layer with phases of the stage loop switched off in the
-DADV_TEST_HOOKS build (ADV_WINO4_DBG bits: 1 input transform, 2 matrix instructions, 4 input loads, 8 input commits, 16 the stage's
wait + barrier, 32 operand reads, 64 weight loads).  Results with a phase off are wrong by construction - a timing probe.  One JSON line per setting."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route, timeit  # noqa: E402

WHAT = ((0, "full kernel"), (1, "no input transform"), (2, "no matrix instructions"), (4, "no input-tile DMA"), (64, "no weight loads"), (8, "no edge fix-up"),
        (12, "no input-tile DMA, no edge fix-up"), (16, "no wait + barrier"), (32, "no operand reads"), (93, "matrix instructions + operand reads only"),
        (125, "matrix instructions only"), (127, "empty stage loop (prologue + epilogue)"))


CASES = (("2d", "256->256 on [2,256,150,497] tile 1"), ("3d", "3D 32->32 on [48,96,312] tile 3"))


def one(case, dbg):
    """a single (layer, ablation) measurement: its own process, because an ablated kernel computes garbage by construction and one
    variant (operand reads off, 3D shape) has been seen to take the process down"""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if case == "2d":
        b, cin, cout, h, w = 2, 256, 256, 150, 497
        x = torch.randn((b, cin, h, w), device=dev)
        prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3), device=dev) * 0.02)
        flops, tile = 2.0 * b * cin * cout * 9 * h * w, 1
    else:
        cin, cout, d, h, w = 32, 32, 48, 96, 312
        x = torch.randn((1, cin, d, h, w), device=dev)
        prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05)
        flops, tile = 2.0 * cin * cout * 27 * d * h * w, 3
    with hooks_route(ADV_WINO4_DBG=str(dbg)):
        ms = timeit(lambda: ops.conv_wino4(x, prep, tile=tile), reps=10)
    torch.cuda.synchronize()
    print(json.dumps({"ms": round(ms, 4), "direct_equiv_tflops": round(flops / ms / 1e9, 1)}), flush=True)


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--one":
        return one(sys.argv[2], int(sys.argv[3]))
    import subprocess
    for case, layer in CASES:
        for dbg, what in WHAT:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", case, str(dbg)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            row = {"layer": layer, "dbg": dbg, "what": what}
            row.update(json.loads(lines[-1]) if r.returncode == 0 and lines else {"ms": None, "error": "the process ended with code %d" % r.returncode})
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
