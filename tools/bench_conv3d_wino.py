#!/usr/bin/env python3
"""The stride-1 3x3x3 layers of the DSGN-shaped step through the direct float32-MFMA kernel (csrc/conv3d.hip) and through the Winograd
kernel (csrc/wino2d.hip, transform in the (H, W) plane, depth taps inside the contraction), forward and backward w.r.t. the input.
TFLOP/s in direct-convolution FLOPs (2 x 27 x Cin x Cout x voxels).  usage: python tools/bench_conv3d_wino.py [--pairs B] [--tiles]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv3d_layers import LAYERS, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tiles", action="store_true", help="time every workgroup shape of the Winograd kernel")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    b = args.pairs
    tot = {"direct": 0.0, "wino": 0.0, "best": 0.0, "gflop": 0.0}
    for name, kind, cin, cout, (d, h, w) in LAYERS:
        if kind != "s1" or cout < 4:
            continue
        x = torch.randn((b, cin, d, h, w), device=dev)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
        bias = torch.randn((cout,), device=dev) * 0.1
        wp, wpt, prep = ops.conv3d_k3_prep(wt), ops.conv3d_k3_prep(wt, transpose=True), ops.Conv3dWinoPrep(wt)
        y = ops.conv3d_k3(x, wp, cout, relu=True, bias=bias)
        g = torch.randn_like(y)
        flops = 2.0 * b * cin * cout * 27 * d * h * w
        row = {"layer": name, "cin": cin, "cout": cout, "input_dhw": [d, h, w], "pairs": b, "gflop": round(flops / 1e9, 2)}
        for tag, fd, fw in (("fwd", lambda: ops.conv3d_k3(x, wp, cout, relu=True, bias=bias), lambda t=-1: ops.conv3d_wino(x, prep, bias, relu=True, tile=t)),
                            ("bwd", lambda: ops.conv3d_k3(g, wpt, cin), lambda t=-1: ops.conv3d_wino_dgrad(g, prep, tile=t))):
            md, mw = timeit(fd, args.reps), timeit(fw, args.reps)
            row.update({tag + "_direct_ms": round(md, 4), tag + "_wino_ms": round(mw, 4), tag + "_direct_tflops": round(flops / md / 1e9, 1),
                        tag + "_wino_tflops": round(flops / mw / 1e9, 1)})
            if args.tiles:
                row[tag + "_wino_ms_by_tile"] = [round(timeit(lambda t=t: fw(t), args.reps), 4) for t in range(6)]
            tot["direct"] += md
            tot["wino"] += mw
            tot["best"] += min(md, mw)
            tot["gflop"] += flops / 1e9
        print(json.dumps(row), flush=True)
        del x, y, g
    print(json.dumps({"summary": "stride-1 3x3x3 layers of one DSGN-shaped step, forward + backward w.r.t. the input, %d pair(s)" % b,
                      "gflop": round(tot["gflop"], 1), "direct_ms": round(tot["direct"], 3), "wino_ms": round(tot["wino"], 3),
                      "best_ms": round(tot["best"], 3), "direct_frac_of_157TF": round(tot["gflop"] / tot["direct"] / 157.3, 3),
                      "wino_frac_of_157TF": round(tot["gflop"] / tot["wino"] / 157.3, 3)}), flush=True)


if __name__ == "__main__":
    main()
