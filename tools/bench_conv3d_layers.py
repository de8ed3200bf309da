#!/usr/bin/env python3
"""Every 3D convolution of the DSGN-shaped step (adapters.DsgnShapedAdapter: plane-sweep volume [64,48,96,312] and 3D geometric volume
[32..128,192,20,304]) timed alone, forward and backward w.r.t. the input, on the kernel the graph runs for it (stride-1 layers: the route
table's pick of direct / Winograd, the direct kernel's time beside it; strided / transposed: the direct float32-MFMA kernels): GFLOP, ms, TFLOP/s
in direct-convolution FLOPs, fraction of the 157.3 TFLOP/s float32 matrix peak.  usage: python tools/bench_conv3d_layers.py [--pairs B] [--reps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops, routes  # noqa: E402

PSV, GV = (48, 96, 312), (192, 20, 304)
LAYERS = [  # name, kind, cin, cout, input dims
    ("dres0a", "s1", 64, 32, PSV), ("dres0b/dres1a/dres1b/cls_a", "s1", 32, 32, PSV), ("cls_b", "s1", 32, 1, PSV),
    ("hg1", "s2", 32, 64, PSV), ("hg2", "s1", 64, 64, (24, 48, 156)), ("hg3", "s2", 64, 64, (24, 48, 156)), ("hg4", "s1", 64, 64, (12, 24, 78)),
    ("hg5", "t2", 64, 64, (12, 24, 78)), ("hg6", "t2", 64, 32, (24, 48, 156)),
    ("gv1", "s1", 32, 64, GV), ("gh1", "s2", 64, 128, GV), ("gh2", "s1", 128, 128, (96, 10, 152)), ("gh3", "s2", 128, 128, (96, 10, 152)),
    ("gh4", "s1", 128, 128, (48, 5, 76)), ("gh5", "t2", 128, 128, (48, 5, 76)), ("gh6", "t2", 128, 64, (96, 10, 152)),
]


def timeit(fn, reps):
    for _ in range(2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    b = args.pairs
    tot_f, tot_ms = 0.0, 0.0
    for name, kind, cin, cout, (d, h, w) in LAYERS:
        x = torch.randn((b, cin, d, h, w), device=dev)
        bias = torch.randn((cout,), device=dev) * 0.1
        if kind == "t2":
            wt = torch.randn((cin, cout, 3, 3, 3), device=dev) * 0.05
            fwd_p, bwd_p = ops.conv_transpose3d_k3_s2_prep(wt), ops.conv3d_k3_s2_prep(wt)
            fwd = lambda: ops.conv_transpose3d_k3_s2(x, fwd_p, cout, relu=True, bias=bias)               # noqa: E731
            y = fwd()
            bwd = lambda: ops.conv3d_k3_s2(y, bwd_p, cin)                                                # noqa: E731
            flops = 2.0 * b * cin * cout * 27 * d * h * w
        elif kind == "s2":
            wt = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
            fwd_p, bwd_p = ops.conv3d_k3_s2_prep(wt), ops.conv_transpose3d_k3_s2_prep(wt)
            fwd = lambda: ops.conv3d_k3_s2(x, fwd_p, cout, relu=True, bias=bias)                         # noqa: E731
            y = fwd()
            bwd = lambda: ops.conv_transpose3d_k3_s2(y, bwd_p, cin)                                      # noqa: E731
            flops = 2.0 * b * cin * cout * 27 * y.shape[2] * y.shape[3] * y.shape[4]
        else:
            direct = None
            wt = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
            fwd_p, bwd_p = ops.conv3d_k3_prep(wt), ops.conv3d_k3_prep(wt, transpose=True)
            fwd = (lambda: ops.conv3d_k3(x, fwd_p, cout)) if cout < 4 else (lambda: ops.conv3d_k3(x, fwd_p, cout, relu=True, bias=bias))
            y = fwd()
            bwd = lambda: ops.conv3d_k3(y, bwd_p, cin)                                                   # noqa: E731
            flops = 2.0 * b * cin * cout * 27 * d * h * w
            direct = None
            if cout >= 4:       # <round 4> what the GRAPH runs for this layer: the route table's pick (ops.Conv3dK3's keys), direct kernel beside it
                wino = ops.Conv3dWinoPrep(wt)
                routes.mode()
                table = routes._state["table"] or {}

                def pick(keys):         # the graph's key carries epilogue flags (skip connection, ReLU, mask): the first variant the table holds
                    for k in keys:
                        r = table.get(routes.key_str(k))
                        if r:
                            return r
                    return routes.fixed_rule({"direct": None, "wino": None})
                shp = tuple(x.shape)
                rf = pick([("f3", cin, cout, shp, False, True), ("f3", cin, cout, shp, True, False), ("f3", cin, cout, shp, False, False), ("f3", cin, cout, shp, True, True)])
                rb = pick([("b3", cin, cout, shp, True), ("b3", cin, cout, shp, False)])
                direct = (timeit(fwd, args.reps), timeit(bwd, args.reps), rf, rb)
                if rf == "wino":
                    fwd = lambda: ops.conv3d_wino(x, wino, bias, None, True)                             # noqa: E731
                if rb == "wino":
                    bwd = lambda: ops.conv3d_wino_dgrad(y, wino)                                         # noqa: E731
        ms_f, ms_b = timeit(fwd, args.reps), timeit(bwd, args.reps)
        extra = {}
        if kind == "s1" and cout >= 4 and direct is not None:
            extra = {"route_fwd": direct[2], "route_bwd": direct[3], "direct_kernel_fwd_ms": round(direct[0], 4), "direct_kernel_bwd_ms": round(direct[1], 4)}
        n = 4 if name.startswith("dres0b") else 1
        tot_f += 2 * n * flops
        tot_ms += n * (ms_f + ms_b)
        print(json.dumps({"layer": name, "kind": {"s1": "stride 1", "s2": "stride 2", "t2": "transposed stride 2"}[kind], "cin": cin, "cout": cout,
                          "input_dhw": [d, h, w], "pairs": b, "gflop": round(flops / 1e9, 2), "fwd_ms": round(ms_f, 4), "bwd_ms": round(ms_b, 4),
                          "fwd_tflops": round(flops / ms_f / 1e9, 1), "bwd_tflops": round(flops / ms_b / 1e9, 1),
                          "fwd_frac_of_157TF": round(flops / ms_f / 1e9 / 157.3, 3), "bwd_frac_of_157TF": round(flops / ms_b / 1e9 / 157.3, 3),
                          "calls_per_step": n, **extra}), flush=True)
        del x, y
    print(json.dumps({"summary": "all 3D convolutions of one DSGN-shaped step (forward + backward w.r.t. the input), %d pair(s)" % b,
                      "gflop": round(tot_f / 1e9, 1), "ms": round(tot_ms, 3), "tflops": round(tot_f / tot_ms / 1e9, 1),
                      "frac_of_157TF": round(tot_f / tot_ms / 1e9 / 157.3, 3)}), flush=True)


if __name__ == "__main__":
    main()
