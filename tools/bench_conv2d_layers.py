#!/usr/bin/env python3
"""Every distinct 2D convolution of the ResNet-101-FPN Stereo R-CNN-shaped step (surrogates.StereoRcnnR101 at 600x1987, both eyes as
a batch of two), timed alone in float32: forward and the backward w.r.t. the input, through torch (MIOpen / rocBLAS) and - where this
package has a kernel for the shape - through libadvengine's float32-MFMA convolutions.  One JSON line per layer shape and one summary
line per layer class (1x1 s1, 1x1 s2, 3x3 s1, 7x7 s2 ...): FLOPs (direct-convolution count, 2 x MACs), time, TFLOP/s, fraction of the
157.3 TFLOP/s float32 matrix peak, share of the step.  VERDICT r2 items 2/3: decide the 2D convolutions on evidence.
usage: python tools/bench_conv2d_layers.py [--rois 512] [--reps 10] [--hip]"""
import argparse
import json
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import surrogates  # noqa: E402

PEAK = 157.3


def layer_list(rois, dev):
    """run one forward of the R101-shaped detector with the trace on -> {shape: calls per forward}"""
    from eval_driving_safety_amd import data
    model = surrogates.StereoRcnnR101(seed=0, rois_per_image=rois).to(dev).eval()
    batch = next(iter(data.SyntheticStereo(1, "srcnn", batch=1, seed=0)))
    extra = surrogates.synthetic_srcnn_extra(batch, dev)
    surrogates.FoldedConv.trace = []
    with torch.no_grad():
        model(batch.imgL.to(dev), batch.imgR.to(dev), extra.im_info, extra.gt_boxes_left, extra.gt_boxes_right, extra.gt_boxes_merge,
              extra.gt_dim_orien, extra.gt_kpts, extra.num_boxes)
    trace, surrogates.FoldedConv.trace = surrogates.FoldedConv.trace, None
    counts = {}
    for t in trace:
        counts[t] = counts.get(t, 0) + 1
    del model
    torch.cuda.empty_cache()
    return counts


def layer_list_dsgn(pairs, dev):
    """the 2D convolutions of one forward of the DSGN-shaped graph (adapters.DsgnShapedAdapter) at ``pairs`` stereo pairs -> {shape: calls}"""
    from eval_driving_safety_amd import adapters, data
    net = adapters.DsgnShapedAdapter(dev, seed=0, hip2d=False)
    batch = next(iter(data.SyntheticStereo(pairs, "dsgn", batch=pairs, seed=0)))
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    adapters.DsgnShapedAdapter.trace2d = []
    with torch.no_grad():
        net.forward_all(x[:pairs], x[pairs:])
    trace, adapters.DsgnShapedAdapter.trace2d = adapters.DsgnShapedAdapter.trace2d, None
    counts = {}
    for (cin, cout, k, s, p, d, b, h, w) in trace:
        key = (cin, cout, k, s, p, b, h, w, d)
        counts[key] = counts.get(key, 0) + 1
    del net
    torch.cuda.empty_cache()
    return counts


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
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--hip", action="store_true", help="also time libadvengine's kernels where one exists for the shape")
    ap.add_argument("--dsgn", action="store_true", help="the 2D layers of the DSGN-shaped graph instead of the R101 Stereo R-CNN-shaped one")
    ap.add_argument("--pairs", type=int, default=1, help="--dsgn: stereo pairs per step")
    ap.add_argument("--sweep", action="store_true", help="--hip: time every tile shape of the 1x1 kernel")
    ap.add_argument("--only", default="", help="only layer classes whose name starts with this (e.g. '1x1 s1')")
    ap.add_argument("--min-gflop", type=float, default=0.0, help="skip shapes below this many GFLOP per call")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    counts = layer_list_dsgn(args.pairs, dev) if args.dsgn else {k + (1,): v for k, v in layer_list(args.rois, dev).items()}
    ops = None
    if args.hip:
        from eval_driving_safety_amd import ops
    rows = []
    for (cin, cout, k, s, p, b, h, w, dil), n in sorted(counts.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] ** 2 * kv[0][5] * kv[0][6] * kv[0][7] / kv[0][3] ** 2):
        ho, wo = (h + 2 * p - dil * (k - 1) - 1) // s + 1, (w + 2 * p - dil * (k - 1) - 1) // s + 1
        flops = 2.0 * b * cin * cout * k * k * ho * wo
        if flops < args.min_gflop * 1e9 or not ("%dx%d s%d" % (k, k, s)).startswith(args.only):
            continue
        x = torch.randn((b, cin, h, w), device=dev)
        wt = torch.randn((cout, cin, k, k), device=dev) * 0.05
        bias = torch.randn((cout,), device=dev)
        y = F.conv2d(x, wt, bias, s, p, dil)
        g = torch.randn_like(y)
        ms_f = timeit(lambda: F.conv2d(x, wt, bias, s, p, dil), args.reps)
        ms_b = timeit(lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [p, p], [dil, dil], False, [0, 0], 1, [True, False, False]), args.reps)
        row = {"layer": "%dx%d s%d%s %d->%d on [%d,%d,%d,%d]" % (k, k, s, " dil2" if dil == 2 else "", cin, cout, b, cin, h, w), "class": "%dx%d s%d" % (k, k, s), "calls_per_forward": n,
               "gflop_per_call": round(flops / 1e9, 3), "miopen_fwd_ms": round(ms_f, 4), "miopen_dgrad_ms": round(ms_b, 4),
               "miopen_fwd_tflops": round(flops / ms_f / 1e9, 1), "miopen_dgrad_tflops": round(flops / ms_b / 1e9, 1)}
        if ops is not None and ops.conv2d_supported(x, wt, s, p, dil):
            prep = ops.Conv2dPrep(wt, s, p, dil)
            ms_hf = timeit(lambda: ops.conv2d(x, prep, bias), args.reps)
            ms_hb = timeit(lambda: ops.conv2d_dgrad(g, prep, (h, w)), args.reps)
            row.update({"hip_fwd_ms": round(ms_hf, 4), "hip_dgrad_ms": round(ms_hb, 4), "hip_fwd_tflops": round(flops / ms_hf / 1e9, 1),
                        "hip_dgrad_tflops": round(flops / ms_hb / 1e9, 1)})
            if prep.has_wino:   # Winograd F(2x2,3x3) on the matrix cores (csrc/wino2d.hip); TFLOP/s in direct-convolution FLOPs, like MIOpen's
                ms_wf = timeit(lambda: ops.conv2d(x, prep, bias, wino=True), args.reps)
                ms_wb = timeit(lambda: ops.conv2d_dgrad(g, prep, (h, w), wino=True), args.reps)
                row.update({"wino_fwd_ms": round(ms_wf, 4), "wino_dgrad_ms": round(ms_wb, 4), "wino_fwd_tflops": round(flops / ms_wf / 1e9, 1),
                            "wino_dgrad_tflops": round(flops / ms_wb / 1e9, 1)})
            if args.sweep:      # every tile shape, forward and backward: what the host's choice should have been
                row["hip_fwd_ms_by_tile"] = [round(timeit(lambda t=t: ops.conv2d(x, prep, bias, tile=t), args.reps), 4) for t in range(6)]
                row["hip_dgrad_ms_by_tile"] = [round(timeit(lambda t=t: ops.conv2d_dgrad(g, prep, (h, w), tile=t), args.reps), 4) for t in range(6)]
        rows.append(row)
        print(json.dumps(row), flush=True)
        del x, wt, y, g
    classes = {}
    for r in rows:
        c = classes.setdefault(r["class"], {"gflop": 0.0, "miopen_ms": 0.0, "best_ms": 0.0})
        n = r["calls_per_forward"]
        c["gflop"] += 2 * n * r["gflop_per_call"]                                  # forward + backward w.r.t. the input
        mi = n * (r["miopen_fwd_ms"] + r["miopen_dgrad_ms"])
        c["miopen_ms"] += mi
        c["best_ms"] += n * (min(r["miopen_fwd_ms"], r.get("hip_fwd_ms", 1e9), r.get("wino_fwd_ms", 1e9)) +
                             min(r["miopen_dgrad_ms"], r.get("hip_dgrad_ms", 1e9), r.get("wino_dgrad_ms", 1e9)))
    tot_ms = sum(c["miopen_ms"] for c in classes.values())
    tot_best = sum(c["best_ms"] for c in classes.values())
    tot_gf = sum(c["gflop"] for c in classes.values())
    for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["miopen_ms"]):
        print(json.dumps({"class": name, "gflop_fwd_plus_dgrad": round(c["gflop"], 1), "miopen_ms": round(c["miopen_ms"], 3),
                          "miopen_tflops": round(c["gflop"] / c["miopen_ms"], 1), "miopen_frac_of_157TF": round(c["gflop"] / c["miopen_ms"] / PEAK, 3),
                          "share_of_conv_time": round(c["miopen_ms"] / tot_ms, 3), "best_of_both_ms": round(c["best_ms"], 3),
                          "best_tflops": round(c["gflop"] / c["best_ms"], 1)}), flush=True)
    print(json.dumps({"summary": ("all 2D convolutions (transposed ones excepted) of one DSGN-shaped forward + input-gradient backward, %d pair(s)" % args.pairs) if args.dsgn else
                      "all 2D convolutions of one R101-FPN Stereo R-CNN-shaped forward + input-gradient backward, 600x1987, both eyes, %d rois" % args.rois,
                      "gflop": round(tot_gf, 1), "miopen_ms": round(tot_ms, 2), "miopen_tflops": round(tot_gf / tot_ms, 1),
                      "miopen_frac_of_157TF": round(tot_gf / tot_ms / PEAK, 3), "best_of_both_ms": round(tot_best, 2),
                      "best_frac_of_157TF": round(tot_gf / tot_best / PEAK, 3)}), flush=True)


if __name__ == "__main__":
    main()
