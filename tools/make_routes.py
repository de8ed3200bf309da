#!/usr/bin/env python3
"""Generates eval_driving_safety_amd/routes_gfx950.json - the table routes.py reads - by MEASURING, on an MI355X, every convolution
layer shape of the two detector graphs (DSGN-shaped at 1, 2 and 4 pairs per step; ResNet-101-FPN Stereo R-CNN-shaped at 1 pair) with
each candidate kernel (ADV_ROUTES=measure: best of three groups of three calls, HIP events).  One rule on top of the raw winner:
torch's operator (MIOpen) must beat this package's best kernel by more than 5 % to be chosen - a near tie goes to the kernel whose
order of float operations this package documents and tests bit for bit.

usage (GPU box):  python tools/make_routes.py [--out eval_driving_safety_amd/routes_gfx950.json] [--log profiles/r04_routes_measured.jsonl]
Run it again after a kernel change; commit both files.  The table is read by every process and rank alike (no timer at run time)."""
import argparse
import json
import os
import sys

os.environ["ADV_ROUTES"] = "measure"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eval_driving_safety_amd import adapters, data, routes, surrogates  # noqa: E402

MARGIN = 1.05


def run_dsgn(pairs, dev):
    net = adapters.DsgnShapedAdapter(dev, seed=0)
    batch = next(iter(data.SyntheticStereo(pairs, "dsgn", batch=pairs, seed=0)))
    batch.extra = net.synthetic_extra(batch, seed=1)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    net.loss_and_grad(x, batch.extra)
    torch.cuda.synchronize()


def run_r101(pairs, dev, rois=512):
    surrogates.FoldedConv.impl = "auto"
    try:
        model = surrogates.StereoRcnnR101(seed=0, rois_per_image=rois).to(dev).eval()
        net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
        batch = next(iter(data.SyntheticStereo(pairs, "srcnn", batch=pairs, seed=0)))
        batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
        x = torch.cat([batch.imgL, batch.imgR]).to(dev)
        net.loss_and_grad(x, batch.extra)
        torch.cuda.synchronize()
    finally:
        surrogates.FoldedConv.impl = "miopen"


def decide(times):
    own = {k: v for k, v in times.items() if k != ""}
    best_own = min(own, key=own.get)
    if "" in times and times[""] * MARGIN < own[best_own]:
        return ""
    return best_own


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(routes.__file__), "routes_gfx950.json"))
    ap.add_argument("--log", default="gpurun_out/routes_measured.jsonl")
    ap.add_argument("--dsgn-pairs", default="1,2,4")
    args = ap.parse_args()
    routes.configure("measure")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for p in [int(v) for v in args.dsgn_pairs.split(",") if v]:
        run_dsgn(p, dev)
        torch.cuda.empty_cache()
    run_r101(1, dev)
    meas = routes.measured()
    table = {k: decide(t) for k, (_, t) in sorted(meas.items())}
    os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
    with open(args.log, "w") as f:
        for k, (raw, t) in sorted(meas.items()):
            f.write(json.dumps({"key": k, "ms_of_3_calls": {(n or "torch"): round(v, 4) for n, v in t.items()}, "fastest": raw or "torch",
                                "table": table[k] or "torch"}) + "\n")
    doc = {"arch": "gfx950", "generated_by": "tools/make_routes.py (ADV_ROUTES=measure on an MI355X)", "device": torch.cuda.get_device_name(0),
           "rule": "fastest of this package's kernels unless torch's operator is more than %d %% faster" % round((MARGIN - 1) * 100),
           "key": "direction|k|cin|cout|dilation|input shape|flags (ops.Conv2dAuto) or f3/b3|cin|cout|shape|flags (ops.Conv3dK3)",
           "routes": table}
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
        f.write("\n")
    counts = {}
    for v in table.values():
        counts[v or "torch"] = counts.get(v or "torch", 0) + 1
    print(json.dumps({"shapes": len(table), "by_route": counts, "out": args.out, "log": args.log}))


if __name__ == "__main__":
    main()
