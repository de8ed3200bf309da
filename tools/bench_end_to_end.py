#!/usr/bin/env python3
"""End-to-end 20-step PGD through a DSGN-SHAPED plane-sweep depth network (adapters.PsvStereoAdapter:
seeded random weights, 2D features -> HIP cost volume -> 3 x conv3d (MIOpen) -> soft-argmin depth ->
smooth-L1), reported SEPARATELY from bench.py's perturbation-path number (never conflate them).
Prints one JSON line.  usage: python tools/bench_end_to_end.py [--pairs B] [--iters 20] [--reps 2]"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import adapters, attacks, data  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    net = adapters.PsvStereoAdapter(dev, seed=0)
    batch = next(iter(data.SyntheticStereo(args.pairs, "dsgn", batch=args.pairs, seed=0)))
    gen = torch.Generator().manual_seed(1)
    gt = torch.rand((args.pairs, 384, 1248), generator=gen) * 38.4 + 2.0
    gt = torch.where(torch.rand((args.pairs, 384, 1248), generator=gen) < 0.05, gt, torch.zeros(()))
    batch.extra = types.SimpleNamespace(disp_true=gt.to(dev))
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, args.iters, save=False, device=dev)
    atk.run_batch(batch, net)                         # warm-up (MIOpen solver search, allocator)
    torch.cuda.synchronize()
    # split: detector fwd+bwd alone vs the whole step
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        net.loss_and_grad(x, batch.extra)
    e1.record()
    torch.cuda.synchronize()
    model_ms = e0.elapsed_time(e1) / 3
    t0 = time.perf_counter()
    for _ in range(args.reps):
        atk.run_batch(batch, net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    first, last = float(atk.last_losses[0]), float(atk.last_losses[-1])
    print(json.dumps({
        "metric": "end-to-end stereo-pairs/s, %d-step PGD through a DSGN-shaped plane-sweep depth net (surrogate, random weights)" % args.iters,
        "value": args.pairs / dt, "unit": "stereo-pairs/s", "pairs": args.pairs, "iters": args.iters,
        "s_per_attack": dt, "detector_fwd_bwd_ms": model_ms,
        "perturbation_share": max(0.0, 1.0 - args.iters * model_ms * 1e-3 / dt),
        "loss_first_iter": first, "loss_last_iter": last, "loss_rose": last > first,
        "note": "dtype f32; conv stacks by MIOpen via torch, cost volume + PGD step by libadvengine.so"}))


if __name__ == "__main__":
    main()
