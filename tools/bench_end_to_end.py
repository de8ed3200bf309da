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


def measure(pairs=1, iters=20, reps=2, warm_iters=None, mfma_conv=True, hourglass=False, dsgn_head=False):
    """-> dict; the first (untimed) attack absorbs MIOpen's solver search"""
    dev = torch.device("cuda", torch.cuda.current_device())
    net = adapters.PsvStereoAdapter(dev, seed=0, mfma_conv=mfma_conv, hourglass=hourglass, dsgn_head=dsgn_head)
    batch = next(iter(data.SyntheticStereo(pairs, "dsgn", batch=pairs, seed=0)))
    gen = torch.Generator().manual_seed(1)
    gt = torch.rand((pairs, 384, 1248), generator=gen) * 38.4 + 2.0
    gt = torch.where(torch.rand((pairs, 384, 1248), generator=gen) < 0.05, gt, torch.zeros(()))
    batch.extra = types.SimpleNamespace(disp_true=gt.to(dev))
    if dsgn_head:
        batch.extra.boxes = net.synthetic_extra(batch, seed=1).boxes
    warm = attacks.PgdAttack("dsgn", 1 / 255, 0.03, warm_iters if warm_iters else iters, save=False, device=dev)
    warm.run_batch(batch, net)
    torch.cuda.synchronize()
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, iters, save=False, device=dev)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    for _ in range(3):
        net.loss_and_grad(x, batch.extra)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.loss_and_grad(x, batch.extra)
    e1.record()
    torch.cuda.synchronize()
    model_ms = e0.elapsed_time(e1) / 10
    t0 = time.perf_counter()
    for _ in range(reps):
        atk.run_batch(batch, net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    first, last = float(atk.last_losses[0]), float(atk.last_losses[-1])
    mfma = None
    if mfma_conv:   # the dominant kernel of the end-to-end path, timed alone on the cost-volume shape: MFMA roofline
        from eval_driving_safety_amd import ops
        xv = torch.randn((pairs, 64, 48, 96, 312), device=dev)
        for _ in range(2):
            ops.conv3d_k3(xv, net.p1, 32)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            ops.conv3d_k3(xv, net.p1, 32)
        c1.record()
        torch.cuda.synchronize()
        ms = c0.elapsed_time(c1) / 10
        flops = 2.0 * 64 * 32 * 27 * pairs * 48 * 96 * 312
        mfma = {"bound": "mfma", "kernel": "conv3d_k3_mfma<2> 64->32 on [%d,64,48,96,312]" % pairs, "achieved": flops / ms / 1e9,
                "peak": 157.3, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / 157.3, "avg_launch_ms": ms,
                "note": "float32 matrix peak (MI355X_MICROARCH.md); where the rest goes: profiles/r02_conv3d_where_the_time_goes.md (tile-count tail 6 %, staging issue 9 %, epilogue 2-6 %, width padding 2.5 %)"}
        del xv
    return {
        "metric": "end-to-end stereo-pairs/s, %d-step PGD through a DSGN-shaped plane-sweep depth net (surrogate, random weights)" % iters,
        "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "iters": iters,
        "s_per_attack": dt, "detector_fwd_bwd_ms": model_ms,
        "loss_first_iter": first, "loss_last_iter": last, "loss_rose": last > first, "roofline": mfma,
        "convs": ("libadvengine float32-MFMA conv3d (all 3D layers, forward and adjoint; the 32->1 score layer on the narrow vector-ALU kernels)" if mfma_conv else "torch / MIOpen"),
        "volume_net": ("3D hourglass: 64->32, 32->32, 32->64 /2, 64->64, 64->64 /2, 64->64, transposed 64->64 + skip, transposed 64->32 + skip, 32->1"
                       if hourglass else "three convolutions 64->32->32->1"),
        "head": ("DSGN-shaped: fused trilinear-upsample+softmax+expectation depth over 192 planes (ops.DepthRegress), plane-sweep features x "
                 "plane probability -> grid_sample into a [B,32,192,20,304] 3D geometric volume (ops.GridSample3d, deterministic gather "
                 "backward), conv3d, bird's-eye-view 2D convolutions, focal + smooth-L1 + BCE detection loss" if dsgn_head
                 else "soft-argmin depth at 1/4 resolution + bilinear up-sampling, smooth-L1"),
        "note": "NOT the headline metric and NOT DSGN: 2D features -> HIP plane-sweep volume [B,64,48,96,312] (fractional disparities) -> "
                "3D convolutions -> depth (and detection) head -> loss; dtype f32; cost volume + convolutions + head kernels + PGD step by "
                "libadvengine.so"}


def _choice_summary(fresh=False):
    """which kernel computed the convolutions of a leg: the decisions come from routes.py (the committed table by default)"""
    from eval_driving_safety_amd import routes
    if fresh:                                   # a new leg: count its own decisions
        routes._state["used"].clear()
        routes._state["misses"].clear()
        return None
    c = {tuple(k.split("|")): v for k, v in routes.used().items()}

    def side(d):
        own = [v for k, v in c.items() if k[0] == d]
        return "%d of %d layer shapes (%d of them by the Winograd kernel)" % (sum(1 for v in own if v), len(own), sum(1 for v in own if v == "wino"))
    out = {"forward": side("f"), "backward": side("b"), "routes": routes.summary()}
    three = {d: [v for k, v in c.items() if k[0] == d] for d in ("f3", "b3")}
    if three["f3"] or three["b3"]:       # stride-1 3x3x3 layers: direct float32-MFMA kernel or the Winograd kernel
        out["stride1_3d_layers_on_the_winograd_kernel"] = {"forward": "%d of %d layer shapes" % (sum(v == "wino" for v in three["f3"]), len(three["f3"])),
                                                            "backward": "%d of %d layer shapes" % (sum(v == "wino" for v in three["b3"]), len(three["b3"]))}
    return out


def _executed(step, wino_equiv, executed, model_ms):
    """the matrix work the step really EXECUTES: Winograd layers at their 2.25x / 4x lower multiply count.  ``mfma_util`` = executed FLOPs /
    time / peak is an upper bound of the matrix pipes' busy share over the whole step (padding and the element-wise / loss kernels
    included in the time); the counters of the individual kernels are in profiles/r06_conv_pmc.json"""
    return {"executed_flops_per_step": executed, "winograd_share_of_direct_flops": wino_equiv / step if step else 0.0,
            "mfma_util": executed / model_ms / 1e9 / 157.3,
            "mfma_util_note": "executed multiply-add FLOPs (Winograd F(2x2) routes at 1/2.25, F(4x4) routes at 1/4 of their direct count) / step time / 157.3 TFLOP/s"}


def _roofline(what, step, executed, ms):
    """``frac`` = the EXECUTED multiply-add FLOPs over the float32 matrix peak (<= 1 by construction); the direct-convolution FLOPs a Winograd
    route replaces are credited beside it as ``direct_equivalent_*`` (an algorithmic figure that can exceed 1: not a roofline fraction)"""
    return {"bound": "mfma", "what": what, "achieved": executed / ms / 1e9, "peak": 157.3, "unit": "TFLOP/s", "frac": executed / ms / 1e9 / 157.3,
            "direct_equivalent_tflops": step / ms / 1e9, "direct_equivalent_frac": step / ms / 1e9 / 157.3,
            "note": "frac counts what the matrix pipes executed; on gfx950 a float32 matrix instruction and the other wave's float32 vector instructions "
                    "do not run at the same time on a SIMD (profiles/r06_mfma_valu_probe.json), so transforms / epilogues lower this figure by construction"}


def measure_dsgn_full(pairs=1, iters=20, reps=1, graph=False, hip2d="auto"):
    """BASELINE configs[1] end to end through the DSGN-shaped graph with SURVEY App. B's layer list (adapters.DsgnShapedAdapter: PSMNet-style
    2D extractor, plane-sweep volume, dres0/dres1 + 3D hourglass, fused depth regression, 3D geometric volume + 64-channel stack + 3D
    hourglass, bird's-eye-view 2D hourglass, head towers): exact FLOPs per detector step from the layer list and the WHOLE step
    against the float32 matrix peak - at ``pairs`` stereo pairs per step (the reference runs 1; 288 GB hold more)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    _choice_summary(fresh=True)
    net = adapters.DsgnShapedAdapter(dev, seed=0, hip2d=hip2d)
    batch = next(iter(data.SyntheticStereo(pairs, "dsgn", batch=pairs, seed=0)))
    batch.extra = net.synthetic_extra(batch, seed=1)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    for _ in range(2):
        net.loss_and_grad(x, batch.extra)
    torch.cuda.synchronize()
    if os.environ.get("ADV_STOP_AFTER_WARMUP") == "1":      # tools/gpu_profile_step.sh: the warm-up alone (MIOpen's solver search), to be subtracted
        return {"stopped": "after the warm-up"}
    from eval_driving_safety_amd import ops as _ops
    _ops.WINO_DIRECT_EQUIV_FLOPS[0] = _ops.WINO4_DIRECT_EQUIV_FLOPS[0] = 0
    step = float(net.flops_per_step(x, batch.extra))
    wino_equiv = float(_ops.WINO_DIRECT_EQUIV_FLOPS[0])       # direct-equivalent FLOPs of the calls that took a Winograd route, one step
    wino4_equiv = float(_ops.WINO4_DIRECT_EQUIV_FLOPS[0])     # ... and of those that took the F(4x4,3x3) route (4x fewer multiply-adds executed)
    executed = step - wino_equiv * (1.0 - 1.0 / 2.25) - wino4_equiv * 0.75
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        net.loss_and_grad(x, batch.extra)
    e1.record()
    torch.cuda.synchronize()
    model_ms = e0.elapsed_time(e1) / n
    atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, iters, save=False, device=dev, graph=graph)
    if graph:                                      # the capture itself (2 warm-up iterations + instantiate) is paid once per batch shape
        atk.run_batch(batch, net)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        atk.run_batch(batch, net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    first, last = float(atk.last_losses[0]), float(atk.last_losses[-1])
    if graph:       # what the eager loop loses between kernels: the replayed iteration against the eager detector step + PGD step
        g = atk.last_graph
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        replay_ms = e0.elapsed_time(e1) / n
        return {"metric": "the same %d-step attack with ONE iteration (detector forward + backward + fused PGD step) captured in a hipGraph" % iters,
                "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "iters": iters, "s_per_attack_incl_capture": dt,
                "graph_replay_ms_per_iteration": replay_ms, "eager_detector_fwd_bwd_ms": model_ms,
                "launch_gap_share_of_eager_step": max(0.0, 1.0 - replay_ms / model_ms),
                "note": "s_per_attack includes the per-batch capture (2 eager warm-up iterations + graph instantiation); replay vs eager is the launch-gap share"}
    return {"metric": "end-to-end stereo-pairs/s, %d-step PGD through the DSGN-shaped graph with SURVEY App. B's layer list (surrogate, random weights)" % iters,
            "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "iters": iters, "s_per_attack": dt, "detector_fwd_bwd_ms": model_ms,
            "flops_per_step": step, "flops_per_step_per_pair": step / pairs,
            "roofline": _roofline("WHOLE detector step (forward + input-gradient backward; everything 3D by libadvengine - stride-1 layers by its direct or its "
                                  "Winograd kernels -, 2D layers by libadvengine or MIOpen, each as the committed route table says, element-wise and loss kernels "
                                  "included): executed multiply-add FLOPs against the float32 matrix peak", step, executed, model_ms),
            "convolutions_2d": {"auto": "per layer shape and direction whichever of {libadvengine direct float32-MFMA 1x1 / 3x3 kernel, libadvengine Winograd "
                                        "F(2x2,3x3) kernel on the matrix cores, MIOpen} the committed route table (routes_gfx950.json) names - all with fused epilogues; the dilation-2 "
                                        "blocks run as dilation-1 blocks on the four parity sub-images", True: "libadvengine for every 1x1 / 3x3 stride-1 layer",
                                False: "torch / MIOpen"}[hip2d],
            "layers_2d_on_libadvengine": _choice_summary(),
            "peak_hbm_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
            **_executed(step, wino_equiv + wino4_equiv, executed, model_ms),
            "loss_first_iter": first, "loss_last_iter": last, "loss_rose": last > first,
            "note": "NOT the headline metric and NOT DSGN's weights; layer list [UPSTREAM-UNVERIFIED] from the published PSMNet / DSGN structures"}


def _graph_leg_in_child(pairs, iters, rois, fwd_flops):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--r101-graph-leg", "--pairs", str(pairs), "--iters", str(iters), "--rois", str(rois)]
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)      # (normally ~40 s)
    except subprocess.TimeoutExpired:
        return {"error": "the child process did not finish in 300 s"}
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        return {"error": "child process ended with code %d" % res.returncode, "stderr_tail": res.stderr[-300:]}
    out = json.loads(lines[-1])
    out["roofline"] = {"bound": "mfma", "what": "ONE replayed iteration - detector forward + backward AND the fused PGD step - in direct-convolution FLOPs (the executed "
                                                "fraction: the eager line's roofline.frac scaled by the two times)",
                       "direct_equivalent_tflops": 2.0 * fwd_flops / out["graph_replay_ms_per_iteration"] / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                       "direct_equivalent_frac": 2.0 * fwd_flops / out["graph_replay_ms_per_iteration"] / 1e9 / 157.3}
    return out


def graph_leg_r101(pairs=1, iters=20, rois=512):
    """the child's side of _graph_leg_in_child: capture, one reused attack (timed), replay timing"""
    from eval_driving_safety_amd import adapters, attacks, data, surrogates
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    surrogates.FoldedConv.impl = "auto"
    model = surrogates.StereoRcnnR101(seed=0, rois_per_image=rois).to(dev).eval()
    model.allow_graph_capture = True              # opt-in: see surrogates.StereoRcnnShaped.allow_graph_capture
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
    batch = next(iter(data.SyntheticStereo(pairs, "srcnn", batch=pairs, seed=0)))
    batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
    atk = attacks.PgdAttack("srcnn", 1.0, 0.03, iters, save=False, device=dev, graph=True)
    atk.run_batch(batch, net)                     # pays the capture (two eager warm-up iterations + instantiation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    atk.run_batch(batch, net)                     # the next batch with the same label set: the capture is reused
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g = atk.last_graph
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return {"metric": "the same %d-step attack with ONE iteration (detector forward + backward + fused PGD step) captured in a hipGraph, capture reused" % iters,
            "value": pairs / dt, "unit": "stereo-pairs/s", "s_per_attack": dt, "graph_replay_ms_per_iteration": e0.elapsed_time(e1) / 10,
            "captures_reused": getattr(atk, "graph_captures_reused", 0),
            "loss_first_iter": float(atk.last_losses[0]), "loss_last_iter": float(atk.last_losses[-1])}


def measure_srcnn_r101(pairs=1, iters=20, reps=1, rois=512, impl="auto", graph_backbone=False, graph_leg=True):
    """BASELINE configs[2] with the upstream LAYER LIST (surrogates.StereoRcnnR101: ResNet-101 [3,4,23,3] + FPN P2-P6 + stereo RPN + RoI
    heads, random weights, batch-norms folded): 20-step PGD at 600x1987, exact FLOPs per detector step from the layer list, the
    whole-step rate against the float32 matrix peak.  ``impl``: "miopen" = every 2D convolution through torch (MIOpen / rocBLAS),
    "hip" = libadvengine's float32-MFMA kernels where one exists for the shape."""
    from eval_driving_safety_amd import surrogates
    dev = torch.device("cuda", torch.cuda.current_device())
    surrogates.FoldedConv.impl = impl
    _choice_summary(fresh=True)
    try:
        model = surrogates.StereoRcnnR101(seed=0, rois_per_image=rois).to(dev).eval()
        net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
        batch = next(iter(data.SyntheticStereo(pairs, "srcnn", batch=pairs, seed=0)))
        batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
        x = torch.cat([batch.imgL, batch.imgR]).to(dev)
        for _ in range(2):                       # MIOpen's solver search, kernel loads
            net.loss_and_grad(x, batch.extra)
        torch.cuda.synchronize()
        if os.environ.get("ADV_STOP_AFTER_WARMUP") == "1":
            return {"stopped": "after the warm-up"}
        model.reset_flops()
        from eval_driving_safety_amd import ops as _ops
        _ops.WINO_DIRECT_EQUIV_FLOPS[0] = _ops.WINO4_DIRECT_EQUIV_FLOPS[0] = 0
        net.loss_and_grad(x, batch.extra)        # the FLOP count comes from the eager graph (a replayed hipGraph runs no Python)
        wino_equiv = float(_ops.WINO_DIRECT_EQUIV_FLOPS[0])
        wino4_equiv = float(_ops.WINO4_DIRECT_EQUIV_FLOPS[0])
        by_class = model.flops_by_class()
        fwd = float(sum(by_class.values()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net.loss_and_grad(x, batch.extra)
        e1.record()
        torch.cuda.synchronize()
        eager_ms = e0.elapsed_time(e1) / 3
        model.use_graph = graph_backbone
        if graph_backbone:
            net.loss_and_grad(x, batch.extra)    # captures the backbone's forward and backward graphs for this shape
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            net.loss_and_grad(x, batch.extra)
        e1.record()
        torch.cuda.synchronize()
        model_ms = e0.elapsed_time(e1) / 5
        atk = attacks.PgdAttack("srcnn", 1.0, 0.03, iters, save=False, device=dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            atk.run_batch(batch, net)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        hip_graph = None
        if graph_leg and model._static_ok(x):
            # The whole iteration (detector forward + backward + fused PGD step) captured once, reused for the next batch, replayed: no launch
            # gaps.  In a CHILD process: a fault inside a replayed graph aborts the process that launched it (this stack has produced such
            # faults), and this one must still print its line.
            hip_graph = _graph_leg_in_child(pairs, iters, rois, fwd)
    finally:
        surrogates.FoldedConv.impl = "miopen"
    step = 2.0 * fwd                               # forward + backward w.r.t. the input: every layer's adjoint costs its forward
    first, last = float(atk.last_losses[0]), float(atk.last_losses[-1])
    return {"metric": "end-to-end stereo-pairs/s, %d-step PGD through a ResNet-101-FPN Stereo R-CNN-shaped detector (upstream layer list, random weights), 600x1987" % iters,
            "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "iters": iters, "s_per_attack": dt, "detector_fwd_bwd_ms": model_ms,
            "flops_per_step": step, "flops_fwd_by_layer_class": {k: v for k, v in by_class.items()},
            "roofline": _roofline("WHOLE detector step (forward + input-gradient backward, every kernel incl. RoIAlign, NMS, losses, element-wise): executed "
                                  "multiply-add FLOPs against the float32 matrix peak", step, step - wino_equiv * (1.0 - 1.0 / 2.25) - wino4_equiv * 0.75, model_ms),
            "convolutions": impl, "layers_2d_on_libadvengine": _choice_summary(), "rois_per_image": rois,
            **_executed(step, wino_equiv + wino4_equiv, step - wino_equiv * (1.0 - 1.0 / 2.25) - wino4_equiv * 0.75, model_ms),
            "backbone_in_hip_graphs": bool(graph_backbone), "detector_fwd_bwd_ms_all_eager": eager_ms, "peak_hbm_gib": peak, "hip_graph": hip_graph,
            "host_read_backs_per_step": 0 if model._static_ok(x) else "several (compacting forward)",
            "loss_first_iter": first, "loss_last_iter": last, "loss_rose": last > first,
            "note": "NOT the headline metric and NOT Stereo R-CNN's weights: bottleneck stacks [3,4,23,3] with the stride on the first 1x1, "
                    "256-channel FPN P2-P6, stereo RPN 3x3 256->512 on both eyes, RoIAlign 7x7 (both eyes) / 14x14 (left) by libadvengine with "
                    "the deterministic gather backward, ops.nms, six losses; rois_per_image is an assumption (cfg.TRAIN.BATCH_SIZE, upstream)"}


def measure_srcnn(pairs=1, iters=20, reps=2):
    """20-step PGD in the Stereo R-CNN pixel space through surrogates.StereoRcnnShaped (siamese backbone + FPN, stereo RPN, ops.nms,
    pyramid ops.RoIAlign 7x7 / 14x14 with the deterministic gather backward, six uncertainty-weighted losses) at 600x1987"""
    from eval_driving_safety_amd import surrogates
    dev = torch.device("cuda", torch.cuda.current_device())
    model = surrogates.StereoRcnnShaped(seed=0).to(dev).eval()
    net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
    batch = next(iter(data.SyntheticStereo(pairs, "srcnn", batch=pairs, seed=0)))
    batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
    warm = attacks.PgdAttack("srcnn", 1.0, 0.03, 3, save=False, device=dev)
    warm.run_batch(batch, net)
    torch.cuda.synchronize()
    atk = attacks.PgdAttack("srcnn", 1.0, 0.03, iters, save=False, device=dev)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        net.loss_and_grad(x, batch.extra)
    e1.record()
    torch.cuda.synchronize()
    model_ms = e0.elapsed_time(e1) / 3
    t0 = time.perf_counter()
    for _ in range(reps):
        atk.run_batch(batch, net)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    first, last = float(atk.last_losses[0]), float(atk.last_losses[-1])
    return {"metric": "end-to-end stereo-pairs/s, %d-step PGD through a Stereo R-CNN-shaped detector (surrogate, random weights), 600x1987" % iters,
            "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "iters": iters, "s_per_attack": dt, "detector_fwd_bwd_ms": model_ms,
            "loss_first_iter": first, "loss_last_iter": last, "loss_rose": last > first,
            "note": "NOT the headline metric and NOT Stereo R-CNN: plain-convolution pyramid (torch / MIOpen) + stereo RPN + ops.nms + pyramid "
                    "ops.RoIAlign (paired-load forward, atomic-free gather backward) + the six losses of stereo_rcnn.py; PGD step by libadvengine.so"}


def measure_patch(pairs=8, iters=2, reps=2):
    """BASELINE configs[3] end to end: the universal round patch (ratio 0.2605 -> D = 101) trained through the DSGN-shaped graph -
    per pair and inner iteration: paste in both eyes (K3), detector forward + backward towards the fake target, windowed gradient
    sum + clamp + patch update (K4); batch 1, world 1 = the reference's own sequence (attack/DSGN/patch_attack.py:367-430)"""
    import tempfile
    dev = torch.device("cuda", torch.cuda.current_device())
    net = adapters.DsgnShapedAdapter(dev, seed=0)
    import contextlib
    import io
    batches = []                                   # the pairs are generated once, outside the timed region (host-side synthesis)
    for b in data.SyntheticStereo(pairs, "dsgn", batch=1, seed=3):
        b.extra = net.synthetic_extra(b, seed=1)
        batches.append(b)

    def fresh():                                   # the trainer pastes in place: every epoch starts from clean copies
        out = []
        for b in batches:
            c = data.StereoBatch(b.imgL.clone(), b.imgR.clone(), list(b.names), b.sizes)
            c.extra = types.SimpleNamespace(disp_true=b.extra.disp_true, boxes=[list(v) for v in b.extra.boxes])
            out.append(c)
        return out

    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        trainer = attacks.PatchTrainer("dsgn", 0.2605, 8 / 255, iters, 1, out_root=tmp, seed=0, device=dev)
        warm = fresh()
        trainer.train(lambda: warm, net)           # warm: kernels loaded, MIOpen solvers found
        epochs = [fresh() for _ in range(reps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in epochs:
            trainer.train(lambda: e, net)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return {"metric": "end-to-end stereo-pairs/s, universal-patch training (D = %d, %d inner iterations per pair) through the DSGN-shaped graph "
                      "(surrogate, random weights)" % (trainer.patch_dim, iters),
            "value": pairs / dt, "unit": "stereo-pairs/s", "pairs": pairs, "inner_iters": iters, "s_per_epoch_of_%d_pairs" % pairs: dt,
            "ms_per_inner_iteration": 1e3 * dt / (pairs * iters), "patch_abs_max": float(trainer.patch.abs().max()),
            "note": "NOT the headline metric and NOT DSGN; pairs pre-generated on the host, the per-epoch patch.npy write included; paste / update "
                    "kernels and every 3D operator by libadvengine.so"}


def measure_distributed(dist, dev, rank, world, fence, iters=20, patch_pairs_per_rank=2):
    """What the scaling run measures beyond the collective-free perturbation kernel - EVERY rank runs this:
      image_sharded_attack   the DSGN-shaped 20-step PGD on the rank's OWN stereo pair (pairs shard by image, no collective);
                             every rank times its own attack on its own clock (no barrier inside: the leg holds no collective at all);
                             aggregate pairs/s = world pairs / the slowest rank's time, per-rank min / max beside it;
      universal_patch        one universal-patch epoch (D = 101, 2 inner iterations per pair, BASELINE configs[3]) with the patch delta
                             all-reduced INSIDE the loop (attacks.PatchTrainer -> Comm.all_reduce_sum_: RCCL over xGMI, gloo in the
                             1-GPU tests); the all-reduce's share of an inner iteration from device events around the collective.
    Returns the dict on rank 0, None elsewhere."""
    import contextlib
    import io
    import tempfile
    from eval_driving_safety_amd.dist import Comm

    def gather(v):                                  # one float per rank -> list (a SUM all-reduce of a one-hot row: both backends do it)
        t = torch.zeros((world,), dtype=torch.float64, device=dev)
        t[rank] = v
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(q) for q in t.cpu()]

    # Phase A holds NO collective (the attack shards by image and needs none): a rank that fails here fails alone, and the flag exchange
    # right after it is the first thing every rank reaches - so one rank's exception cannot leave the others waiting inside a collective.
    ok, why, mine, atk, net = 1.0, None, 0.0, None, None
    try:
        net = adapters.DsgnShapedAdapter(dev, seed=0)   # every rank holds the same detector (model replicas only)
        batch = next(iter(data.SyntheticStereo(1, "dsgn", batch=1, seed=100 + rank)))
        batch.extra = net.synthetic_extra(batch, seed=1 + rank)
        x = torch.cat([batch.imgL, batch.imgR]).to(dev)
        for _ in range(2):
            net.loss_and_grad(x, batch.extra)
        atk = attacks.PgdAttack("dsgn", 1 / 255, 0.03, iters, save=False, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        atk.run_batch(batch, net)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
    except Exception as e:                              # reported, and the patch leg (which has collectives) is skipped on every rank
        ok, why = 0.0, repr(e)
    oks = gather(ok)
    if min(oks) < 1.0:
        return {"error": "the image-sharded leg failed on rank(s) %s%s" % ([i for i, v in enumerate(oks) if v < 1.0], ": " + why if why else "")} if rank == 0 else None
    times = gather(mine)
    sharded = {"metric": "aggregate stereo-pairs/s, %d-step PGD through the DSGN-shaped graph, one pair per rank, image-sharded, no collective" % iters,
               "value": world / max(times), "unit": "stereo-pairs/s", "per_rank_pairs_per_s_min": 1.0 / max(times), "per_rank_pairs_per_s_max": 1.0 / min(times),
               "per_rank_s_per_attack": times, "loss_rose": bool(float(atk.last_losses[-1]) > float(atk.last_losses[0]))}

    class TimedComm(Comm):
        """the production Comm with device events around the exchange (events on the stream the collective is enqueued on)"""
        spans = []

        def all_reduce_sum_(self, t):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = Comm.all_reduce_sum_(self, t)
            e1.record()
            self.spans.append((e0, e1, t.numel()))
            return out

    comm = TimedComm(rank, world, False)
    n_pairs = world * patch_pairs_per_rank
    batches = []                                    # every rank generates the whole round-robin list and owns i % world == rank
    for b in data.SyntheticStereo(n_pairs, "dsgn", batch=1, seed=3):
        b.extra = net.synthetic_extra(b, seed=1)
        batches.append(b)

    def fresh():
        out = []
        for b in batches:
            c = data.StereoBatch(b.imgL.clone(), b.imgR.clone(), list(b.names), b.sizes)
            c.extra = types.SimpleNamespace(disp_true=b.extra.disp_true, boxes=[list(v) for v in b.extra.boxes])
            out.append(c)
        return out

    inner = 2
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        trainer = attacks.PatchTrainer("dsgn", 0.2605, 8 / 255, inner, 1, out_root=tmp, seed=0, comm=comm, device=dev)
        warm = fresh()
        trainer.train(lambda: warm, net)
        epoch = fresh()
        TimedComm.spans = []
        fence()
        t0 = time.perf_counter()
        trainer.train(lambda: epoch, net)
        torch.cuda.synchronize()
        dt_mine = time.perf_counter() - t0
    fence()
    delta_spans = [(a, b) for a, b, n in TimedComm.spans if n > 2]         # the [3*D*D + 1] patch exchanges (not the 2-element loss statistics)
    ar_ms = sum(a.elapsed_time(b) for a, b in delta_spans)
    dts, ars = gather(dt_mine), gather(ar_ms)
    dt = max(dts)
    rounds = patch_pairs_per_rank * inner
    patch = {"metric": "aggregate stereo-pairs/s, universal-patch training (D = %d, %d inner iterations per pair) through the DSGN-shaped graph, "
                       "pairs dealt round-robin over the ranks, patch delta all-reduced every inner iteration" % (trainer.patch_dim, inner),
             "value": n_pairs / dt, "unit": "stereo-pairs/s", "pairs": n_pairs, "per_rank_s_per_epoch": dts,
             "ms_per_inner_iteration": 1e3 * dt / rounds, "all_reduces_per_rank": len(delta_spans),
             "all_reduce_ms_per_inner_iteration": [v / max(1, len(delta_spans)) for v in ars],
             "all_reduce_share_of_inner_iteration": max(ars) / (1e3 * dt), "message_bytes": 4 * (3 * trainer.patch_dim ** 2 + 1),
             "patch_abs_max": float(trainer.patch.abs().max()),
             "note": "the share includes what a rank WAITS inside the collective for the slowest rank's detector step (the exchange is the "
                     "round's only synchronisation point); bench.py's patch_allreduce object times the bare collective"}
    if rank != 0:
        return None
    return {"image_sharded_attack": sharded, "universal_patch": patch, "world": world, "backend": dist.get_backend() if world > 1 or dist.is_initialized() else "none"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--miopen", action="store_true", help="route the wide convolutions through torch / MIOpen instead")
    ap.add_argument("--hourglass", action="store_true", help="the 3D-hourglass volume network instead of three convolutions")
    ap.add_argument("--dsgn-head", action="store_true", help="fused depth regression + 3D geometric volume + bird's-eye-view detection head")
    ap.add_argument("--patch", action="store_true", help="universal-patch training through the DSGN-shaped graph instead")
    ap.add_argument("--srcnn", action="store_true", help="the Stereo R-CNN-shaped surrogate at 600x1987 instead")
    ap.add_argument("--r101", action="store_true", help="the ResNet-101-FPN Stereo R-CNN-shaped detector (upstream layer list) at 600x1987")
    ap.add_argument("--full", action="store_true", help="the DSGN-shaped graph with SURVEY App. B's layer list (adapters.DsgnShapedAdapter)")
    ap.add_argument("--graph", action="store_true", help="--full: one PGD iteration captured in a hipGraph; --r101: the backbone + FPN forward / backward as hipGraphs")
    ap.add_argument("--r101-graph-leg", action="store_true", help="(child process of --r101) the captured-iteration measurement alone")
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--hip2d", action="store_true", help="--r101: libadvengine's 2D convolution kernels where one exists")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if args.r101_graph_leg:
        print(json.dumps(graph_leg_r101(args.pairs, args.iters, args.rois)))
        return
    if args.full:
        print(json.dumps(measure_dsgn_full(args.pairs, args.iters, args.reps, graph=args.graph, hip2d=(False if args.miopen else (True if args.hip2d else "auto")))))
        return
    if args.r101:
        print(json.dumps(measure_srcnn_r101(args.pairs, args.iters, args.reps, args.rois, "hip" if args.hip2d else ("miopen" if args.miopen else "auto"), graph_backbone=args.graph)))
        return
    if args.srcnn:
        print(json.dumps(measure_srcnn(args.pairs, args.iters, args.reps)))
        return
    if args.patch:
        print(json.dumps(measure_patch(max(args.pairs, 4), 2, args.reps)))
        return
    print(json.dumps(measure(args.pairs, args.iters, args.reps, mfma_conv=not args.miopen, hourglass=args.hourglass, dsgn_head=args.dsgn_head)))


if __name__ == "__main__":
    main()
