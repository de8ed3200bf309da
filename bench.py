#!/usr/bin/env python3
"""bench.py - KITTI stereo-pairs/s of 20-step PGD on DSGN-shaped inputs (BASELINE.json configs[1]),
perturbation path, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one complete 20-iteration L-inf PGD attack (eps 0.03, alpha 1/255, the reference's
script default) of a resident batch of stereo pairs, exactly the per-image work of the reference's
loop body around the detector call (attack/DSGN/pgd_attack.py:279-374):
    clean = denormalize(x0)                         (:297-298)  adv_denormalize_index_f32 (also emits the clean image as a
                                                                device-verified 8-bit index; --no-clean-index: adv_denormalize_f32)
    export iterate 0 as 8-bit HWC                   (:279-294)  adv_export_u8_f32
    20 x { step + project + re-normalise + export } (:339-374)  adv_pgd_step_indexed_f32 / adv_pgd_step_f32 (one launch,
                                                                both eyes of every pair of the batch)
The detector's forward/backward (upstream DSGN, not part of the reference tree) is the caller's: its
gradient is a resident synthetic buffer here, so `value` is the throughput of the perturbation engine
with inputs in HBM, not of an end-to-end attack.  PNG encoding / disk are outside the timed region.

Pairs shard by image over ranks with no collective: every rank attacks its own
`--pairs` pairs, so scaling is weak and value = pairs of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 384, 1248            # DSGN network input (hard-asserted by the reference, patch_attack.py:318-320)
CROP_H, CROP_W = 375, 1242  # KITTI native size the PNGs are cropped back to (pgd_attack.py:192)
N_ITER, EPS, ALPHA = 20, 0.03, 1.0 / 255.0
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=0, help="stereo pairs resident per GPU (per step); 0 = 256 (dsgn) / 96 (srcnn)")
    ap.add_argument("--workload", default="dsgn", choices=["dsgn", "srcnn"],
                    help="dsgn = BASELINE configs[1] (the headline, default); srcnn = configs[2]: 20-step PGD in the Stereo R-CNN "
                         "pixel space on 600x1987 pairs (alpha 1.0, eps 0.03*255) - a parity-test configuration, timed on request")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clean-index", action="store_true",
                    help="read the clean image as float32 in every step instead of as the verified 8-bit index (dsgn workload)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the (separately reported) surrogate-detector attack")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def cpu_baseline(sample_pairs):
    """The oracle (a port: op-for-op restatement of the reference lines, pinned by golden vectors)
    timed on this host: the same step as above for `sample_pairs` pairs."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    try:
        from oracle import oracle_c
        impl, cores = oracle_c, oracle_c.num_threads()
        kind_note = "C oracle (oracle/oracle.c, OpenMP)"
    except Exception:
        from oracle import oracle_np as impl
        cores, kind_note = 1, "numpy oracle (oracle/oracle_np.py)"
    x0 = np.concatenate([synth.dsgn_normalised(i, H, W) for i in range(2)])
    g = synth.gradient(3, x0.shape, 1.0)
    t_budget, done, t0 = 20.0, 0, time.perf_counter()
    target = sample_pairs if sample_pairs > 0 else 10 ** 9
    while done < target:
        clean = impl.denormalize(x0)
        for i in range(2):
            impl.tensor2im_u8(x0[i], CROP_H, CROP_W)
        x = x0
        for _ in range(N_ITER):
            x = impl.pgd_step_norm01(x, g, clean, ALPHA, EPS)
            for i in range(2):
                impl.tensor2im_u8(x[i], CROP_H, CROP_W)
        done += 1
        if sample_pairs <= 0 and time.perf_counter() - t0 > t_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "stereo-pairs/s", "cores": int(cores), "kind": "port",
            "sample": "%d KITTI-shaped pairs x 20-step PGD + 8-bit export, %s, %.1f s" % (done, kind_note, dt)}


def pmc_traffic(pairs, kernel_prefix="pgd_step_vec4<0, 1>"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_pmc_hbm.json, written by tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same bench; gfx950 correction 2*FETCH_SIZE + WRITE_SIZE), scaled to `pairs`.
    Counters cannot be read from inside the timed run, so this is a recorded measurement, not a live one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            k = [v for name, v in d["kernels"].items() if name.startswith(kernel_prefix)][0]
            prof_pairs = d["bench_lines_under_profiler"][-1]["config"]["pairs_per_gpu"]
            return k["hbm_bytes_per_launch"] * pairs / prof_pairs, os.path.basename(path)
        except Exception:
            continue
    return None, None


def main():
    args = parse()
    global H, W, CROP_H, CROP_W, ALPHA, EPS
    srcnn = args.workload == "srcnn"
    if srcnn:   # attack/Stereo-RCNN/pgd_attack.py: network scale 600x1987, no crop (quirk Q14), alpha 1.0, eps = 255*0.03 (:57)
        H, W, CROP_H, CROP_W, ALPHA, EPS = 600, 1987, 600, 1987, 1.0, 255 * 0.03
    if args.pairs <= 0:
        args.pairs = 96 if srcnn else 256
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # test hooks (a 1-GPU box cannot host two RCCL ranks): ADV_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
        # ADV_BENCH_BACKEND=gloo swaps the backend, so the world > 1 code path can be smoke-tested anywhere
        share = os.environ.get("ADV_BENCH_SHARE_GPU") == "1"
        backend = os.environ.get("ADV_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(0 if share else local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())

    from eval_driving_safety_amd import ops   # raises if libadvengine.so is not built
    sp = ops.Space.srcnn() if srcnn else ops.Space.dsgn()

    n_img = 2 * args.pairs                    # both eyes of every pair in one launch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x0 = torch.randint(0, 256, (n_img, 3, H, W), device=dev, generator=gen, dtype=torch.int32).float()
    if srcnn:
        x0 -= torch.tensor([102.9801, 115.9465, 122.7717], device=dev).view(1, 3, 1, 1)   # BGR minus PIXEL_MEANS
    else:
        # ToTensor's v/255 as a TRUE division (what the CPU loader computes); torch-on-GPU would turn a division by a
        # Python scalar into a multiplication by the reciprocal, which is not the same float32 function
        x0.div_(torch.full((), 255.0, device=dev))
        ops.normalize(x0, sp, out=x0)         # what the DSGN loader hands over: normalised float32
    grad = torch.randn((n_img, 3, H, W), device=dev, generator=gen)
    clean = torch.empty_like(x0)
    x = torch.empty_like(x0)
    spare = torch.empty_like(x0) if srcnn else None   # planes are not whole cache lines: alternate buffers
    u8 = ops.alloc_u8(n_img, CROP_H, W, dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    use_index = (not srcnn) and (not args.no_clean_index)
    cidx = [None]

    def step(k=None):
        if srcnn:
            clean.copy_(x0)                   # explicit clone of the clean pair (pgd_attack.py:122-123, quirk Q6)
        elif use_index:                       # denormalize + verified 8-bit index of the clean image (read as bytes by the 20 steps)
            _, cidx[0] = ops.denormalize_indexed(x0, sp, out=clean, reuse=cidx[0])
        else:
            ops.denormalize(x0, sp, out=clean)
        ops.export_u8(x0, sp, (CROP_H, CROP_W), out=u8)
        if k is not None:
            ev0[k].record()
        kw = {"clean_index": cidx[0]} if use_index else {}
        ops.pgd_step(x0, grad, clean, sp, ALPHA, EPS, out=x, u8_out=u8, crop=(CROP_H, CROP_W), **kw)
        cur, nxt = x, (spare if srcnn else x)
        for _ in range(N_ITER - 1):
            ops.pgd_step(cur, grad, clean, sp, ALPHA, EPS, out=nxt, u8_out=u8, crop=(CROP_H, CROP_W), **kw)
            cur, nxt = nxt, cur
        if k is not None:
            ev1[k].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: pgd_step_vec4<AFFINE, rows-dword u8>; HIP events on the launch stream
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / (args.steps * N_ITER)
    elems = 3 * H * W
    alg_bytes = n_img * (16 * elems + 3 * CROP_H * CROP_W)     # 16 B/elt + the 8-bit export
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    index_verified = bool(int(cidx[0].ok.item())) if use_index and cidx[0] is not None else None
    if rank == 0:
        traffic, traffic_src = (None, None) if srcnn else pmc_traffic(
            args.pairs, "pgd_step_vec4_idx<1>" if index_verified else "pgd_step_vec4<0, 1>")
        out = {
            "metric": "KITTI stereo-pairs/sec for 20-step PGD on %s (perturbation path; detector fwd+bwd is the caller's)"
                      % ("Stereo R-CNN" if srcnn else "DSGN"),
            "value": world * args.pairs * args.steps / elapsed,
            "unit": "stereo-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: 20-step PGD Linf eps=0.03*255 alpha=1.0, Stereo R-CNN pixel space (BGR minus "
                                    "PIXEL_MEANS), 600x1987 network-scale pairs, %d stereo pairs resident per GPU, gradient = resident "
                                    "synthetic buffer, 8-bit HWC export of all 21 iterates" % args.pairs) if srcnn else
                                   ("BASELINE configs[1]: 20-step PGD Linf eps=0.03 alpha=1/255, DSGN pixel space, "
                                    "KITTI 1242x375 padded to 1248x384, %d stereo pairs resident per GPU, gradient = "
                                    "resident synthetic buffer, 8-bit HWC export of all 21 iterates" % args.pairs),
                       "pairs_per_gpu": args.pairs, "pgd_iters": N_ITER, "eps": EPS, "alpha": ALPHA,
                       "parallelism": "image-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "pgd_step_shifted<IDENTITY,U8_BYTES>" if srcnn else
                         ("pgd_step_vec4_idx<U8_ROWS_DWORD>" if use_index else "pgd_step_vec4<AFFINE,U8_ROWS_DWORD>"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "clean_image_read_as": ("uint8 index (verified on the device)" if index_verified else "float32"),
                         # `achieved` follows the contract (ALGORITHMIC bytes: 16 B/elt + export); the kernel really moves
                         # `traffic` bytes - less when the clean image is read as bytes - so the HBM itself is this busy:
                         "hbm_utilisation": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None},
        }
        if world == 1 and not args.no_cpu_baseline and not srcnn:
            out["cpu_baseline"] = cpu_baseline(args.cpu_pairs)
        if world == 1 and not args.no_end_to_end and not srcnn:
            # the end-to-end number is reported BESIDE the kernel-path one, never folded into `value`
            del x0, grad, clean, x, u8
            torch.cuda.empty_cache()
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_end_to_end
                out["end_to_end"] = bench_end_to_end.measure(pairs=1, iters=N_ITER, reps=3)
            except Exception as e:
                out["end_to_end"] = {"error": repr(e)}
    else:
        out = None

    # Outside the timed region, N > 1 only: latency of the one collective the attacks have - the all-reduce(SUM)
    # of the universal-patch delta [3,D,D] (D = 101: BASELINE configs[3], 122 KB) over RCCL / xGMI - checked
    # against the closed form, so the multi-GPU patch path runs on real links whenever the scaling bench does.
    # Never part of `value`; a watchdog prints the throughput line without it if the collective stalls.
    if world > 1:
        import threading

        def bail():
            if rank == 0:
                out["patch_allreduce"] = {"error": "timed out after 120 s"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(120.0, bail)
        dog.daemon = True
        dog.start()
        try:
            d = 101
            delta = torch.full((3, d, d), float(rank + 1), device=dev)
            for _ in range(5):
                dist.all_reduce(delta.clone())
            torch.cuda.synchronize()
            reps = 50
            bufs = [delta.clone() for _ in range(reps)]
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for b in bufs:
                dist.all_reduce(b)
            c1.record()
            torch.cuda.synchronize()
            ok = bool((bufs[-1] == world * (world + 1) / 2).all().item())
            patch_comm = {"collective": "all_reduce(SUM) of the patch delta [3,101,101] f32 (122412 B) over RCCL",
                          "avg_us": 1e3 * c0.elapsed_time(c1) / reps, "correct": ok}
        except Exception as e:                       # report, never fail the throughput line
            patch_comm = {"error": repr(e)}
        dog.cancel()
        if rank == 0:
            out["patch_allreduce"] = patch_comm
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
