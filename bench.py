#!/usr/bin/env python3
"""bench.py - KITTI stereo-pairs/s of 20-step PGD on DSGN-shaped inputs (BASELINE.json configs[1]),
perturbation path, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one complete 20-iteration L-inf PGD attack (eps 0.03, alpha 1/255, the reference's
script default) of a resident batch of stereo pairs, exactly the per-image work of the reference's
loop body around the detector call (attack/DSGN/pgd_attack.py:279-374):
    clean = denormalize(x0), export iterate 0       (:279-298)  adv_clean_index_build_f32 (one pass: clean image, its
                                                                per-image device-verified 8-bit index, the 8-bit export)
    20 x { step + project + re-normalise + export } (:339-374)  adv_pgd_step_indexed_f32 (one launch, both eyes of every
                                                                pair of the batch, the iterate updated in place)
The input is what the reference's loader hands over: 8-bit 375x1242 images (low-pass
noise; the right eye is the left one shifted by a ground-plane disparity) -> /255 -> ImageNet normalisation ->
ZERO-PADDED in normalised space to 384x1248.  `float_path` in the JSON line is the same attack with every stream read
as float32 (adv_denormalize_f32 + adv_export_u8_f32 + adv_pgd_step_f32), timed in the same run.
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
SR_H, SR_W = 600, 1987      # Stereo R-CNN network scale (attack/Stereo-RCNN/patch_attack.py:170-172)


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
                    help="headline = the all-float32 path (the clean image read as float32 in every step)")
    ap.add_argument("--alternate", action="store_true", help="alternate two iterate buffers instead of updating in place (same speed within 1 %%)")
    ap.add_argument("--unpadded", action="store_true", help="fill the whole 384x1248 frame with 8-bit pixels (round-1 input; no padding)")
    ap.add_argument("--no-float-path", action="store_true", help="skip the second (all-float32) measurement")
    ap.add_argument("--no-srcnn", action="store_true", help="skip the configs[2] (Stereo R-CNN shape) object")
    ap.add_argument("--no-delivered", action="store_true", help="skip the leg that delivers every iterate to pinned host memory (PCIe-inclusive figure)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the (separately reported) surrogate-detector attack")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--torch-cpu-baseline", action="store_true", help="(internal) print the torch-CPU baseline object and exit; touches no GPU")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(sample_pairs):
    """The oracle (a port: op-for-op restatement of the reference lines, pinned by golden vectors) timed on this host:
    the same step as above for a bounded sample; beside it the reference's own torch-CPU formulation (the ~12
    elementwise torch ops per eye of attack/DSGN/pgd_attack.py:339-354 plus tensor2im) on all host threads."""
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
    x0 = np.concatenate([synth.dsgn_padded(i, CROP_H, CROP_W, H, W) for i in range(2)])
    g = synth.gradient(3, x0.shape, 1.0)
    t_budget, done, t0 = 12.0, 0, time.perf_counter()
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
    out = {"value": done / dt, "unit": "stereo-pairs/s", "cores": int(cores), "kind": "port",
           "sample": "%d KITTI-shaped pairs x 20-step PGD + 8-bit export, %s, %.1f s" % (done, kind_note, dt)}
    out["torch_cpu"] = torch_cpu_subprocess()
    return out


def torch_cpu_subprocess(timeout_s=90):
    """run `bench.py --torch-cpu-baseline` in a fresh process: its intra-op thread pool must not fight the C oracle's
    OpenMP team (measured: 125 s per pair when both live in one process, 256 + 128 spinning threads)"""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:                                   # never lose the bench line over the second baseline
        return {"error": repr(e)}


def torch_cpu_baseline(budget_s=10.0):
    """attack/DSGN/pgd_attack.py:196-207,339-354 + tensor2im (:157-179) as torch-CPU eager ops, one pair per pass as the
    reference runs it (batch 1), torch's default intra-op thread count (= what the reference script would get)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    x0_np = np.concatenate([synth.dsgn_padded(i, CROP_H, CROP_W, H, W) for i in range(2)])
    g_np = synth.gradient(3, x0_np.shape, 1.0)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    threads = torch.get_num_threads()

    def denormalize(im):                                     # :196-200
        for i in range(3):
            im.data[0][i] = im.data[0][i] * std[i] + mean[i]
        return im

    def normalize(im):                                       # :203-207
        for i in range(3):
            im.data[0][i] = (im.data[0][i] - mean[i]) / std[i]
        return im

    def tensor2im(t):                                        # :157-179 (+ the crop of save_img, :192)
        a = t.cpu().float().numpy()
        for i in range(3):
            a[i] = a[i] * std[i] + mean[i]
        a = a * 255
        return np.transpose(a, (1, 2, 0)).astype(np.uint8)[:CROP_H, :CROP_W]

    xs = [torch.from_numpy(x0_np[i:i + 1].copy()) for i in range(2)]
    gs = [torch.from_numpy(g_np[i:i + 1].copy()) for i in range(2)]
    done, t0 = 0, time.perf_counter()
    while True:
        imgs = [x.clone() for x in xs]
        cleans = [denormalize(x.clone()) for x in xs]
        for im in imgs:
            tensor2im(im[0])
        for _ in range(N_ITER):
            for e in range(2):
                im = denormalize(imgs[e])
                adv = im + ALPHA * gs[e].sign()
                eta = torch.clamp(adv - cleans[e], min=-EPS, max=EPS)
                im = torch.clamp(cleans[e] + eta, min=0, max=1)
                imgs[e] = normalize(im).detach()
                tensor2im(imgs[e][0])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": done / dt, "unit": "stereo-pairs/s", "threads": threads, "cpu": model, "kind": "reference formulation (torch-CPU eager)",
            "sample": "%d KITTI-shaped pairs x 20-step PGD + tensor2im, batch 1 as the reference runs it, %.1f s" % (done, dt)}


def pmc_traffic(pairs, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_pmc_hbm*.json, written by tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE /
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


# ---------------------------------------------------------------------------------------------- synthetic input
def kitti_like_input(torch, ops, sp, pairs, dev, gen, padded=True):
    """[2*pairs,3,384,1248] float32 as the DSGN loader hands it over: 8-bit low-pass noise images of 375x1242, the right
    eye = the left one shifted row by row by a ground-plane disparity fu*b/z (z from 40 m at the top to 5 m at the
    bottom), ToTensor (v/255 as a TRUE division - torch on a GPU would turn a division by a Python scalar into a
    multiplication by the reciprocal, which is not the float32 function a CPU loader computes), ImageNet
    normalisation, zero padding in normalised space (data.dsgn_transform does the same on the host)."""
    F = torch.nn.functional
    vh, vw = (CROP_H, CROP_W) if padded else (H, W)
    out = torch.zeros((2 * pairs, 3, H, W), device=dev)
    z = torch.linspace(40.0, 5.0, vh, device=dev)
    disp = (721.5377 * 0.54 / z).round().long()                              # 10 .. 78 px
    cols = (torch.arange(vw, device=dev)[None, :] + disp[:, None]).clamp_(max=vw - 1)   # right(x) = left(x + d)
    two55 = torch.full((), 255.0, device=dev)
    chunk = 32
    for p0 in range(0, pairs, chunk):
        b = min(chunk, pairs - p0)
        low = torch.randint(0, 256, (b, 3, vh // 8 + 2, vw // 8 + 2), device=dev, generator=gen, dtype=torch.int32).float()
        left = F.interpolate(low, size=(vh, vw), mode="bilinear", align_corners=False)
        left += torch.randint(-8, 9, (b, 3, vh, vw), device=dev, generator=gen, dtype=torch.int32).float()
        left.clamp_(0, 255).round_()
        right = torch.gather(left, 3, cols[None, None].expand(b, 3, vh, vw))
        for eye, img in ((0, left), (1, right)):
            t = img.div(two55).contiguous()
            ops.normalize(t, sp, out=t)                                      # (t - mean) / std, true division
            out[eye * pairs + p0:eye * pairs + p0 + b, :, :vh, :vw] = t
    return out, (vh, vw)


class PgdBench:
    """one resident batch and its buffers; `attack()` enqueues one complete 20-step attack"""

    def __init__(self, torch, ops, sp, x0, grad, valid, crop, use_index, in_place, affine=True):
        self.torch, self.ops, self.sp = torch, ops, sp
        self.x0, self.grad, self.valid, self.crop = x0, grad, valid, crop
        self.use_index, self.in_place, self.affine = use_index, in_place, affine
        n = x0.shape[0]
        self.clean = torch.empty_like(x0)
        self.a = torch.empty_like(x0)
        self.b = None if in_place else torch.empty_like(x0)
        self.u8 = ops.alloc_u8(n, crop[0], x0.shape[3], x0.device)
        self.cidx = None

    def attack(self, ev0=None, ev1=None):
        ops, sp = self.ops, self.sp
        if not self.affine and not self.use_index:
            self.clean.copy_(self.x0)        # explicit clone of the clean pair (Stereo R-CNN pgd_attack.py:122-123, quirk Q6)
            ops.export_u8(self.x0, sp, self.crop, out=self.u8)
        elif self.use_index:                 # denormalize + per-image verified 8-bit index + iterate-0 export, one pass
            _, self.cidx = ops.denormalize_indexed(self.x0, sp, out=self.clean, reuse=self.cidx, valid=self.valid,
                                                   u8_out=self.u8, crop=self.crop)
        else:
            ops.denormalize(self.x0, sp, out=self.clean)
            ops.export_u8(self.x0, sp, self.crop, out=self.u8)
        if ev0 is not None:
            ev0.record()
        kw = {"clean_index": self.cidx} if self.use_index else {}
        cur, nxt = self.x0, self.a
        for _ in range(N_ITER):
            ops.pgd_step(cur, self.grad, self.clean, sp, ALPHA, EPS, out=nxt, u8_out=self.u8, crop=self.crop, **kw)
            cur, nxt = nxt, (nxt if self.in_place else (self.b if nxt is self.a else self.a))
        if ev1 is not None:
            ev1.record()

    def timed(self, steps, warmup, fence):
        torch = self.torch
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        for _ in range(warmup):
            self.attack()
        fence()
        t0 = time.perf_counter()
        for k in range(steps):
            self.attack(ev0[k], ev1[k])
        fence()
        elapsed = time.perf_counter() - t0
        kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / (steps * N_ITER)   # HIP events on the launch stream
        return elapsed, kern_ms


def export_delivered(torch, ops, sp, x0, grad, valid, crop, fence, pairs, steps=2):
    """The same resident batch and the same 20-step attack, with every iterate DELIVERED: the reference hands each step's 8-bit images to
    the host (attack/DSGN/pgd_attack.py:357-374: tensor2im + save_img per step; iterate 0 at :279-294).  Two device export buffers alternate;
    a side stream copies each one into a ring of two PINNED host buffers as soon as its step kernel has written it (event-fenced both ways:
    the copy waits for the kernel, the kernel that reuses a buffer waits for its copy), so the D2H of iterate k runs under the kernels of
    iterates k+1 ....  Reported BESIDE `value`, never in it: with delivery the path is bound by PCIe, not by HBM.  Also measured: the 21
    copies alone and the kernels alone, so that the share of the kernel time hidden behind the copies can be read off."""
    dev = x0.device
    n = x0.shape[0]
    u8 = [ops.alloc_u8(n, crop[0], x0.shape[3], dev) for _ in range(2)]
    host = [torch.empty(tuple(u8[0].shape), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    clean, cur = torch.empty_like(x0), torch.empty_like(x0)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    state = {"cidx": None}

    def deliver(slot):
        ready[slot].record(main)
        side.wait_event(ready[slot])
        with torch.cuda.stream(side):
            host[slot].copy_(u8[slot], non_blocking=True)
            freed[slot].record(side)

    def attack(copies=True, kernels=True):
        if kernels:
            _, state["cidx"] = ops.denormalize_indexed(x0, sp, out=clean, reuse=state["cidx"], valid=valid, u8_out=u8[0], crop=crop)
        if copies:
            deliver(0)
        src = x0
        for it in range(N_ITER):
            slot = (it + 1) & 1
            if copies:
                main.wait_event(freed[slot])           # the copy that last read this export buffer (recorded before its first use too)
            if kernels:
                ops.pgd_step(src, grad, clean, sp, ALPHA, EPS, out=cur, u8_out=u8[slot], crop=crop, clean_index=state["cidx"])
                src = cur
            if copies:
                deliver(slot)

    for e in freed:
        e.record(side)

    def timed(**kw):
        attack(**kw)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            attack(**kw)
        fence()
        return (time.perf_counter() - t0) / steps

    t_kernels = timed(copies=False)
    t_copies = timed(kernels=False)
    t_both = timed()
    # the delivered bytes are the attack's own: iterate 20 on the host equals the device buffer the last step wrote
    ok = bool(torch.equal(host[N_ITER & 1], u8[N_ITER & 1].cpu()))
    nbytes = (N_ITER + 1) * u8[0].numel()
    hidden = max(0.0, min(1.0, (t_kernels + t_copies - t_both) / t_kernels))
    return {"metric": "stereo-pairs/s with all 21 8-bit iterates of every pair delivered to pinned host memory (PCIe-inclusive; NOT `value`)",
            "value": pairs / t_both, "unit": "stereo-pairs/s", "pairs": pairs, "steps": steps,
            "ms_per_attack": {"kernels_and_copies": 1e3 * t_both, "kernels_alone": 1e3 * t_kernels, "copies_alone": 1e3 * t_copies},
            "d2h_bytes_per_attack": nbytes, "d2h_GBps": nbytes / t_both / 1e9, "d2h_GBps_copies_alone": nbytes / t_copies / 1e9,
            "kernel_time_hidden_behind_copies": hidden, "bound": "pcie",
            "delivered_equals_device": ok,
            "how": "two alternating device export buffers [2B,375,1248,3] u8, side-stream D2H into a ring of two pinned host buffers, "
                   "event-fenced both ways; reference: attack/DSGN/pgd_attack.py:357-374 writes every iterate of every pair"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` (this parent has not imported torch.cuda or touched a
    GPU, and it never replaces itself: it waits for the child), relay rank 0's single JSON line and exit with the child's
    status (the ranks' watchdog exit code 3 included)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ADV_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:
        if line.startswith("{"):
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        print("bench.py: the %d-rank child exited 0 without a JSON line" % args.gpus, file=sys.stderr)
        rc = 4
    sys.exit(rc)


def main():
    args = parse()
    global H, W, CROP_H, CROP_W, ALPHA, EPS
    if args.torch_cpu_baseline:
        print(json.dumps(torch_cpu_baseline()), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # launched plainly: become the launcher (before anything touches a GPU)
        self_launch(args)
    srcnn = args.workload == "srcnn"
    if srcnn:   # attack/Stereo-RCNN/pgd_attack.py: network scale 600x1987, no crop (quirk Q14), alpha 1.0, eps = 255*0.03 (:57)
        H, W, CROP_H, CROP_W, ALPHA, EPS = SR_H, SR_W, SR_H, SR_W, 1.0, 255 * 0.03
    if args.pairs <= 0:
        args.pairs = 96 if srcnn else 256
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:         # a launcher that started a different number of ranks: refuse rather than mislabel the line
        if rank == 0:
            print("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d; start it plainly (`python bench.py --gpus %d` launches its own "
                  "ranks) or with --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    # ADV_BENCH_FORCE_DIST=1: take the distributed code path (process group, barriers, MAX-reduce, patch all-reduce probe) even
    # with WORLD_SIZE=1, so that the RCCL branch can be exercised on a single-GPU box
    use_dist = world > 1 or os.environ.get("ADV_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    dev = torch.device("cuda", torch.cuda.current_device())

    from eval_driving_safety_amd import ops   # raises if libadvengine.so is not built
    sp = ops.Space.srcnn() if srcnn else ops.Space.dsgn()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            if dist.get_backend() == "gloo":
                dist.barrier()
            else:
                dist.barrier(device_ids=[torch.cuda.current_device()])
        torch.cuda.synchronize()

    n_img = 2 * args.pairs                    # both eyes of every pair in one launch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    if srcnn:
        x0 = torch.randint(0, 256, (n_img, 3, H, W), device=dev, generator=gen, dtype=torch.int32).float()
        x0 -= torch.tensor([102.9801, 115.9465, 122.7717], device=dev).view(1, 3, 1, 1)   # BGR minus PIXEL_MEANS
        valid = (H, W)
    else:
        x0, valid = kitti_like_input(torch, ops, sp, args.pairs, dev, gen, padded=not args.unpadded)
    grad = torch.randn((n_img, 3, H, W), device=dev, generator=gen)
    crop = (CROP_H, CROP_W)
    use_index = not args.no_clean_index
    main_b = PgdBench(torch, ops, sp, x0, grad, valid, crop, use_index, (not args.alternate), affine=not srcnn)
    elapsed, kern_ms = main_b.timed(args.steps, args.warmup, fence)
    per_rank = None
    if use_dist:                   # MAX over ranks is the job's time; every rank's own clock is kept for the line (min/max pairs/s)
        t = torch.zeros((dist.get_world_size(), 2), device=dev, dtype=torch.float64)     # own row filled, SUM = a gather (one
        t[dist.get_rank(), 0], t[dist.get_rank(), 1] = elapsed, kern_ms                      # collective both backends support)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [(float(v[0]), float(v[1])) for v in t.cpu()]
        elapsed = max(v[0] for v in per_rank)

    elems = 3 * H * W
    alg_bytes = n_img * (16 * elems + 3 * CROP_H * CROP_W)     # 16 B/elt + the 8-bit export
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    flags = main_b.cidx.verified() if use_index and main_b.cidx is not None else None
    n_indexed = sum(flags) if flags is not None else 0

    # the same attack with every stream read as float32, timed in the same run (dsgn workload, rank 0's own clock)
    float_path = None
    if not srcnn and use_index and not args.no_float_path:
        del main_b.a, main_b.b
        fb = PgdBench(torch, ops, sp, x0, grad, valid, crop, False, (not args.alternate))
        fe, fk = fb.timed(max(2, args.steps // 3), 1, fence)
        fsteps = max(2, args.steps // 3)
        float_path = {"value": world * args.pairs * fsteps / fe, "unit": "stereo-pairs/s", "ms_per_step": 1e3 * fe / fsteps,
                      "kernel": "pgd_step_vec4<AFFINE,U8_ROWS_DWORD>", "avg_launch_ms": fk,
                      "achieved": alg_bytes / (fk * 1e-3) / 1e9, "frac": alg_bytes / (fk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "adv_denormalize_f32 + adv_export_u8_f32 + 20 x adv_pgd_step_f32: what a batch of images WITHOUT "
                              "an 8-bit origin takes (it reads exactly the algorithmic bytes)"}
        del fb

    if rank == 0:
        kern_name = (("pgd_step_shifted<IDENTITY,IDX>" if n_indexed else "pgd_step_shifted<IDENTITY>") if srcnn else ("pgd_step_vec4_idx<U8_ROWS_DWORD>" if n_indexed else "pgd_step_vec4<AFFINE,U8_ROWS_DWORD>"))
        traffic, traffic_src = (None, None) if srcnn else pmc_traffic(args.pairs, "pgd_step_vec4_idx<1>" if n_indexed else "pgd_step_vec4<0, 1>")
        out = {
            "metric": "KITTI stereo-pairs/sec for 20-step PGD on %s (perturbation path; detector fwd+bwd is the caller's)"
                      % ("Stereo R-CNN" if srcnn else "DSGN"),
            "value": world * args.pairs * args.steps / elapsed,
            "unit": "stereo-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "rccl_world": dist.get_world_size() if use_dist else 1,       # what the process group itself says, not what --gpus asked for
            "backend": (dist.get_backend() if use_dist else "none (single process)"),
            "launched_by": ("bench.py itself (child torch.distributed.run)" if os.environ.get("ADV_BENCH_SELF_LAUNCHED") == "1"
                            else ("an outer launcher" if "TORCHELASTIC_RUN_ID" in os.environ or world > 1 else "plain python")),
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: 20-step PGD Linf eps=0.03*255 alpha=1.0, Stereo R-CNN pixel space (BGR minus "
                                    "PIXEL_MEANS), 600x1987 network-scale pairs, %d stereo pairs resident per GPU, gradient = resident "
                                    "synthetic buffer, 8-bit HWC export of all 21 iterates" % args.pairs) if srcnn else
                                   ("BASELINE configs[1]: 20-step PGD Linf eps=0.03 alpha=1/255, DSGN pixel space, 8-bit KITTI-shaped "
                                    "%s, %d stereo pairs resident per GPU, gradient = resident synthetic buffer, 8-bit HWC export of "
                                    "all 21 iterates" % ("1242x375 images normalised and zero-padded to 1248x384 as the loader does"
                                                         if not args.unpadded else "images filling the whole 1248x384 frame (no padding)",
                                                         args.pairs)),
                       "pairs_per_gpu": args.pairs, "pgd_iters": N_ITER, "eps": EPS, "alpha": ALPHA,
                       "iterate_buffers": "in place" if (not args.alternate) else "two alternating",
                       "parallelism": "image-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": kern_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "clean_image_read_as": ("uint8 index for %d of %d images (verified per image on the device)" % (n_indexed, n_img)
                                                 if use_index else "float32"),
                         # `achieved` follows the contract (ALGORITHMIC bytes: 16 B/elt + export); the kernel really moves
                         # `traffic` bytes - less when the clean image is read as bytes - so the HBM itself is this busy:
                         "hbm_utilisation": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None},
        }
        if per_rank is not None:
            rates = [args.pairs * args.steps / e for e, _ in per_rank]
            out["per_rank"] = {"pairs_per_s_min": min(rates), "pairs_per_s_max": max(rates), "pairs_per_s": rates,
                               "kernel_avg_launch_ms": [k for _, k in per_rank]}
        if float_path is not None:
            out["float_path"] = float_path
    else:
        out = None

    if rank == 0 and world == 1 and not srcnn and use_index and not args.no_delivered:
        try:        # the attack with every iterate handed to the host, as the reference's loop does (beside `value`, never in it)
            out["export_delivered"] = export_delivered(torch, ops, sp, x0, grad, valid, crop, fence, args.pairs)
        except Exception as e:
            out["export_delivered"] = {"error": repr(e)}
    del main_b, x0, grad
    torch.cuda.empty_cache()

    if rank == 0 and world == 1:
        if not srcnn and not args.no_srcnn:
            try:
                out["configs2_srcnn"] = srcnn_object(torch, ops, dev, fence)
            except Exception as e:
                out["configs2_srcnn"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if not args.no_end_to_end and not srcnn:
            # the end-to-end number is reported BESIDE the kernel-path one, never folded into `value`.
            # (Measured BEFORE the CPU baselines: their OpenMP teams keep spinning on the host cores afterwards.)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_end_to_end
                # BASELINE configs[1] end to end: the DSGN-shaped graph with SURVEY App. B's layer list, exact FLOPs per step, the WHOLE
                # step against the float32 matrix peak - at the reference's 1 pair per step and at 4 (288 GB hold far more)
                out["end_to_end_dsgn_shaped"] = bench_end_to_end.measure_dsgn_full(pairs=1, iters=N_ITER, reps=1)
                out["end_to_end_dsgn_shaped"]["batch_of_4_pairs"] = bench_end_to_end.measure_dsgn_full(pairs=4, iters=N_ITER, reps=1)
                try:    # the same loop with one iteration captured in a hipGraph (launch gaps of ~10^3 kernels per step)
                    out["end_to_end_dsgn_shaped"]["hip_graph"] = bench_end_to_end.measure_dsgn_full(pairs=1, iters=N_ITER, reps=1, graph=True)
                except Exception as e:
                    out["end_to_end_dsgn_shaped"]["hip_graph"] = {"error": repr(e)}
                # BASELINE configs[2] end to end: ResNet-101-FPN Stereo R-CNN-shaped detector at 600x1987
                out["end_to_end_srcnn_shaped"] = bench_end_to_end.measure_srcnn_r101(pairs=1, iters=N_ITER, reps=1)
                # BASELINE configs[3]: universal-patch training through the DSGN-shaped graph
                out["end_to_end_patch"] = bench_end_to_end.measure_patch(pairs=4, iters=2, reps=1)
            except Exception as e:
                out["end_to_end_error"] = repr(e)
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and not srcnn:
            out["cpu_baseline"] = cpu_baseline(args.cpu_pairs)

    # Outside the timed region, N > 1 only: latency of the one collective the attacks have - the all-reduce(SUM)
    # of the universal-patch delta [3,D,D] (D = 101: BASELINE configs[3], 122 KB) over RCCL / xGMI - checked
    # against the closed form, so the multi-GPU patch path runs on real links whenever the scaling bench does.
    # Never part of `value`.  If the collective stalls, the watchdog prints the throughput line with the error and the
    # process exits NON-ZERO, so that a launcher can tell a stall from success.
    if use_dist:
        import threading

        def bail():
            if rank == 0:
                out["patch_allreduce"] = {"error": "a collective of the N > 1 legs timed out"}
                print(json.dumps(out), flush=True)
            os._exit(3)

        dog = threading.Timer(600.0 if not (args.no_end_to_end or srcnn) else 120.0, bail)
        dog.daemon = True
        dog.start()
        # N > 1: what the ranks do END TO END - each its own 20-step DSGN-shaped attack (image-sharded, no collective) and a universal-patch
        # epoch with the delta all-reduced inside the loop (tools/bench_end_to_end.measure_distributed).  Beside `value`, never in it.
        if not args.no_end_to_end and not srcnn:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_end_to_end
                legs = bench_end_to_end.measure_distributed(dist, dev, rank, world, fence, iters=N_ITER)
            except Exception as e:
                legs = {"error": repr(e)}
            if rank == 0:
                out["distributed_end_to_end"] = legs
            torch.cuda.empty_cache()
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
            patch_comm = {"collective": "all_reduce(SUM) of the patch delta [3,101,101] f32 (122412 B) over %s" % dist.get_backend(),
                          "avg_us": 1e3 * c0.elapsed_time(c1) / reps, "correct": ok}
        except Exception as e:                       # report, never fail the throughput line
            patch_comm = {"error": repr(e)}
        dog.cancel()
        if rank == 0:
            out["patch_allreduce"] = patch_comm
    if rank == 0:
        from eval_driving_safety_amd import routes
        # which kernel computes each convolution of the end-to-end legs: the committed table (same in every process and rank)
        out["routes"] = dict(routes.summary(), unknown_shapes=routes.misses()[:8])
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def srcnn_object(torch, ops, dev, fence, pairs=64, steps=3, use_index=True):
    """BASELINE configs[2] beside the headline: 20-step PGD in the Stereo R-CNN pixel space (attack/Stereo-RCNN/
    pgd_attack.py:177-243) on 600x1987 pairs, alpha 1.0, eps 0.03*255, 8-bit export of every iterate."""
    global ALPHA, EPS
    sp = ops.Space.srcnn()
    n_img = 2 * pairs
    gen = torch.Generator(device=dev).manual_seed(99)
    x0 = torch.randint(0, 256, (n_img, 3, SR_H, SR_W), device=dev, generator=gen, dtype=torch.int32).float()
    x0 -= torch.tensor([102.9801, 115.9465, 122.7717], device=dev).view(1, 3, 1, 1)
    grad = torch.randn((n_img, 3, SR_H, SR_W), device=dev, generator=gen)
    keep = (ALPHA, EPS)
    ALPHA, EPS = 1.0, 255 * 0.03
    try:
        b = PgdBench(torch, ops, sp, x0, grad, None, (SR_H, SR_W), use_index, True, affine=False)
        elapsed, kern_ms = b.timed(steps, 1, fence)
        verified = sum(b.cidx.verified()) if b.cidx is not None else 0
        fl = None
        if use_index:                        # the all-float32 kernel beside it, as the headline does
            bf = PgdBench(torch, ops, sp, x0, grad, None, (SR_H, SR_W), False, True, affine=False)
            el_f, k_f = bf.timed(steps, 1, fence)
            fl = {"value": pairs * steps / el_f, "unit": "stereo-pairs/s", "kernel": "pgd_step_shifted<IDENTITY>", "avg_launch_ms": k_f}
    finally:
        ALPHA, EPS = keep
    alg = n_img * (16 * 3 * SR_H * SR_W + 3 * SR_H * SR_W)
    moved = alg - (3 * verified * 3 * SR_H * SR_W if use_index else 0)
    out = {"metric": "stereo-pairs/s, 20-step PGD in the Stereo R-CNN pixel space (perturbation path)", "value": pairs * steps / elapsed,
           "unit": "stereo-pairs/s", "pairs": pairs, "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
           "roofline": {"bound": "hbm", "kernel": "pgd_step_shifted<IDENTITY%s>" % (",IDX" if use_index else ""),
                        "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": kern_ms,
                        "algorithmic_bytes_per_launch": alg, "bytes_moved_per_launch_by_design": moved,
                        "hbm_utilisation_by_design_bytes": moved / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "clean_image_read_as": ("uint8 index for %d of %d images (verified per image on the device)" % (verified, n_img)
                                                if use_index else "float32")}}
    # recorded PMC bytes of this object's kernel (the profile runs the default bench, i.e. this object at 64 pairs)
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm*.json")), reverse=True)[0]
        with open(path) as f:
            kern = json.load(f)["kernels"]
        key = "pgd_step_shifted<1, 3, true>" if (use_index and verified) else "pgd_step_shifted<1, 3, false>"
        traffic = kern[key]["hbm_bytes_per_launch"] * pairs / 64
        out["roofline"]["traffic"] = traffic
        out["roofline"]["traffic_source"] = os.path.basename(path)
        out["roofline"]["hbm_utilisation"] = traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    except Exception:
        out["roofline"]["traffic"] = None
    if fl is not None:
        out["float_path"] = fl
    return out


if __name__ == "__main__":
    main()
