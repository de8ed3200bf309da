#!/usr/bin/env python3
"""Per-kernel timings of the secondary kernels (everything except the PGD step that bench.py measures):
affine maps, stand-alone export, patch paste / delta / apply, PSV cost-volume forward / backward.
HIP events on torch's current stream; prints one JSON line per kernel with algorithmic GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402



import contextlib


@contextlib.contextmanager
def hooks_route(**env):
    """A/B routes live in the -DADV_TEST_HOOKS build only (libadvengine_hooks.so): inside the block the calls go there, with the
    switches in the environment; the shipped library reads none"""
    import os
    from eval_driving_safety_amd import _lib
    with _lib.using(_lib.HOOKS_LIB_PATH):
        os.environ.update(env)
        try:
            yield
        finally:
            for k in env:
                del os.environ[k]

def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def line(name, ms, nbytes, **kw):
    print(json.dumps(dict(kernel=name, ms=round(ms, 4), algorithmic_GB=round(nbytes / 1e9, 4),
                          GBps=round(nbytes / ms / 1e6, 1), frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 3), **kw)))


DSGN_MEAN = (0.485, 0.456, 0.406)
DSGN_STD = (0.229, 0.224, 0.225)


def torch_eager_pgd_step(x, grad, clean, alpha, eps):
    """The reference's own formulation (attack/DSGN/pgd_attack.py:339-354) as PyTorch-ROCm eager ops on the GPU,
    made batch-correct: what running the reference script on this GPU would launch per step (baseline only)."""
    d = x.clone()
    for c in range(3):
        d[:, c] = d[:, c] * DSGN_STD[c] + DSGN_MEAN[c]
    adv = d + alpha * grad.sign()
    eta = torch.clamp(adv - clean, min=-eps, max=eps)
    y = torch.clamp(clean + eta, min=0, max=1)
    for c in range(3):
        y[:, c] = (y[:, c] - DSGN_MEAN[c]) / DSGN_STD[c]
    return y.detach()


def pgd_vs_eager(dev):
    H, W = 384, 1248
    sp = ops.Space.dsgn()
    for n in (2, 64, 512):                       # one pair (the reference's batch), 32 pairs, 256 pairs
        x = torch.randn((n, 3, H, W), device=dev)
        g = torch.randn_like(x)
        clean = torch.rand_like(x)
        nbytes = 16 * x.numel()
        reps = 200 if n == 2 else 20
        t_hip = timeit(lambda: ops.pgd_step(x, g, clean, sp, 1 / 255, 0.03, out=x), reps=reps)
        line("pgd_step HIP (in place, no export)", t_hip, nbytes, images=n)
        if n <= 64:
            t_eager = timeit(lambda: torch_eager_pgd_step(x, g, clean, 1 / 255, 0.03), reps=max(5, reps // 4))
            line("pgd_step PyTorch-ROCm eager (reference formulation)", t_eager, nbytes, images=n, speedup_of_hip=round(t_eager / t_hip, 2))
        del x, g, clean
        torch.cuda.empty_cache()


def pcie(dev):
    """what a caller pays if the stereo pairs start in (pinned) host memory and the 8-bit iterates end there:
    measured H2D / D2H rates and the resulting PCIe-inclusive bound in pairs/s for the 20-step attack"""
    import time
    pairs = 32
    n = 2 * pairs
    H, W = 384, 1248
    host_x = torch.empty((n, 3, H, W), dtype=torch.float32, pin_memory=True)
    dev_x = torch.empty((n, 3, H, W), dtype=torch.float32, device=dev)
    host_u8 = torch.empty((n, 375, W, 3), dtype=torch.uint8, pin_memory=True)
    dev_u8 = torch.empty((n, 375, W, 3), dtype=torch.uint8, device=dev)
    out = {}
    for name, dst, src in (("h2d_f32", dev_x, host_x), ("d2h_u8", host_u8, dev_u8)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out[name + "_GBps"] = round(src.numel() * src.element_size() / dt / 1e9, 2)
    bytes_in = 2 * 3 * H * W * 4                    # both eyes, float32
    bytes_out = 21 * 2 * 375 * W * 3                # 21 exported iterates, whole rows
    t = bytes_in / (out["h2d_f32_GBps"] * 1e9) + bytes_out / (out["d2h_u8_GBps"] * 1e9)
    out.update(bytes_in_per_pair=bytes_in, bytes_out_per_pair=bytes_out, pcie_bound_pairs_per_s=round(1 / t, 1),
               note="transfers serialised; H2D and D2H can overlap each other and the kernels on separate streams")
    print(json.dumps(out))


def conv(dev):
    """the 3x3x3 convolutions applied to the DSGN-sized cost volume: MFMA kernel vs torch (MIOpen), TFLOP/s against the
    157.3 TFLOP/s float32 matrix peak"""
    import torch.nn.functional as F
    D, H, W = 48, 96, 312
    for cin, cout in ((64, 32), (32, 32), (32, 64)):
        x = torch.randn((1, cin, D, H, W), device=dev)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
        wp = ops.conv3d_k3_prep(wt)
        flops = 2.0 * cin * cout * 27 * D * H * W
        ms = timeit(lambda: ops.conv3d_k3(x, wp, cout), reps=10)
        F.conv3d(x, wt, padding=1)                   # MIOpen solver search outside the timing
        ms_t = timeit(lambda: F.conv3d(x, wt, padding=1), reps=10)
        print(json.dumps(dict(kernel="conv3d_k3 %d->%d on [1,%d,%d,%d,%d]" % (cin, cout, cin, D, H, W), ms=round(ms, 3),
                              TFLOPs=round(flops / ms / 1e9, 1), frac_of_157TF=round(flops / ms / 1e9 / 157.3, 3),
                              torch_miopen_ms=round(ms_t, 3), speedup_vs_miopen=round(ms_t / ms, 2))))
        del x, wt, wp
    # the hourglass's other layer types (round 2): stride 2 and transposed, against torch / MIOpen
    x = torch.randn((1, 32, D, H, W), device=dev)
    wd = torch.randn((64, 32, 3, 3, 3), device=dev) * 0.05
    wp = ops.conv3d_k3_s2_prep(wd)
    flops = 2.0 * 32 * 64 * 27 * (D // 2) * (H // 2) * (W // 2)
    wplain = ops.conv3d_k3_prep(wd)
    ms_direct = timeit(lambda: ops.conv3d_k3_s2(x, wplain, 64), reps=10)
    with hooks_route(ADV_CONV_S2_GENERIC="1"):
        ms_generic = timeit(lambda: ops.conv3d_k3_s2(x, wplain, 64), reps=3)
    ms_s2d = timeit(lambda: ops.space_to_depth2(x), reps=10)
    print(json.dumps(dict(kernel="space_to_depth2 [1,32,%d,%d,%d]" % (D, H, W), ms=round(ms_s2d, 3), GBps=round(2 * x.numel() * 4 / ms_s2d / 1e6, 1),
                          scalar_staging_strided_kernel_ms=round(ms_generic, 3))))
    print(json.dumps(dict(kernel="conv3d_k3 stride 2 DIRECT (two-channel stages, stride-2 LDS operand reads; conv3d_k3_s2_mfma), 32->64 on [1,32,%d,%d,%d]" % (D, H, W),
                          ms=round(ms_direct, 3), TFLOPs=round(flops / ms_direct / 1e9, 1), frac_of_157TF=round(flops / ms_direct / 1e9 / 157.3, 3))))
    ms = timeit(lambda: ops.conv3d_k3_s2(x, wp, 64), reps=10)
    F.conv3d(x, wd, stride=2, padding=1)
    ms_t = timeit(lambda: F.conv3d(x, wd, stride=2, padding=1), reps=10)
    print(json.dumps(dict(kernel="conv3d_k3 stride 2 (space-to-depth + masked stride-1 kernel), 32->64 on [1,32,%d,%d,%d]" % (D, H, W), ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1),
                          frac_of_157TF=round(flops / ms / 1e9 / 157.3, 3), torch_miopen_ms=round(ms_t, 3), speedup_vs_miopen=round(ms_t / ms, 2))))
    xs = torch.randn((1, 64, D // 2, H // 2, W // 2), device=dev)
    wu = torch.randn((64, 32, 3, 3, 3), device=dev) * 0.05
    classes = ops.conv_transpose3d_k3_s2_prep(wu)
    flops = 2.0 * 64 * 32 * 27 * (D // 2) * (H // 2) * (W // 2)                 # 27 multiply-adds per INPUT voxel
    ms = timeit(lambda: ops.conv_transpose3d_k3_s2(xs, classes, 32), reps=10)
    F.conv_transpose3d(xs, wu, stride=2, padding=1, output_padding=1)
    ms_t = timeit(lambda: F.conv_transpose3d(xs, wu, stride=2, padding=1, output_padding=1), reps=10)
    print(json.dumps(dict(kernel="conv_transpose3d_k3 stride 2 (8 masked classes), 64->32 on [1,64,%d,%d,%d]" % (D // 2, H // 2, W // 2), ms=round(ms, 3),
                          TFLOPs=round(flops / ms / 1e9, 1), frac_of_157TF=round(flops / ms / 1e9 / 157.3, 3), torch_miopen_ms=round(ms_t, 3),
                          speedup_vs_miopen=round(ms_t / ms, 2))))
    bias, skip = torch.randn((32,), device=dev), torch.randn((1, 32, D, H, W), device=dev)      # the hourglass's up layer: relu(up(x) + bias + skip)
    ms = timeit(lambda: ops.conv_transpose3d_k3_s2(xs, classes, 32, relu=True, bias=bias, residual=skip), reps=10)
    ms_t = timeit(lambda: F.relu(F.conv_transpose3d(xs, wu, bias, stride=2, padding=1, output_padding=1) + skip), reps=10)
    print(json.dumps(dict(kernel="conv_transpose3d_k3 stride 2 + bias + skip connection + ReLU in the epilogue, 64->32 on [1,64,%d,%d,%d]" % (D // 2, H // 2, W // 2),
                          ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1), torch_miopen_plus_add_relu_ms=round(ms_t, 3), speedup=round(ms_t / ms, 2))))


def conv_narrow(dev):
    """the 32 -> 1 score layer and its adjoint: narrow vector-ALU kernels vs the padded matrix kernel vs torch / MIOpen"""
    import os
    import torch.nn.functional as F
    D, H, W = 48, 96, 312
    x = torch.randn((1, 32, D, H, W), device=dev)
    wt = torch.randn((1, 32, 3, 3, 3), device=dev) * 0.05
    wp, wpt = ops.conv3d_k3_prep(wt), ops.conv3d_k3_prep(wt, transpose=True)
    ms = timeit(lambda: ops.conv3d_k3(x, wp, 1), reps=10)
    with hooks_route(ADV_CONV_NO_NARROW="1"):
        ms_pad = timeit(lambda: ops.conv3d_k3(x, wp, 1), reps=10)
    F.conv3d(x, wt, padding=1)
    ms_t = timeit(lambda: F.conv3d(x, wt, padding=1), reps=10)
    nbytes = x.numel() * 4 + D * H * W * 4
    print(json.dumps(dict(kernel="conv3d_k3 32->1 narrow_out<1> on [1,32,%d,%d,%d]" % (D, H, W), ms=round(ms, 3), GBps=round(nbytes / ms / 1e6, 1),
                          frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 3), padded_matrix_kernel_ms=round(ms_pad, 3), torch_miopen_ms=round(ms_t, 3))))
    g = torch.randn((1, 1, D, H, W), device=dev)
    ms = timeit(lambda: ops.conv3d_k3(g, wpt, 32), reps=10)
    torch.nn.grad.conv3d_input(x.shape, wt, g, padding=1)
    ms_t = timeit(lambda: torch.nn.grad.conv3d_input(x.shape, wt, g, padding=1), reps=10)
    print(json.dumps(dict(kernel="conv3d_k3 adjoint 1->32 narrow_in<1> on [1,1,%d,%d,%d]" % (D, H, W), ms=round(ms, 3), GBps=round(nbytes / ms / 1e6, 1),
                          frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 3), torch_miopen_ms=round(ms_t, 3))))


def roi(dev):
    """RoIAlign forward at the Stereo R-CNN shapes (600x1987 image, 256-channel FPN levels, 7x7 / 14x14 bins, 2x2 samples):
    paired 8-byte tap loads against four single gathers per sample"""
    import os
    import numpy as np
    rs = np.random.RandomState(0)
    for name, stride, pooled, n in (("P2 box head", 4, 7, 512), ("P3 box head", 8, 7, 512), ("P4 box head", 16, 7, 512), ("P2 keypoint head", 4, 14, 128)):
        h, w = (600 + stride - 1) // stride, (1987 + stride - 1) // stride
        feat = torch.randn((1, 256, h, w), device=dev)
        side = rs.uniform(7, 28, n) * stride                                  # what the FPN level assignment sends to a level
        x1, y1 = rs.uniform(0, 1987 - side), rs.uniform(0, 600 - side)
        rois = torch.tensor(np.stack([np.zeros(n), x1, y1, x1 + side, y1 + side * rs.uniform(0.5, 1.0, n)], 1).astype(np.float32), device=dev)
        ms = timeit(lambda: ops.roi_align(feat, rois, pooled, 1.0 / stride, 2), reps=20)
        with hooks_route(ADV_ROI_FWD_DIRECT="1"):
            ms_d = timeit(lambda: ops.roi_align(feat, rois, pooled, 1.0 / stride, 2), reps=20)
        print(json.dumps(dict(kernel="roi_align_fwd<paired loads> %s: %d rois x 256 ch on %dx%d, %dx%d bins" % (name, n, h, w, pooled, pooled), ms=round(ms, 4),
                              single_gathers_ms=round(ms_d, 4), speedup=round(ms_d / ms, 2))))


def volume(dev):
    """csrc/volume.hip at the DSGN sizes, against torch's unfused operators"""
    import torch.nn.functional as F
    D, h, w = 48, 96, 312
    up = (192, 384, 1248)
    cost = torch.randn((1, D, h, w), device=dev)
    zv = torch.linspace(2.0, 40.4, 192, device=dev)
    g = torch.randn((1, 384, 1248), device=dev)
    ms = timeit(lambda: ops.depth_regress(cost, zv, up, False, with_stats=True), reps=10)
    depth, stats = ops.depth_regress(cost, zv, up, False, with_stats=True)
    ms_b = timeit(lambda: ops.depth_regress_bwd(cost, zv, depth, stats, g, False), reps=10)

    def torch_fwd():
        u = F.interpolate(cost[:, None], size=up, mode="trilinear", align_corners=False)[:, 0]
        return (torch.softmax(u, 1) * zv.view(1, -1, 1, 1)).sum(1)

    ms_t = timeit(torch_fwd, reps=5)
    cr = cost.clone().requires_grad_(True)

    def torch_fb():
        u = F.interpolate(cr[:, None], size=up, mode="trilinear", align_corners=False)[:, 0]
        ((torch.softmax(u, 1) * zv.view(1, -1, 1, 1)).sum(1) * g).sum().backward()

    ms_tb = timeit(torch_fb, reps=5)
    print(json.dumps(dict(kernel="depth_regress fused [1,48,96,312] -> 192 planes -> [1,384,1248]", fwd_ms=round(ms, 4), bwd_ms=round(ms_b, 4),
                          torch_unfused_fwd_ms=round(ms_t, 3), torch_unfused_fwd_bwd_ms=round(ms_tb, 3),
                          speedup_fwd=round(ms_t / ms, 1), speedup_fwd_bwd=round(ms_tb / (ms + ms_b), 1))))
    C, zo, yo, xo = 32, 192, 20, 304
    vol = torch.randn((1, C, D, h, w), device=dev)
    net_grid = adapters_grid(dev)
    out_bytes = C * zo * yo * xo * 4
    ms = timeit(lambda: ops.grid_sample3d(vol, net_grid, True), reps=10)
    F.grid_sample(vol, net_grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    ms_t = timeit(lambda: F.grid_sample(vol, net_grid, mode="bilinear", padding_mode="zeros", align_corners=True), reps=10)
    t0 = timeit(lambda: ops.GridSamplePlan(net_grid, (D, h, w), True), reps=3, warm=1)
    plan = ops.GridSamplePlan(net_grid, (D, h, w), True)
    go = torch.randn((1, C, zo, yo, xo), device=dev)
    ms_b = timeit(lambda: ops.grid_sample3d_bwd(go, plan), reps=10)
    ms_b_nc = timeit(lambda: ops.grid_sample3d_bwd(go, plan, channels_last=False), reps=10)
    vr = vol.clone().requires_grad_(True)

    def torch_gs_fb():
        F.grid_sample(vr, net_grid, mode="bilinear", padding_mode="zeros", align_corners=True).backward(go)

    ms_tb = timeit(torch_gs_fb, reps=5)
    print(json.dumps(dict(kernel="grid_sample3d [1,32,48,96,312] -> [1,32,192,20,304] (DSGN PSV -> 3DGV)", fwd_ms=round(ms, 4),
                          fwd_GBps=round((out_bytes + vol.numel() * 4) / ms / 1e6, 1), torch_fwd_ms=round(ms_t, 4), bwd_gather_ms=round(ms_b, 4), bwd_gather_ncdhw_ms=round(ms_b_nc, 4),
                          torch_fwd_bwd_ms=round(ms_tb, 4), plan_build_ms_once_per_calibration=round(t0, 3),
                          plan_MB=round(plan.buf.numel() * 4 / 1e6, 1))))
    n = 2 * 192 * 304
    x = torch.randn((n, 1), device=dev)
    t = (torch.rand(n, device=dev) < 0.01).to(torch.int32)
    ms = timeit(lambda: ops.sigmoid_focal_loss(x, t, 2.0, 0.25, want_grad=True), reps=20)
    print(json.dumps(dict(kernel="sigmoid_focal_loss fwd+grad on %d logits" % n, ms=round(ms, 4))))


def adapters_grid(dev):
    from eval_driving_safety_amd import adapters
    net = adapters.PsvStereoAdapter(dev, seed=0, dsgn_head=True, mfma_conv=False)
    return net.gv_grid


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if "--volume" in sys.argv:
        volume(dev)
        return
    if "--roi" in sys.argv:
        roi(dev)
        return
    if "--conv-narrow" in sys.argv:
        conv_narrow(dev)
        return
    if "--conv" in sys.argv:
        conv(dev)
        return
    if "--pcie" in sys.argv:
        pcie(dev)
        return
    if "--eager" in sys.argv:
        pgd_vs_eager(dev)
        return
    H, W = 384, 1248
    # ---- PSV at the DSGN shape
    for B in (1, 4, 16):
        C, D, h, w = 32, 48, H // 4, W // 4
        left = torch.randn((B, C, h, w), device=dev)
        right = torch.randn((B, C, h, w), device=dev)
        depth = 2.0 + 0.8 * torch.arange(D, device=dev)
        shift = (721.5377 * 0.54 / depth / 4).round().to(torch.int32).repeat(B, 1).contiguous()
        cost = torch.empty((B, 2 * C, D, h, w), device=dev)
        vol = cost.numel() * 4
        feat = 2 * left.numel() * 4
        line("psv_build fwd", timeit(lambda: ops.psv_build(left, right, shift, out=cost)), vol + feat, B=B)
        g = torch.randn_like(cost)
        line("psv_build bwd", timeit(lambda: ops.psv_build_bwd(g, shift)), vol + feat, B=B)
        shift_f = (721.5377 * 0.54 / depth / 4).to(torch.float32).repeat(B, 1).contiguous()      # fractional disparities
        line("psv_build_lerp fwd", timeit(lambda: ops.psv_build_lerp(left, right, shift_f, out=cost)), vol + feat, B=B)
        line("psv_build_lerp bwd", timeit(lambda: ops.psv_build_lerp_bwd(g, shift_f)), vol + feat, B=B)
        del cost, g
    if "--psv" in sys.argv:
        return
    # ---- affine / export at 512 images
    n = 512
    sp = ops.Space.dsgn()
    x = torch.randn((n, 3, H, W), device=dev)
    out = torch.empty_like(x)
    E = x.numel() * 4
    line("denormalize", timeit(lambda: ops.denormalize(x, sp, out=out)), 2 * E, images=n)
    line("normalize", timeit(lambda: ops.normalize(x, sp, out=out)), 2 * E, images=n)
    u8 = ops.alloc_u8(n, 375, W, dev)
    line("export_u8", timeit(lambda: ops.export_u8(x, sp, (375, 1242), out=u8)), n * (3 * 375 * W * 4 + 3 * 375 * 1242), images=n)
    del out
    # ---- Stereo R-CNN shaped PGD step (config 3): 64 pairs of 600x1987
    ns = 128
    sps = ops.Space.srcnn()
    xs = torch.randn((ns, 3, 600, 1987), device=dev) * 50
    gs = torch.randn_like(xs)
    cs = xs.clone()
    u8s = ops.alloc_u8(ns, 600, 1987, dev)
    Es = xs.numel() * 4
    line("pgd_step srcnn (in place, no u8)", timeit(lambda: ops.pgd_step(xs, gs, cs, sps, 1.0, 7.65, out=xs)), 4 * Es, images=ns)
    line("pgd_step srcnn (in place, dense u8: flat 12-byte stores)", timeit(lambda: ops.pgd_step(xs, gs, cs, sps, 1.0, 7.65, out=xs, u8_out=u8s), reps=5),
         4 * Es + ns * 3 * 600 * 1987, images=ns)
    xo = torch.empty_like(xs)
    line("pgd_step srcnn (out of place, no u8)", timeit(lambda: ops.pgd_step(xs, gs, cs, sps, 1.0, 7.65, out=xo)), 4 * Es, images=ns)
    line("pgd_step srcnn (out of place, dense u8, line-aligned kernel)",
         timeit(lambda: ops.pgd_step(xs, gs, cs, sps, 1.0, 7.65, out=xo, u8_out=u8s), reps=5), 4 * Es + ns * 3 * 600 * 1987, images=ns)
    del xs, gs, cs, u8s, xo
    # ---- patch kernels, batch of 64 pairs
    B, r = 64, 38
    d = 2 * r + 1
    img = x[:2 * B]
    patch = torch.randn((1, 3, d, d), device=dev)
    c2 = torch.stack([torch.randint(160, 340, (2 * B,), device=dev), torch.randint(300, 1000, (2 * B,), device=dev)], 1).to(torch.int32).contiguous()
    line("patch_paste_batch", timeit(lambda: ops.patch_paste_batch(img, patch, c2, r)), 2 * B * 2 * 3 * d * d * 4, pairs=B)
    c3 = torch.stack([c2[:B, 0], c2[:B, 1], c2[:B, 1] - 64], 1).contiguous()
    gl, gr = img[:B], img[B:]
    line("patch_delta_batch", timeit(lambda: ops.patch_delta_batch(gl, gr, c3, r, 8 / 255)), B * 2 * 3 * d * d * 4 + 3 * d * d * 4, pairs=B)
    delta = torch.zeros((3, d, d), device=dev)
    line("patch_apply", timeit(lambda: ops.patch_apply(patch, delta)), 3 * 3 * d * d * 4)
    one = x[:1]
    line("patch_paste (1 image)", timeit(lambda: ops.patch_paste(one, patch, 200, 600, r)), 2 * 3 * d * d * 4)
    line("patch_update (1 pair)", timeit(lambda: ops.patch_update(patch, x[:1], x[1:2], 200, 600, 536, r, 8 / 255)), 4 * 3 * d * d * 4)


if __name__ == "__main__":
    main()
