"""Parity of the HIP library (through the C ABI, via eval_driving_safety_amd.ops) against
  (1) the golden vectors produced by executing the reference's own statements, and
  (2) the numpy oracle on fresh seeded inputs, ragged shapes, batches and edge cases.
Bit-exact everywhere: float32 results are compared as raw bytes (tolerance 0; the
north_star allows 1e-4 L-inf on perturbations - we do not need it)."""
import hashlib
import random

import numpy as np
import pytest

import synth
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from eval_driving_safety_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def same_bits(a, b, what="", any_nan=False):
    """raw-byte equality.  any_nan=True: a NaN matches a NaN of any sign / payload - an x86 host GENERATES
    0xFFC00000 for inf-inf or 0*inf where gfx950 generates 0x7FC00000; that is the only tolerated difference."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if any_nan and a.dtype == np.float32:
        a, b = a.copy(), b.copy()
        a[np.isnan(a)] = np.float32("nan")
        b[np.isnan(b)] = np.float32("nan")
    if a.tobytes() != b.tobytes():
        av, bv = a.reshape(-1), b.reshape(-1)
        if a.dtype == np.float32:
            bad = np.flatnonzero(av.view(np.uint32) != bv.view(np.uint32))
        else:
            bad = np.flatnonzero(av != bv)
        i = bad[0]
        raise AssertionError("%s: %d of %d elements differ; first at %d: got %r want %r"
                             % (what, bad.size, av.size, i, av[i], bv[i]))


# ------------------------------------------------------------------------------- golden: PGD
DSGN_PGD = ["dsgn_pgd_default", "dsgn_pgd_fgsm", "dsgn_pgd_cfg2", "dsgn_pgd_specials", "dsgn_pgd_ragged", "dsgn_pgd_padded"]
SRCNN_PGD = ["srcnn_pgd_default", "srcnn_pgd_cfg3", "srcnn_pgd_specials", "srcnn_pgd_zero_eps", "srcnn_pgd_zero_eps_tiny_alpha"]


@pytest.mark.parametrize("name", DSGN_PGD)
def test_golden_dsgn_pgd(name, golden, golden_index, ops):
    g, m = golden(name), golden_index["cases"][name]
    sp = ops.Space.dsgn()
    ch, cw = m["crop_h"], m["crop_w"]
    for eye in "LR":
        x = dev(g["x0" + eye])
        clean = ops.denormalize(x, sp)
        same_bits(host(clean), g["clean" + eye], "clean")
        same_bits(host(ops.export_u8(x, sp, (ch, cw)))[0, :, :cw], g["u8%s_0" % eye], "u8_0")
        for k in range(m["n_iter"]):
            u8 = ops.alloc_u8(1, ch, m["w"], x.device)
            x = ops.pgd_step(x, dev(g["g%s_%d" % (eye, k)]), clean, sp, m["alpha"], m["eps"], u8_out=u8, crop=(ch, cw))
            same_bits(host(x), g["x%s_%d" % (eye, k + 1)], "%s x%s_%d" % (name, eye, k + 1))
            same_bits(host(u8)[0, :, :cw], g["u8%s_%d" % (eye, k + 1)], "%s u8%s_%d" % (name, eye, k + 1))


@pytest.mark.parametrize("name", SRCNN_PGD)
def test_golden_srcnn_pgd(name, golden, golden_index, ops):
    g, m = golden(name), golden_index["cases"][name]
    sp = ops.Space.srcnn()
    for eye in "LR":
        x = dev(g["x0" + eye])
        clean = x.clone()
        for k in range(m["n_iter"]):
            u8 = ops.alloc_u8(1, m["h"], m["w"], x.device)
            x = ops.pgd_step(x, dev(g["g%s_%d" % (eye, k)]), clean, sp, m["alpha"], m["eps"], u8_out=u8)
            same_bits(host(x), g["x%s_%d" % (eye, k + 1)], "%s x%s_%d" % (name, eye, k + 1))
            # the float HWC image is reference-pinned; its 8-bit rounding follows OpenCV (oracle, unpinned)
            want = O.srcnn_export_u8(g["x%s_%d" % (eye, k + 1)][0])
            same_bits(host(u8)[0], want, "%s u8%s_%d" % (name, eye, k + 1))


def test_golden_fullsize_digests(golden_index, ops):
    """KITTI-shaped 384x1248 pair (DSGN) and 600x1987 (Stereo R-CNN): in-place stepping, digests."""
    m = golden_index["cases"]["dsgn_pgd_fullsize"]
    sp = ops.Space.dsgn()
    for eye, off in (("L", 0), ("R", 1)):
        x = dev(synth.dsgn_normalised(m["seed"] + off, m["h"], m["w"]))
        clean = ops.denormalize(x, sp)
        u8 = ops.alloc_u8(1, m["crop_h"], m["w"], x.device)
        for k in range(m["n_iter"]):
            g = dev(synth.gradient(1000 * m["seed"] + 2 * k + off, tuple(x.shape), m["grad_scale"]))
            ops.pgd_step(x, g, clean, sp, m["alpha"], m["eps"], out=x, u8_out=u8, crop=(m["crop_h"], m["crop_w"]))
            assert sha(host(x)) == m["digests"]["x%s_%d" % (eye, k + 1)]
            assert sha(host(u8)[0, :, :m["crop_w"]]) == m["digests"]["u8%s_%d" % (eye, k + 1)]
    m = golden_index["cases"]["srcnn_pgd_fullsize"]
    sp = ops.Space.srcnn()
    x = dev(synth.srcnn_meansub(m["seed"], m["h"], m["w"]))
    clean = x.clone()
    for k in range(m["n_iter"]):
        g = dev(synth.gradient(2000 * m["seed"] + 2 * k, tuple(x.shape), m["grad_scale"]))
        ops.pgd_step(x, g, clean, sp, m["alpha"], m["eps"], out=x)
        assert sha(host(x)) == m["digests"]["xL_%d" % (k + 1)]


# ------------------------------------------------------------------------------- golden: patch
PATCH = ["dsgn_patch_default", "dsgn_patch_zero", "dsgn_patch_100px", "srcnn_patch_default"]


@pytest.mark.parametrize("name", PATCH)
def test_golden_patch(name, golden, golden_index, ops):
    g, m = golden(name), golden_index["cases"][name]
    dsgn = m["model"] == "dsgn"
    H, W, r = m["H"], m["W"], m["radius"]
    cy, cxl, cxr = m["center_l"][0], m["center_l"][1], m["center_r"][1]
    mk = synth.dsgn_normalised if dsgn else synth.srcnn_meansub
    xL, xR = dev(mk(m["seed"] + 10, H, W)), dev(mk(m["seed"] + 11, H, W))
    patch = dev(g["patch_0"])
    sp = ops.Space.dsgn() if dsgn else ops.Space.srcnn()
    lo, hi = (None, None) if dsgn else (sp.lo, sp.hi)
    assert sha(host(ops.disc_mask(H, W, cy, cxl, r, xL.device))) == m["digests"]["mask_l"]
    assert sha(host(ops.disc_mask(H, W, cy, cxr, r, xL.device))) == m["digests"]["mask_r"]
    gaccL = gaccR = None
    for k in range(m["iters"]):
        ops.patch_paste(xL, patch, cy, cxl, r)
        ops.patch_paste(xR, patch, cy, cxr, r)
        assert sha(host(xL)) == m["digests"]["pastedL_%d" % k], "pasted left image"
        assert sha(host(xR)) == m["digests"]["pastedR_%d" % k], "pasted right image"
        gl = dev(synth.gradient(3000 * m["seed"] + 2 * k, tuple(xL.shape), m["grad_scale"]))
        gr = dev(synth.gradient(3000 * m["seed"] + 2 * k + 1, tuple(xR.shape), m["grad_scale"]))
        gaccL = gl if gaccL is None else gaccL + gl      # autograd's accumulation into the same leaf
        gaccR = gr if gaccR is None else gaccR + gr
        assert sha(host(gaccL)) == m["digests"]["gradL_%d" % k]
        delta = torch.empty_like(patch[0])
        ops.patch_update(patch, gaccL, gaccR, cy, cxl, cxr, r, m["eps"], lo=lo, hi=hi, delta_out=delta)
        same_bits(host(patch), g["patch_%d" % (k + 1)], "%s patch_%d" % (name, k + 1))
        same_bits(host(delta)[None], O.patch_delta(host(gaccL), host(gaccR), cy, cxl, cxr, r, m["eps"]), "delta_out")


# ------------------------------------------------------------------------------- oracle: shapes / batches / paths
@pytest.mark.parametrize("kind", ["dsgn", "srcnn"])
@pytest.mark.parametrize("shape", [(1, 3, 8, 12), (3, 3, 5, 7), (2, 3, 9, 10), (4, 3, 64, 96), (1, 3, 1, 1), (2, 3, 3, 4),
                                   (3, 3, 6, 10), (2, 3, 10, 26), (5, 3, 20, 66), (2, 3, 5, 12), (3, 3, 33, 100), (2, 3, 600, 1987)])
def test_pgd_batches_and_ragged_shapes(kind, shape, ops):
    n, _, h, w = shape
    rs = np.random.RandomState(h * 131 + w)
    if kind == "dsgn":
        sp, step = ops.Space.dsgn(), O.pgd_step_norm01
        x = np.concatenate([synth.dsgn_normalised(100 + i, h, w) for i in range(n)])
        clean = O.denormalize(np.concatenate([synth.dsgn_normalised(200 + i, h, w) for i in range(n)]))
        alpha, eps = 2 / 255, 0.02
    else:
        sp, step = ops.Space.srcnn(), O.pgd_step_meansub255
        x = np.concatenate([synth.srcnn_meansub(100 + i, h, w) for i in range(n)])
        clean = np.concatenate([synth.srcnn_meansub(200 + i, h, w) for i in range(n)])
        alpha, eps = 1.0, 7.65
    g = synth.gradient(int(rs.randint(1 << 30)), shape, 1.0, specials=True)
    want = step(x, g, clean, alpha, eps)
    got = ops.pgd_step(dev(x), dev(g), dev(clean), sp, alpha, eps)
    same_bits(host(got), want, "pgd %s %s" % (kind, shape))
    # dense cropped export (byte-store path) and pitched export (dword path when W % 4 == 0)
    ch, cw = max(1, h - 1), max(1, w - 1)
    dense = torch.zeros((n, ch, cw, 3), dtype=torch.uint8, device="cuda")
    pitched = ops.alloc_u8(n, ch, w, "cuda")
    ops.pgd_step(dev(x), dev(g), dev(clean), sp, alpha, eps, u8_out=dense, crop=(ch, cw))
    ops.pgd_step(dev(x), dev(g), dev(clean), sp, alpha, eps, u8_out=pitched, crop=(ch, cw))
    exp = O.tensor2im_u8 if kind == "dsgn" else (lambda a, hh, ww: O.srcnn_export_u8(a)[:hh, :ww])
    for i in range(n):
        same_bits(host(dense)[i], exp(want[i], ch, cw), "dense u8")
        same_bits(host(pitched)[i, :, :cw], exp(want[i], ch, cw), "pitched u8")
        same_bits(host(ops.export_u8(dev(want), sp, (ch, cw)))[i, :, :cw], exp(want[i], ch, cw), "export_u8")


@pytest.mark.parametrize("kind", ["dsgn", "srcnn"])
@pytest.mark.parametrize("offset_floats", [0, 4, 12, 28])
def test_pgd_planes_that_are_not_whole_cache_lines(kind, offset_floats, ops):
    """the shifted-tile kernel: every mix of in place / out of place, export on / off, buffers starting inside a
    128-byte line (all four with the same residue), planes of 65 and 1033 float4"""
    for h, w in ((10, 26), (37, 100), (4, 1033 * 1)):
        if (h * w) % 4:
            continue
        n = 3
        shape = (n, 3, h, w)
        numel = int(np.prod(shape))
        if kind == "dsgn":
            sp, step, alpha, eps = ops.Space.dsgn(), O.pgd_step_norm01, 2 / 255, 0.02
            x = np.concatenate([synth.dsgn_normalised(300 + i, h, w) for i in range(n)])
            clean = O.denormalize(np.concatenate([synth.dsgn_normalised(400 + i, h, w) for i in range(n)]))
            exp = O.tensor2im_u8
        else:
            sp, step, alpha, eps = ops.Space.srcnn(), O.pgd_step_meansub255, 1.0, 7.65
            x = np.concatenate([synth.srcnn_meansub(300 + i, h, w) for i in range(n)])
            clean = np.concatenate([synth.srcnn_meansub(400 + i, h, w) for i in range(n)])
            exp = lambda a, hh, ww: O.srcnn_export_u8(a)[:hh, :ww]
        g = synth.gradient(h * w + offset_floats, shape, 1.0, specials=True)
        want = step(x, g, clean, alpha, eps)

        def placed(a):
            buf = torch.zeros(numel + 64, dtype=torch.float32, device="cuda")
            v = buf[offset_floats:offset_floats + numel].view(shape)
            v.copy_(dev(a))
            return v

        for inplace in (False, True):
            for with_u8 in (False, True):
                tx, tg, tc = placed(x), placed(g), placed(clean)
                out = tx if inplace else placed(np.zeros(shape, np.float32))
                u8 = ops.alloc_u8(n, h, w, "cuda") if with_u8 else None
                ops.pgd_step(tx, tg, tc, sp, alpha, eps, out=out, u8_out=u8)
                same_bits(host(out), want, "%s %dx%d off %d inplace %s u8 %s" % (kind, h, w, offset_floats, inplace, with_u8))
                if with_u8:
                    for i in range(n):
                        same_bits(host(u8)[i], exp(want[i], h, w), "u8 %s %dx%d off %d inplace %s" % (kind, h, w, offset_floats, inplace))


def test_pgd_unaligned_pointers_take_the_scalar_path(ops):
    """a view starting 4 bytes into an allocation is not 16-byte aligned"""
    shape = (2, 3, 16, 24)
    numel = int(np.prod(shape))
    x = synth.dsgn_normalised(1, 16, 24).repeat(2, axis=0)
    g = synth.gradient(2, shape)
    clean = O.denormalize(synth.dsgn_normalised(3, 16, 24).repeat(2, axis=0))
    want = O.pgd_step_norm01(x, g, clean, 1 / 255, 0.03)

    def shifted(a):
        buf = torch.zeros(numel + 1, dtype=torch.float32, device="cuda")
        v = buf[1:].view(shape)
        v.copy_(dev(a))
        assert v.data_ptr() % 16 != 0
        return v

    out = shifted(np.zeros(shape, np.float32))
    ops.pgd_step(shifted(x), shifted(g), shifted(clean), ops.Space.dsgn(), 1 / 255, 0.03, out=out)
    same_bits(host(out), want, "unaligned")


def test_normalize_denormalize(ops):
    sp = ops.Space.dsgn()
    for shape in [(2, 3, 6, 10), (1, 3, 5, 7)]:
        x = (np.random.RandomState(0).randn(*shape) * 3).astype(np.float32)
        same_bits(host(ops.denormalize(dev(x), sp)), O.denormalize(x), "denormalize")
        same_bits(host(ops.normalize(dev(x), sp)), O.normalize(x), "normalize")
        t = dev(x)
        ops.normalize(t, sp, out=t)
        same_bits(host(t), O.normalize(x), "normalize in place")


def test_patch_batch_forms_match_per_image_oracle(ops):
    n, h, w, r = 5, 96, 160, 9
    d = 2 * r + 1
    rs = np.random.RandomState(5)
    img = np.concatenate([synth.dsgn_normalised(40 + i, h, w) for i in range(n)])
    patch = synth.patch_init(9, d)
    cy = rs.randint(r, h - r, size=n)
    cxl = rs.randint(r + 20, w - r, size=n)
    cxr = cxl - 20
    t = dev(img)
    ops.patch_paste_batch(t, dev(patch), dev(np.stack([cy, cxl], 1).astype(np.int32)), r)
    for i in range(n):
        same_bits(host(t)[i:i + 1], O.patch_paste(img[i:i + 1], patch, int(cy[i]), int(cxl[i]), r), "paste %d" % i)
    gl = synth.gradient(77, img.shape, 5e-5)
    gr = synth.gradient(78, img.shape, 5e-5)
    centers = dev(np.stack([cy, cxl, cxr], 1).astype(np.int32))
    delta = ops.patch_delta_batch(dev(gl), dev(gr), centers, r, 8 / 255)
    want = None
    for i in range(n):
        di = O.patch_delta(gl[i:i + 1], gr[i:i + 1], int(cy[i]), int(cxl[i]), int(cxr[i]), r, 8 / 255)
        want = di if want is None else want + di
    same_bits(host(delta)[None], want, "summed delta")
    p = dev(patch)
    ops.patch_apply(p, delta)
    same_bits(host(p), O.patch_apply_delta(patch, want), "apply")
    p = dev(synth.patch_init(10, d, -140, 170))
    ops.patch_apply(p, delta, lo=O.SRCNN_LO, hi=O.SRCNN_HI)
    same_bits(host(p), O.patch_apply_delta(synth.patch_init(10, d, -140, 170), want, O.SRCNN_LO, O.SRCNN_HI), "apply+clamp")


def test_paste_is_idempotent_and_touches_only_the_square(ops):
    h, w, r = 384, 1248, 38
    random.seed(3)
    (cy, cx), _ = O.round_mask_centers(random, h, w, r)
    img = synth.dsgn_normalised(8, h, w)
    patch = synth.patch_init(1, 2 * r + 1)
    t = dev(img)
    ops.patch_paste(t, dev(patch), cy, cx, r)
    once = host(t).copy()
    ops.patch_paste(t, dev(patch), cy, cx, r)
    same_bits(host(t), once, "idempotent")
    outside = np.ones((h, w), bool)
    outside[cy - r:cy + r + 1, cx - r:cx + r + 1] = False
    assert np.array_equal(once[0][:, outside], img[0][:, outside])
    m = O.disc_mask(h, w, cy, cx, r).astype(bool)
    assert np.array_equal(once[0][:, m], O.patch_paste(img, patch, cy, cx, r)[0][:, m])


def test_argument_errors_are_reported_not_launched(ops):
    from eval_driving_safety_amd._lib import AdvEngineError
    sp = ops.Space.dsgn()
    x = torch.zeros((1, 3, 8, 8), device="cuda")
    with pytest.raises(AdvEngineError):
        ops.pgd_step(x, x, x, sp, 0.1, -1.0)                    # negative eps
    with pytest.raises(AdvEngineError):
        ops.pgd_step(x, x, x, sp, 0.1, float("nan"))
    with pytest.raises(AdvEngineError):
        ops.patch_paste(x, torch.zeros((3, 5, 5), device="cuda"), 1, 4, 2)   # window leaves the image
    with pytest.raises(AdvEngineError):
        ops.denormalize(x, ops.Space.srcnn())                   # identity space has no affine map
    with pytest.raises(TypeError):
        ops.pgd_step(x.cpu(), x, x, sp, 0.1, 0.1)               # no CPU path
    with pytest.raises(TypeError):
        ops.pgd_step(x.double(), x, x, sp, 0.1, 0.1)


# ------------------------------------------------------------------------------- BASELINE-size properties
def test_config2_size_properties(ops):
    """20-step PGD, eps 0.03, alpha 1/255 on 64 KITTI-shaped pairs (128 images, ~3 GB resident):
    size-independent invariants of the projection, and agreement of a sampled image with the oracle."""
    n, h, w = 128, 384, 1248
    sp = ops.Space.dsgn()
    gen = torch.Generator(device="cuda").manual_seed(0)
    u8src = torch.randint(0, 256, (n, 3, h, w), device="cuda", generator=gen, dtype=torch.int32)
    x0 = (u8src.float() / 255.0)
    ops.normalize(x0, sp, out=x0)
    clean = ops.denormalize(x0, sp)
    x = x0.clone()
    alpha, eps = 1 / 255, 0.03
    probe = [0, 57, 127]
    xs = {i: host(x[i:i + 1]).copy() for i in probe}
    for k in range(20):
        g = torch.randn((n, 3, h, w), device="cuda", generator=gen)
        for i in probe:
            xs[i] = O.pgd_step_norm01(xs[i], host(g[i:i + 1]), host(clean[i:i + 1]), alpha, eps)
        ops.pgd_step(x, g, clean, sp, alpha, eps, out=x)
    for i in probe:
        same_bits(host(x[i:i + 1]), xs[i], "image %d after 20 steps" % i)
    d = ops.denormalize(x, sp)
    lim = np.float32(eps)
    # the projection is applied in pixel space BEFORE re-normalising: allow the normalise/denormalise
    # round trip (measured 6e-8 in the reference) on top of eps
    assert float((d - clean).abs().max()) <= float(lim) + 2e-7
    assert float(d.min()) >= -2e-7 and float(d.max()) <= 1 + 2e-7
    # with random-sign gradients most pixels have moved, none stayed beyond the ball
    assert float((d - clean).abs().mean()) > 1e-3


def test_calls_are_capturable_in_a_hip_graph(ops):
    """the header promises enqueue-only entry points: capture a 3-step PGD + paste + patch update sequence with
    torch's stream capture, replay it twice on fresh inputs, compare with the oracle"""
    h, w, r = 96, 160, 9
    sp = ops.Space.dsgn()
    x_np = np.concatenate([synth.dsgn_normalised(60, h, w), synth.dsgn_normalised(61, h, w)])
    g_np = synth.gradient(62, x_np.shape, 1.0)
    clean_np = O.denormalize(x_np)
    patch_np = synth.patch_init(63, 2 * r + 1)
    x, g, clean, patch = dev(x_np), dev(g_np), dev(clean_np), dev(patch_np)
    u8 = ops.alloc_u8(2, h, w, "cuda")
    static_x = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # warm-up outside capture, as torch recommends
        ops.pgd_step(static_x, g, clean, sp, 1 / 255, 0.03, out=static_x, u8_out=u8)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(3):
            ops.pgd_step(static_x, g, clean, sp, 1 / 255, 0.03, out=static_x, u8_out=u8)
        ops.patch_paste(static_x[0:1], patch, 40, 70, r)
        ops.patch_update(patch, g[0:1], g[1:2], 40, 70, 50, r, 8 / 255)
    for trial in range(2):
        static_x.copy_(x)
        patch.copy_(dev(patch_np))
        graph.replay()
        torch.cuda.synchronize()
        want = x_np
        for _ in range(3):
            want = O.pgd_step_norm01(want, g_np, clean_np, 1 / 255, 0.03)
        same_bits(host(u8)[1], O.tensor2im_u8(want[1], h, w), "u8 from the graph, trial %d" % trial)
        want = want.copy()
        want[0:1] = O.patch_paste(want[0:1], patch_np, 40, 70, r)
        same_bits(host(static_x), want, "iterate from the graph, trial %d" % trial)
        same_bits(host(patch), O.patch_update(patch_np, g_np[0:1], g_np[1:2], 40, 70, 50, r, 8 / 255), "patch from the graph")


def test_extreme_float_values(ops):
    """denormals, huge magnitudes, infinities and NaN in every operand; alpha / eps of 0 and of denormal size.
    torch-CPU (and the numpy oracle) keep float32 denormals; so must the kernels (no flush-to-zero)."""
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -3e-39, 1.17549435e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf,
                         np.nan, 1.0, -1.0, 0.5, 0.485, 0.229, 2.6399999, -2.1179039, 1e-7, 255.0, -122.7717], dtype=np.float32)
    rs = np.random.RandomState(0)
    n = specials.size
    h, w = 8, n * 2
    grid = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2)
    with np.errstate(all="ignore"):
        for kind in ("dsgn", "srcnn"):
            for alpha, eps in ((0.0, 0.0), (1e-45, 1e-45), (1 / 255, 0.03), (3e38, 3e38), (1.0, 0.0)):
                x = specials[rs.randint(0, n, size=(2, 3, h, w))]
                g = specials[rs.randint(0, n, size=(2, 3, h, w))]
                cl = specials[rs.randint(0, n, size=(2, 3, h, w))]
                x.reshape(-1)[:grid.shape[0]] = specials[grid[:, 0]][: x.size]
                cl.reshape(-1)[:grid.shape[0]] = specials[grid[:, 1]][: x.size]
                if kind == "dsgn":
                    want = O.pgd_step_norm01(x, g, cl, alpha, eps)
                    got = ops.pgd_step(dev(x), dev(g), dev(cl), ops.Space.dsgn(), alpha, eps)
                    same_bits(host(ops.denormalize(dev(x), ops.Space.dsgn())), O.denormalize(x), "denormalize extremes", any_nan=True)
                    same_bits(host(ops.normalize(dev(x), ops.Space.dsgn())), O.normalize(x), "normalize extremes", any_nan=True)
                else:
                    want = O.pgd_step_meansub255(x, g, cl, alpha, eps)
                    got = ops.pgd_step(dev(x), dev(g), dev(cl), ops.Space.srcnn(), alpha, eps)
                same_bits(host(got), want, "%s alpha %g eps %g" % (kind, alpha, eps), any_nan=True)
    # and the patch kernels
    patch = specials[rs.randint(0, n, size=(1, 3, 5, 5))]
    img = specials[rs.randint(0, n, size=(1, 3, 12, 16))]
    t = dev(img)
    ops.patch_paste(t, dev(patch), 5, 7, 2)
    with np.errstate(all="ignore"):
        want = O.patch_paste(img, patch, 5, 7, 2)
    win = (slice(None), slice(None), slice(3, 8), slice(5, 10))
    same_bits(host(t)[win], want[win], "paste extremes (bounding square)", any_nan=True)
    gl = specials[rs.randint(0, n, size=(1, 3, 12, 16))]
    gr = specials[rs.randint(0, n, size=(1, 3, 12, 16))]
    p = dev(patch)
    ops.patch_update(p, dev(gl), dev(gr), 5, 7, 4, 2, 8 / 255, lo=O.SRCNN_LO, hi=O.SRCNN_HI)
    with np.errstate(all="ignore"):
        same_bits(host(p), O.patch_update(patch, gl, gr, 5, 7, 4, 2, 8 / 255, lo=O.SRCNN_LO, hi=O.SRCNN_HI), "update extremes", any_nan=True)


def test_more_images_than_grid_rows(ops):
    """n > 65535 images: grid.y saturates and the kernels loop over images"""
    n, h, w = 70001, 1, 8
    rs = np.random.RandomState(1)
    x = (rs.randn(n, 3, h, w)).astype(np.float32)
    g = rs.randn(n, 3, h, w).astype(np.float32)
    clean = rs.rand(n, 3, h, w).astype(np.float32)
    sp = ops.Space.dsgn()
    u8 = ops.alloc_u8(n, h, w, "cuda")
    got = ops.pgd_step(dev(x), dev(g), dev(clean), sp, 1 / 255, 0.03, u8_out=u8)
    want = O.pgd_step_norm01(x, g, clean, 1 / 255, 0.03)
    same_bits(host(got), want, "70001 images")
    for i in (0, 65534, 65535, 65536, 70000):
        same_bits(host(u8)[i], O.tensor2im_u8(want[i], h, w), "u8 of image %d" % i)
    same_bits(host(ops.denormalize(dev(x), sp)), O.denormalize(x), "denormalize 70001 images")
    # Stereo R-CNN space with planes of 2 float4 (not whole cache lines): the shifted kernel's image loop
    xs = (x * 50).astype(np.float32)
    got = ops.pgd_step(dev(xs), dev(g), dev(xs), ops.Space.srcnn(), 1.0, 7.65, u8_out=u8)
    want = O.pgd_step_meansub255(xs, g, xs, 1.0, 7.65)
    same_bits(host(got), want, "70001 images, shifted kernel")
    same_bits(host(u8)[69999], O.srcnn_export_u8(want[69999]), "u8 shifted kernel")


def test_patch_windows_touching_the_image_border(ops):
    h, w, r = 40, 56, 6
    d = 2 * r + 1
    img = synth.dsgn_normalised(70, h, w)
    patch = synth.patch_init(71, d)
    gl, gr = synth.gradient(72, img.shape, 5e-5), synth.gradient(73, img.shape, 5e-5)
    for cy, cx in ((r, r), (h - 1 - r, w - 1 - r), (r, w - 1 - r), (h - 1 - r, r)):
        t = dev(img)
        ops.patch_paste(t, dev(patch), cy, cx, r)
        same_bits(host(t), O.patch_paste(img, patch, cy, cx, r), "paste at (%d,%d)" % (cy, cx))
        p = dev(patch)
        ops.patch_update(p, dev(gl), dev(gr), cy, cx, cx, r, 8 / 255)
        same_bits(host(p), O.patch_update(patch, gl, gr, cy, cx, cx, r, 8 / 255), "update at (%d,%d)" % (cy, cx))
    from eval_driving_safety_amd._lib import AdvEngineError
    for cy, cx in ((r - 1, r), (r, w - r), (h - r, r)):
        with pytest.raises(AdvEngineError):
            ops.patch_paste(dev(img), dev(patch), cy, cx, r)


def _make_input(m, off=0):
    if m.get("padded"):
        return synth.dsgn_padded(m["seed"] + off, m["crop_h"], m["crop_w"], m["h"], m["w"])
    return synth.dsgn_normalised(m["seed"] + off, m["h"], m["w"])


@pytest.mark.parametrize("name", DSGN_PGD + ["dsgn_pgd_fullsize", "dsgn_pgd_fullsize_padded"])
def test_indexed_clean_image_path(name, golden, golden_index, ops):
    """the clean image held as one byte per element: verified per image on the device, results unchanged - on
    whole-frame 8-bit images and on images zero-padded in normalised space as the DSGN loader pads them"""
    sp = ops.Space.dsgn()
    m = golden_index["cases"][name]
    valid = (m["crop_h"], m["crop_w"]) if m.get("padded") else None
    if name.startswith("dsgn_pgd_fullsize"):
        for eye, off in (("L", 0), ("R", 1)):
            x = dev(_make_input(m, off))
            u8 = ops.alloc_u8(1, m["crop_h"], m["w"], x.device)
            clean, ci = ops.denormalize_indexed(x, sp, valid=valid, u8_out=u8, crop=(m["crop_h"], m["crop_w"]))
            assert ci.verified() == [True], "8-bit derived input must verify"
            same_bits(host(u8), host(ops.export_u8(x, sp, (m["crop_h"], m["crop_w"]))), "fused iterate-0 export")
            if m.get("padded"):    # without the padding rule the same image must NOT verify (0.485 is not an 8-bit level)
                assert ops.denormalize_indexed(x, sp)[1].verified() == [False]
            spare = torch.empty_like(x)
            for k in range(m["n_iter"]):
                g = dev(synth.gradient(1000 * m["seed"] + 2 * k + off, tuple(x.shape), m["grad_scale"]))
                ops.pgd_step(x, g, clean, sp, m["alpha"], m["eps"], out=spare, u8_out=u8, crop=(m["crop_h"], m["crop_w"]), clean_index=ci)
                x, spare = spare, x
                assert sha(host(x)) == m["digests"]["x%s_%d" % (eye, k + 1)]
                assert sha(host(u8)[0, :, :m["crop_w"]]) == m["digests"]["u8%s_%d" % (eye, k + 1)]
        return
    g = golden(name)
    if m["w"] % 4:
        assert not ops.can_index_clean(dev(g["x0L"]), sp)
        with pytest.raises(Exception):
            ops.denormalize_indexed(dev(g["x0L"]), sp)
        return
    x = dev(g["x0L"])
    u8 = ops.alloc_u8(1, m["crop_h"], m["w"], x.device)
    clean, ci = ops.denormalize_indexed(x, sp, valid=valid, u8_out=u8, crop=(m["crop_h"], m["crop_w"]))
    same_bits(host(clean), g["cleanL"], "clean")
    same_bits(host(u8)[0, :, :m["crop_w"]], g["u8L_0"], "iterate-0 export fused into the index build")
    assert ci.verified() == [True]
    want_index = np.rint(g["cleanL"] * np.float32(255))
    if valid is not None:
        want_index[:, :, valid[0]:, :] = 0
        want_index[:, :, :, valid[1]:] = 0
    same_bits(host(ci.index).astype(np.float32), want_index, "index = round(clean*255) inside the valid corner, 0 outside")
    assert tuple(ci.lut.shape) == (2, 3, 256)
    same_bits(host(ci.lut)[1], host(ci.lut)[0], "affine spaces: one table, stored twice")
    lut = host(ci.lut)[0]
    for c in range(3):      # the table is the ToTensor -> Normalize -> denormalize chain of every 8-bit level
        v = np.arange(256, dtype=np.float32) / np.float32(255)
        t = (v - np.float32(synth.DSGN_MEAN[c])) / np.float32(synth.DSGN_STD[c])
        same_bits(lut[c], t * np.float32(synth.DSGN_STD[c]) + np.float32(synth.DSGN_MEAN[c]), "lut channel %d" % c)
    for k in range(m["n_iter"]):
        x = ops.pgd_step(x, dev(g["gL_%d" % k]), clean, sp, m["alpha"], m["eps"], u8_out=u8, crop=(m["crop_h"], m["crop_w"]), clean_index=ci)
        same_bits(host(x), g["xL_%d" % (k + 1)], "%s xL_%d (indexed)" % (name, k + 1))
        same_bits(host(u8)[0, :, :m["crop_w"]], g["u8L_%d" % (k + 1)], "%s u8L_%d (indexed)" % (name, k + 1))


def test_indexed_clean_flags_are_per_image(ops):
    """one image without an 8-bit origin (or with a dirty padding border) takes the float path; the others of the same
    launch stay on the index path; every image equals the oracle either way"""
    sp = ops.Space.dsgn()
    h, w, vh, vw = 24, 40, 21, 37
    rs = np.random.RandomState(3)
    imgs = [synth.dsgn_padded(20 + i, vh, vw, h, w) for i in range(5)]
    imgs[1] = rs.randn(1, 3, h, w).astype(np.float32)                  # arbitrary floats: no 8-bit origin
    imgs[3] = imgs[3].copy()
    imgs[3][0, 2, h - 1, w - 1] = np.float32(1e-3)                     # padding that is not the loader's zero
    one_off = imgs[4].copy()
    one_off[0, 1, 5, 7] = np.nextafter(one_off[0, 1, 5, 7], np.float32(10))   # one wrong element is enough
    imgs.append(one_off)
    x = np.concatenate(imgs)
    g = synth.gradient(4, x.shape)
    clean, ci = ops.denormalize_indexed(dev(x), sp, valid=(vh, vw))
    assert ci.verified() == [True, False, True, False, True, False]
    same_bits(host(clean), O.denormalize(x), "clean")
    want = x
    cur = dev(x)
    for k in range(3):
        cur = ops.pgd_step(cur, dev(g), clean, sp, 1 / 255, 0.03, clean_index=ci)
        want = O.pgd_step_norm01(want, g, O.denormalize(x), 1 / 255, 0.03)
        same_bits(host(cur), want, "mixed index / float batch, step %d" % k)
    # per-image valid corners (KITTI frames differ by a few pixels): [n,2] on the device
    sizes = [(21, 37), (24, 40), (19, 33), (21, 36)]
    x = np.concatenate([synth.dsgn_padded(30 + i, vh_, vw_, h, w) for i, (vh_, vw_) in enumerate(sizes)])
    clean, ci = ops.denormalize_indexed(dev(x), sp, valid=sizes)
    assert ci.verified() == [True] * 4 and isinstance(ci.valid, torch.Tensor)
    g = synth.gradient(5, x.shape)
    got = ops.pgd_step(dev(x), dev(g), clean, sp, 1 / 255, 0.03, clean_index=ci)
    same_bits(host(got), O.pgd_step_norm01(x, g, O.denormalize(x), 1 / 255, 0.03), "per-image valid corners")
    # a valid corner that is too large for an image exposes its padding to the 8-bit check: that image falls back
    _, ci = ops.denormalize_indexed(dev(x), sp, valid=[(21, 37)] * 4)
    assert ci.verified() == [True, False, False, False]


def test_indexed_extreme_values_fall_back(ops):
    """NaN / inf / huge inputs never verify and never crash the index build"""
    sp = ops.Space.dsgn()
    x = synth.dsgn_normalised(3, 8, 16)
    for bad in (np.float32("nan"), np.float32("inf"), -np.float32("inf"), np.float32(3e38), np.float32(-1e-45)):
        y = x.copy()
        y[0, 0, 3, 5] = bad
        clean, ci = ops.denormalize_indexed(dev(y), sp)
        assert ci.verified() == [False]
        same_bits(host(clean), O.denormalize(y), "clean", any_nan=True)
        g = synth.gradient(6, y.shape)
        got = ops.pgd_step(dev(y), dev(g), clean, sp, 1 / 255, 0.03, clean_index=ci)
        same_bits(host(got), O.pgd_step_norm01(y, g, O.denormalize(y), 1 / 255, 0.03), "fallback", any_nan=True)


@pytest.mark.parametrize("shape", [(600, 1987), (20, 33), (12, 50), (30, 36)])
def test_srcnn_shape_in_place_with_export(shape, ops):
    """planes that are not whole cache lines: in-place update WITH the fused export (dense = aligned 12-byte stores,
    cropped = byte stores), several images per launch (every image has its own plane misalignment), vs the oracle"""
    h, w = shape
    sp = ops.Space.srcnn()
    n = 3
    x0 = np.concatenate([synth.srcnn_meansub(40 + i, h, w) for i in range(n)])
    g = synth.gradient(41, x0.shape, 1.0)
    want = x0
    x = dev(x0)
    clean = x.clone()
    u8 = ops.alloc_u8(n, h, w, x.device)
    u8c = ops.alloc_u8(n, h - 2, w, x.device)
    for k in range(3):
        want = O.pgd_step_meansub255(want, g, x0, 1.0, 7.65)
        if k == 1:      # cropped export: byte stores
            ops.pgd_step(x, dev(g), clean, sp, 1.0, 7.65, out=x, u8_out=u8c, crop=(h - 2, w - 3))
            for i in range(n):
                same_bits(host(u8c)[i, :, :w - 3], O.srcnn_export_u8(want[i])[:h - 2, :w - 3], "cropped export, image %d" % i)
        else:           # dense export, in place
            ops.pgd_step(x, dev(g), clean, sp, 1.0, 7.65, out=x, u8_out=u8)
            for i in range(n):
                same_bits(host(u8)[i], O.srcnn_export_u8(want[i]), "dense export, image %d step %d" % (i, k))
        same_bits(host(x), want, "in-place iterate %d" % k)
    same_bits(host(ops.export_u8(x, sp)), np.stack([O.srcnn_export_u8(want[i]) for i in range(n)]), "stand-alone dense export")


@pytest.mark.parametrize("shape", [(24, 51), (600, 1987), (12, 36)])
def test_srcnn_indexed_clean_image_equals_float_path_and_oracle(shape, ops):
    """Stereo R-CNN pixel space with the clean pair held as a verified 8-bit index: images whose loader subtracted the means in
    float32 (table A), in float64 rounded once (numpy's ``im -= pixel_means``: table B), and one image that is no 8-bit image
    (falls back to its float clean copy) - in one launch, in place, export fused, bit-identical to the oracle."""
    h, w = shape
    sp = ops.Space.srcnn()
    n = 10 if h < 600 else 8
    imgs = []
    for i in range(n):
        a = synth.srcnn_meansub(70 + i, h, w)                              # float32 arithmetic
        if i % 3 == 1:                                                     # float64 arithmetic, rounded once
            u8 = synth.u8_image(70 + i, h, w).transpose(2, 0, 1).astype(np.float64)
            a = (u8 - np.array(O.SRCNN_PIXEL_MEANS, np.float64).reshape(3, 1, 1)).astype(np.float32)[None]
        imgs.append(a)
    imgs[4] = imgs[4] * np.float32(0.999)                                  # not from 8-bit pixels
    x0 = np.concatenate(imgs)
    a_only = synth.srcnn_meansub(71, h, w)
    u8 = synth.u8_image(71, h, w).transpose(2, 0, 1).astype(np.float64)
    b_form = (u8 - np.array(O.SRCNN_PIXEL_MEANS, np.float64).reshape(3, 1, 1)).astype(np.float32)[None]
    assert a_only.tobytes() != b_form.tobytes(), "the two loader formulas must differ for this test to mean anything"
    g = synth.gradient(77, x0.shape, 1.0)
    x = dev(x0)
    assert ops.can_index_clean(x, sp)
    u8o = ops.alloc_u8(n, h, w, x.device)
    clean, cidx = ops.denormalize_indexed(x, sp, u8_out=u8o)
    same_bits(host(clean), x0, "clean pair = clone of x")
    flags = cidx.ok.cpu().tolist()
    for i in range(n):
        if i == 4:
            assert flags[i] == 0, "image 4 is not 8-bit derived"
        elif i % 3 == 1:
            assert flags[i] in (2, 3) and flags[i] & 2, "float64-subtracted image must match table B (image %d: %d)" % (i, flags[i])
        else:
            assert flags[i] & 1, "float32-subtracted image must match table A (image %d: %d)" % (i, flags[i])
        same_bits(host(u8o)[i], O.srcnn_export_u8(x0[i]), "iterate-0 export, image %d" % i)
    want = x0
    xf = dev(x0)
    cf = xf.clone()
    for k in range(3):
        want = O.pgd_step_meansub255(want, g, x0, 1.0, 7.65)
        ops.pgd_step(x, dev(g), clean, sp, 1.0, 7.65, out=x, u8_out=u8o, clean_index=cidx)
        ops.pgd_step(xf, dev(g), cf, sp, 1.0, 7.65, out=xf)
        same_bits(host(x), want, "indexed iterate %d vs oracle" % k)
        assert torch.equal(x, xf), "indexed vs float path, iterate %d" % k
        for i in (0, 1, 4, n - 1):
            same_bits(host(u8o)[i], O.srcnn_export_u8(want[i]), "export, image %d step %d" % (i, k))
    # an odd, small batch (the last workgroup row holds one image)
    x2 = dev(x0[:3])
    c2, ci2 = ops.denormalize_indexed(x2, sp)
    ops.pgd_step(x2, dev(g[:3]), c2, sp, 1.0, 7.65, out=x2, clean_index=ci2)
    same_bits(host(x2), O.pgd_step_meansub255(x0[:3], g[:3], x0[:3], 1.0, 7.65), "small batch through the indexed entry point")


@pytest.mark.parametrize("aligned_rows", [True, False])
def test_import_u8_equals_the_host_loader_transform(aligned_rows, ops):
    """ops.import_u8 (ToTensor + Normalize + zero padding on the device, from 8-bit HWC pixels) against data.dsgn_transform on the
    host: the same bits for x; clean = denormalize(x); the index is the pixels; images of different sizes in one buffer; then two
    indexed PGD steps equal the float path started from the host-transformed input"""
    from eval_driving_safety_amd import data
    sp = ops.Space.dsgn()
    H, W = 24, 40
    sizes = [(21, 37), (24, 40), (19, 33)] if aligned_rows else [(21, 35), (20, 38), (21, 38)]
    hb, wb = max(s[0] for s in sizes), max(s[1] for s in sizes)
    if aligned_rows:
        wb = (wb + 3) // 4 * 4                       # rows of the uint8 buffer 4-byte aligned: the 12-byte load path
    rs = np.random.RandomState(5)
    u8 = np.zeros((3, hb, wb, 3), np.uint8)
    want = []
    for i, (h, w) in enumerate(sizes):
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        u8[i, :h, :w] = img
        want.append(data.dsgn_transform(torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))), (H, W)).numpy())
    want = np.stack(want)
    x, clean, ci = ops.import_u8(dev(u8), sp, (H, W), valid=sizes)
    same_bits(host(x), want, "x = the host loader's transform")
    same_bits(host(clean), O.denormalize(want), "clean = denormalize(x)")
    assert ci.verified() == [True, True, True]
    idx = host(ci.index)
    for i, (h, w) in enumerate(sizes):
        assert np.array_equal(idx[i, :, :h, :w], u8[i, :h, :w].transpose(2, 0, 1)) and not idx[i, :, h:, :].any() and not idx[i, :, :, w:].any()
    _, ci2 = ops.denormalize_indexed(dev(want), sp, valid=sizes)        # the verified build agrees with the by-construction one
    assert torch.equal(ci2.index, ci.index) and ci2.verified() == [True, True, True]
    g = synth.gradient(9, want.shape, 1.0)
    xs, ref = x.clone(), want
    for _ in range(2):
        ops.pgd_step(xs, dev(g), clean, sp, 1 / 255, 0.03, out=xs, clean_index=ci)
        ref = O.pgd_step_norm01(ref, g, O.denormalize(want), 1 / 255, 0.03)
    same_bits(host(xs), ref, "two indexed steps from the imported batch")
    # a common size and no index: the plain transform
    x1, c1, none = ops.import_u8(dev(u8[:1, :sizes[0][0], :]).contiguous(), sp, (H, W), valid=(sizes[0][0], sizes[0][1]), want_clean=False, want_index=False)
    assert c1 is None and none is None
    same_bits(host(x1), want[:1], "import without clean image / index")


def test_indexed_equals_float_path_at_bench_scale(ops):
    """BASELINE configs[1] at the bench's own scale: 256 zero-padded KITTI-shaped pairs (512 images) built exactly as bench.py
    builds them; the indexed kernel and the all-float32 kernel (itself pinned by the golden vectors) must agree on every
    bit of every iterate and every export byte, all 512 images must verify, and sampled images must equal the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dev_ = torch.device("cuda", 0)
    sp = ops.Space.dsgn()
    pairs = 256
    gen = torch.Generator(device=dev_).manual_seed(77)
    x0, valid = bench.kitti_like_input(torch, ops, sp, pairs, dev_, gen, padded=True)
    assert valid == (375, 1242) and float(x0[:, :, 375:].abs().max()) == 0 and float(x0[:, :, :, 1242:].abs().max()) == 0
    grad = torch.randn(x0.shape, device=dev_, generator=gen)
    crop = (375, 1242)
    u8_i, u8_f = ops.alloc_u8(2 * pairs, 375, 1248, dev_), ops.alloc_u8(2 * pairs, 375, 1248, dev_)
    clean_i, ci = ops.denormalize_indexed(x0, sp, valid=valid, u8_out=u8_i, crop=crop)
    assert all(ci.verified()), "every loader-shaped image must take the index path"
    clean_f = ops.denormalize(x0, sp)
    assert torch.equal(clean_i, clean_f)
    ops.export_u8(x0, sp, crop, out=u8_f)
    assert torch.equal(u8_i[:, :, :1242], u8_f[:, :, :1242]), "iterate-0 export"
    xi, xf = x0.clone(), x0.clone()
    sample = [0, 255, 256, 511]
    want = host(x0[sample])
    clean_s, grad_s = O.denormalize(want), host(grad[sample])
    for k in range(3):
        ops.pgd_step(xi, grad, clean_i, sp, 1 / 255, 0.03, out=xi, u8_out=u8_i, crop=crop, clean_index=ci)
        ops.pgd_step(xf, grad, clean_f, sp, 1 / 255, 0.03, out=xf, u8_out=u8_f, crop=crop)
        assert torch.equal(xi, xf), "iterate %d differs between the indexed and the float32 kernel" % (k + 1)
        assert torch.equal(u8_i[:, :, :1242], u8_f[:, :, :1242]), "export %d" % (k + 1)
        want = O.pgd_step_norm01(want, grad_s, clean_s, 1 / 255, 0.03)
        same_bits(host(xi[sample]), want, "sampled images vs oracle, step %d" % (k + 1))


def test_gpu_reference_numerics_match_torch_on_this_gpu(ops):
    """Space.dsgn(reference_on_gpu=True): the reference's own formulation executed by torch ON THE GPU (where `tensor / std[c]`
    becomes a multiplication by the float32 reciprocal) is reproduced bit for bit - float path and indexed path alike - while
    the default space reproduces the CPU run (golden vectors) and stays within a few ulp of it"""
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    alpha, eps = 1.0 / 255, 0.03
    x_np = np.concatenate([synth.dsgn_padded(61, 21, 37, 24, 40), synth.dsgn_padded(62, 21, 37, 24, 40)])
    g = dev(synth.gradient(63, x_np.shape))

    def denormalize(im):                      # attack/DSGN/pgd_attack.py:196-200, batch-correct
        out = im.clone()
        for c in range(3):
            out[:, c] = out[:, c] * std[c] + mean[c]
        return out

    def normalize(im):                        # :203-207
        out = im.clone()
        for c in range(3):
            out[:, c] = (out[:, c] - mean[c]) / std[c]
        return out

    sp_gpu, sp_cpu = ops.Space.dsgn(reference_on_gpu=True), ops.Space.dsgn()
    x = dev(x_np)
    clean_t = denormalize(x)
    clean, ci = ops.denormalize_indexed(x, sp_gpu, valid=(21, 37))
    assert torch.equal(clean, clean_t) and ci.verified() == [True, True]
    cur_t, cur, cur_i, cur_c = x.clone(), x.clone(), x.clone(), x.clone()
    differs = False
    for k in range(4):
        d = denormalize(cur_t)                                                   # :339-354 as torch-ROCm eager ops
        adv = d + alpha * g.sign()
        eta = torch.clamp(adv - clean_t, min=-eps, max=eps)
        cur_t = normalize(torch.clamp(clean_t + eta, min=0, max=1)).detach()
        cur = ops.pgd_step(cur, g, clean, sp_gpu, alpha, eps)
        cur_i = ops.pgd_step(cur_i, g, clean, sp_gpu, alpha, eps, clean_index=ci)
        assert torch.equal(cur, cur_t), "float path vs torch on the GPU, step %d" % (k + 1)
        assert torch.equal(cur_i, cur_t), "indexed path vs torch on the GPU, step %d" % (k + 1)
        cur_c = ops.pgd_step(cur_c, g, clean, sp_cpu, alpha, eps)
        differs |= not torch.equal(cur_c, cur_t)
        assert float((cur_c - cur_t).abs().max()) <= 2e-6                        # a few ulp: the iterate feeds back
    assert differs, "the CPU-path and GPU-path re-normalisations should not be identical functions"
    assert torch.equal(ops.normalize(clean, sp_gpu), normalize(clean_t))


@pytest.mark.gpu
def test_pgd_step_beyond_2_31_elements_per_stream(ops):
    """maximum sizes: ONE launch over 1500 KITTI-shaped images - 2.16e9 elements (8.6 GB) per stream, so element offsets pass 2^31
    and byte offsets 2^33 - gives the bytes of the same work done in launches of 500 images whose offsets all fit 32 bits, on
    the all-float32 kernel and on the indexed-clean-image kernel, 8-bit export included; the LAST image (beyond every 32-bit
    boundary) also equals the oracle.  Skipped when the device has less than 80 GB free."""
    free, _ = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GB of free device memory")
    sp = ops.Space.dsgn()
    n, h, w, ch, cw = 1500, 384, 1248, 375, 1242
    assert n * 3 * h * w > 2 ** 31
    d = torch.device("cuda", 0)
    gen = torch.Generator(device=d).manual_seed(77)
    u = torch.randint(0, 256, (n, ch, cw, 3), device=d, generator=gen, dtype=torch.uint8)
    x, clean, ci = ops.import_u8(u, sp, (h, w))      # the loader's transform on the device (itself one launch beyond 2^31 elements)
    del u
    g = torch.empty_like(x)
    for i in range(0, n, 100):
        g[i:i + 100].normal_(generator=gen)
    assert torch.equal(ops.denormalize_indexed(x[-2:], sp, valid=(ch, cw))[0], clean[-2:])       # the import's clean image, re-derived
    alpha, eps = 1.0 / 255, 0.03
    for index in (None, ci):
        full = torch.empty_like(x)
        u8_full = ops.alloc_u8(n, ch, w, d)
        ops.pgd_step(x, g, clean, sp, alpha, eps, out=full, u8_out=u8_full, crop=(ch, cw), clean_index=index)
        for i in range(0, n, 500):
            s = slice(i, i + 500)
            sub = None if index is None else ops.denormalize_indexed(x[s], sp, valid=(ch, cw))[1]
            part = torch.empty_like(x[s])
            u8_part = ops.alloc_u8(500, ch, w, d)
            ops.pgd_step(x[s], g[s], clean[s], sp, alpha, eps, out=part, u8_out=u8_part, crop=(ch, cw), clean_index=sub)
            assert torch.equal(part, full[s]), (index is not None, i)
            assert torch.equal(u8_part[:, :, :cw], u8_full[s][:, :, :cw]), (index is not None, i)
            del part, u8_part, sub
        want = O.pgd_step_norm01(host(x[-1:]), host(g[-1:]), host(clean[-1:]), alpha, eps)
        same_bits(host(full[-1:]), want, "last image vs oracle")
        del full, u8_full


@pytest.mark.gpu
def test_srcnn_pgd_step_beyond_2_31_elements_per_stream(ops):
    """the same maximum-size check in the Stereo R-CNN pixel space (600x1987, identity space, in place, export fused): 604 images =
    2.16e9 elements per stream in ONE launch against launches of 151 images, float and indexed clean image; last image vs oracle."""
    free, _ = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GB of free device memory")
    sp = ops.Space.srcnn()
    n, h, w = 604, 600, 1987
    assert n * 3 * h * w > 2 ** 31
    d = torch.device("cuda", 0)
    gen = torch.Generator(device=d).manual_seed(78)
    means = torch.tensor(O.SRCNN_PIXEL_MEANS, dtype=torch.float32, device=d).view(1, 3, 1, 1)
    x0 = torch.empty((n, 3, h, w), dtype=torch.float32, device=d)
    g = torch.empty_like(x0)
    for i in range(0, n, 151):
        x0[i:i + 151] = torch.randint(0, 256, (151, 3, h, w), device=d, generator=gen, dtype=torch.uint8).float() - means   # float32 subtraction
        g[i:i + 151].normal_(generator=gen)
    clean, ci = ops.denormalize_indexed(x0, sp)
    assert all(ci.verified())
    for index in (None, ci):
        full = x0.clone()
        u8_full = ops.alloc_u8(n, h, w, d)
        ops.pgd_step(full, g, clean, sp, 1.0, 7.65, out=full, u8_out=u8_full, clean_index=index)
        for i in range(0, n, 151):
            s = slice(i, i + 151)
            part = x0[s].clone()
            u8_part = ops.alloc_u8(151, h, w, d)
            sub_clean, sub = ops.denormalize_indexed(part, sp)
            ops.pgd_step(part, g[s], sub_clean, sp, 1.0, 7.65, out=part, u8_out=u8_part, clean_index=None if index is None else sub)
            assert torch.equal(part, full[s]) and torch.equal(u8_part, u8_full[s]), (index is not None, i)
            del part, u8_part, sub_clean, sub
        same_bits(host(full[-1:]), O.pgd_step_meansub255(host(x0[-1:]), host(g[-1:]), host(x0[-1:]), 1.0, 7.65), "last image vs oracle")
        del full, u8_full
