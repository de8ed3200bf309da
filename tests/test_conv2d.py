"""csrc/conv2d.hip (2D convolutions of the detectors' backbones on the float32 matrix cores) against the oracle's fmaf chain (bit
for bit) and against torch's conv2d (1e-4).  The layers are upstream detector code (ResNet-101-FPN of Stereo R-CNN reached at
attack/Stereo-RCNN/pgd_attack.py:156, DSGN's 2D networks at attack/DSGN/pgd_attack.py:308): unpinned against upstream; what is
pinned is this package's arithmetic (oracle) and the operator's semantics (torch)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_c as C


def _case(b, cin, cout, h, w, k=1, seed=0):
    rs = np.random.RandomState(seed)
    return rs.randn(b, cin, h, w).astype(np.float32), (rs.randn(cout, cin, k, k) * (1.0 / (cin * k * k)) ** 0.5).astype(np.float32), rs


# (B, Cin, Cout, H, W): odd row lengths (the rule for feature maps), channel counts that are not multiples of the stage / tile,
# fewer pixels than a tile, the R101 shapes at reduced size
ONE = [(1, 16, 32, 5, 7), (2, 24, 18, 19, 63), (1, 3, 70, 9, 11), (2, 64, 256, 13, 41), (1, 256, 64, 10, 33), (1, 40, 3, 38, 125),
       (3, 130, 129, 7, 9), (1, 4, 4, 1, 1), (1, 1, 5, 2, 2), (1, 1024, 18, 6, 10)]


@pytest.mark.parametrize("shape", ONE[:6])
def test_oracle_conv2d_matches_torch(shape):
    b, cin, cout, h, w = shape
    x, wt, rs = _case(*shape, seed=sum(shape))
    bias = rs.randn(cout).astype(np.float32)
    for k, pad, stride, dil in ((1, 0, 1, 1), (3, 1, 1, 1), (3, 1, 2, 1), (1, 0, 2, 1), (3, 2, 1, 2)):
        wk = (rs.randn(cout, cin, k, k) * (1.0 / (cin * k * k)) ** 0.5).astype(np.float32)
        got = C.conv2d(x, wk, bias, stride=stride, padding=pad, relu=True, dilation=dil, chunk=8)
        ref = F.relu(F.conv2d(torch.tensor(x), torch.tensor(wk), torch.tensor(bias), stride, pad, dil)).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        if stride == 1:
            g = rs.randn(*ref.shape).astype(np.float32)
            gx = C.conv2d(g, wk, padding=pad, transpose=True, dilation=dil, chunk=8)
            refg = torch.nn.grad.conv2d_input(x.shape, torch.tensor(wk), torch.tensor(g), padding=pad, dilation=dil).numpy()
            assert np.abs(gx - refg).max() <= 1e-5 * max(1.0, np.abs(refg).max())


def test_oracle_conv2d_epilogue_order():
    """+ bias, + residual, ReLU, then the mask - each a separate float32 operation"""
    x, wt, rs = _case(1, 8, 5, 4, 6, seed=3)
    bias, res, mask = rs.randn(5).astype(np.float32), rs.randn(1, 5, 4, 6).astype(np.float32), rs.randn(1, 5, 4, 6).astype(np.float32)
    plain = C.conv2d(x, wt)
    want = np.maximum(plain + bias[None, :, None, None] + res, np.float32(0)) * (mask > 0)
    assert C.conv2d(x, wt, bias, res, mask, relu=True).tobytes() == (want + np.float32(0)).astype(np.float32).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ONE)
def test_hip_conv2d_1x1_bit_exact_vs_oracle_every_tile_shape(shape):
    from eval_driving_safety_amd import ops
    b, cin, cout, h, w = shape
    x, wt, rs = _case(*shape, seed=sum(shape) + 1)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = rs.randn(b, cout, h, w).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, wt, bias, res, mask))
    assert ops.conv2d_supported(tx, tw, 1, 0)
    prep = ops.Conv2dPrep(tw)
    want_plain = C.conv2d(x, wt)
    want_full = C.conv2d(x, wt, bias, res, mask, relu=True)
    for tile in (-1, 0, 1, 2, 3, 4, 5):
        assert ops.conv2d(tx, prep, tile=tile).cpu().numpy().tobytes() == want_plain.tobytes(), ("plain", tile)
        assert ops.conv2d(tx, prep, tb, tr, True, tm, tile=tile).cpu().numpy().tobytes() == want_full.tobytes(), ("bias + residual + relu + mask", tile)
    ref = F.conv2d(tx, tw, tb)
    got = ops.conv2d(tx, prep, tb)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    # the backward w.r.t. the input: the same kernel on W^T, with the skip-path gradient and the ReLU mask of the layer's input fused
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gres = rs.randn(b, cin, h, w).astype(np.float32)
    tg, tgr = torch.tensor(g, device=dev), torch.tensor(gres, device=dev)
    assert ops.conv2d_dgrad(tg, prep).cpu().numpy().tobytes() == C.conv2d(g, wt, transpose=True).tobytes()
    want_b = C.conv2d(g, wt, residual=gres, mask=x, transpose=True)
    for tile in (-1, 3, 4, 5):
        assert ops.conv2d_dgrad(tg, prep, residual=tgr, mask=tx, tile=tile).cpu().numpy().tobytes() == want_b.tobytes(), tile
    refg = torch.nn.grad.conv2d_input(x.shape, tw, tg)
    assert float((ops.conv2d_dgrad(tg, prep) - refg).abs().max()) <= 1e-4 * max(1.0, float(refg.abs().max()))


@pytest.mark.gpu
def test_hip_conv2d_refuses_tensors_of_fewer_than_four_floats():
    """the kernels load whole float4s, clamped into the tensor: a tensor of fewer than four floats has no such address"""
    from eval_driving_safety_amd import ops, _lib
    dev = torch.device("cuda", 0)
    for k, pad in ((1, 0), (3, 1)):
        x, wt = torch.randn((1, 3, 1, 1), device=dev), torch.randn((5, 3, k, k), device=dev)
        assert not ops.conv2d_supported(x, wt, 1, pad)
        prep = ops.Conv2dPrep(wt, 1, pad, 1)
        for kw in ({}, {"wino": True}) if k == 3 else ({},):
            with pytest.raises(_lib.AdvEngineError):
                ops.conv2d(x, prep, **kw)
    assert ops.conv2d_supported(torch.randn((1, 4, 1, 1), device=dev), torch.randn((4, 4, 1, 1), device=dev), 1, 0)
    assert not ops.conv2d_supported(torch.randn((1, 4, 1, 1), device=dev), torch.randn((1, 4, 1, 1), device=dev), 1, 0)    # the gradient has one float


@pytest.mark.gpu
def test_hip_conv2d_autograd_matches_torch():
    """ops.Conv2d inside autograd: relu(conv + bias + skip) forward, gradients to x and to the skip tensor"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((2, 64, 19, 63), device=dev, generator=gen, requires_grad=True)
    skip = torch.randn((2, 256, 19, 63), device=dev, generator=gen, requires_grad=True)
    wt = torch.randn((256, 64, 1, 1), device=dev, generator=gen) * 0.1
    bias = torch.randn((256,), device=dev, generator=gen)
    prep = ops.Conv2dPrep(wt)
    y = ops.Conv2d.apply(x, prep, bias, skip, True)
    ref = F.relu(F.conv2d(x, wt, bias) + skip)
    assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    gy = torch.randn(y.shape, device=dev, generator=gen)
    gx, gs = torch.autograd.grad(y, (x, skip), gy)
    rx, rsk = torch.autograd.grad(ref, (x, skip), gy)
    assert float((gx - rx).abs().max()) <= 1e-4 * float(rx.abs().max())
    assert torch.equal(gs * (ref > 0), rsk * (y > 0)) or float((gs - rsk).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_conv2d_auto_skip_out_adds_the_skip_gradient_in_the_dgrad_epilogue():
    """a residual block y = conv_b(relu(conv_a(x))) + x with Conv2dAuto(skip_out=True) on layer a (the skip path's gradient meets the
    convolution's in a's backward: no addition by the autograd engine) and the chained ReLU mask; x itself a ReLU output whose mask the
    block applies over both paths - against the same block written with torch's operators"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    for k, pad in ((3, 1), (1, 0)):
        pre = torch.randn((2, 24, 17, 38), device=dev, generator=gen, requires_grad=True)
        wa = torch.randn((24, 24, k, k), device=dev, generator=gen) * 0.1
        wb = torch.randn((24, 24, k, k), device=dev, generator=gen) * 0.1
        ba, bb = torch.randn((24,), device=dev, generator=gen), torch.randn((24,), device=dev, generator=gen)
        go = torch.randn((2, 24, 17, 38), device=dev, generator=gen)
        x = F.relu(pre)
        ref = F.conv2d(F.relu(F.conv2d(x, wa, ba, 1, pad)), wb, bb, 1, pad) + x
        (gref,) = torch.autograd.grad(ref, pre, go)
        pa, pb = ops.Conv2dPrep(wa, 1, pad, 1), ops.Conv2dPrep(wb, 1, pad, 1)
        # the producer of x leaves its mask to the block ("consumer"): emulate it with a Function-free clamp whose gradient is the identity
        class _ReluNoMask(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.clamp_min(0)

            @staticmethod
            def backward(ctx, g):
                return g
        x2 = _ReluNoMask.apply(pre)
        t, skip = ops.Conv2dAuto.apply(x2, pa, wa, ba, None, "consumer", True, True)      # relu left to b; mask_input: x's own mask; skip_out
        y = ops.Conv2dAuto.apply(t, pb, wb, bb, skip, False, True)                          # + skip; chain_in: the mask of t
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        (got,) = torch.autograd.grad(y, pre, go)
        assert float((got - gref).abs().max()) <= 1e-4 * float(gref.abs().max()), k


@pytest.mark.gpu
def test_hip_conv2d_r101_layer_shapes_full_size():
    """the 1x1 layers of the ResNet-101-FPN step at their real size (600x1987 -> 150x497 ... 19x63, both eyes): a sampled set of output
    elements against the oracle's chain (the full oracle would take minutes), the whole tensor within 1e-4 of torch"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(9)
    for cin, cout, h, w in ((256, 64, 150, 497), (64, 256, 150, 497), (512, 128, 75, 249), (1024, 256, 38, 125), (256, 1024, 38, 125), (2048, 512, 19, 63)):
        x = torch.randn((2, cin, h, w), device=dev, generator=gen)
        wt = torch.randn((cout, cin, 1, 1), device=dev, generator=gen) * (1.0 / cin) ** 0.5
        bias = torch.randn((cout,), device=dev, generator=gen)
        y = ops.conv2d(x, ops.Conv2dPrep(wt), bias, relu=True)
        ref = F.relu(F.conv2d(x, wt, bias))
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (cin, cout, h, w)
        # sampled pixels: last pixel of the plane, tile borders, a few in the middle
        ps = sorted({0, 31, 32, 255, 256, h * w - 1, h * w - 2, h * w // 2, h * w // 3 + 7})
        xs = x.reshape(2, cin, h * w)[:, :, ps].cpu().numpy().reshape(2, cin, 1, len(ps))
        want = C.conv2d(xs, wt.cpu().numpy(), bias.cpu().numpy(), relu=True)
        got = y.reshape(2, cout, h * w)[:, :, ps].cpu().numpy().reshape(2, cout, 1, len(ps))
        assert got.tobytes() == want.tobytes(), (cin, cout, h, w)


# (B, Cin, Cout, H, W, dilation): channel counts around the 8-channel stage and the 32 / 64-channel tiles, rows / columns around the
# 8 / 16 x 32 tile, rows shorter than a float4, PSMNet's dilated blocks, the detectors' layer shapes at reduced size
THREE = [(1, 8, 64, 8, 32, 1), (2, 3, 32, 19, 63, 1), (1, 32, 32, 24, 40, 1), (1, 64, 64, 13, 41, 1), (2, 12, 70, 9, 33, 1), (1, 128, 128, 10, 37, 2),
         (1, 5, 7, 3, 2, 1), (1, 16, 130, 17, 65, 2), (1, 256, 18, 6, 10, 1), (1, 9, 9, 1, 1, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", THREE)
def test_hip_conv2d_3x3_bit_exact_vs_oracle_both_tiles(shape):
    from eval_driving_safety_amd import ops
    b, cin, cout, h, w, dil = shape
    x, wt, rs = _case(b, cin, cout, h, w, k=3, seed=sum(shape) + 2)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = rs.randn(b, cout, h, w).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, wt, bias, res, mask))
    assert ops.conv2d_supported(tx, tw, 1, dil, dil)
    prep = ops.Conv2dPrep(tw, 1, dil, dil)
    want_plain = C.conv2d(x, wt, padding=dil, dilation=dil, chunk=8)
    want_full = C.conv2d(x, wt, bias, res, mask, padding=dil, dilation=dil, relu=True, chunk=8)
    for tile in (-1, 0, 1, 2):
        assert ops.conv2d(tx, prep, tile=tile).cpu().numpy().tobytes() == want_plain.tobytes(), ("plain", tile)
        assert ops.conv2d(tx, prep, tb, tr, True, tm, tile=tile).cpu().numpy().tobytes() == want_full.tobytes(), ("bias + residual + relu + mask", tile)
    ref = F.conv2d(tx, tw, tb, 1, dil, dil)
    got = ops.conv2d(tx, prep, tb)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gres = rs.randn(b, cin, h, w).astype(np.float32)
    tg, tgr = torch.tensor(g, device=dev), torch.tensor(gres, device=dev)
    assert ops.conv2d_dgrad(tg, prep).cpu().numpy().tobytes() == C.conv2d(g, wt, padding=dil, dilation=dil, transpose=True, chunk=8).tobytes()
    want_b = C.conv2d(g, wt, residual=gres, mask=x, padding=dil, dilation=dil, transpose=True, chunk=8)
    for tile in (0, 1, 2):
        assert ops.conv2d_dgrad(tg, prep, residual=tgr, mask=tx, tile=tile).cpu().numpy().tobytes() == want_b.tobytes(), tile
    refg = torch.nn.grad.conv2d_input(x.shape, tw, tg, padding=dil, dilation=dil)
    assert float((ops.conv2d_dgrad(tg, prep) - refg).abs().max()) <= 1e-4 * max(1.0, float(refg.abs().max()))


@pytest.mark.gpu
def test_hip_conv2d_3x3_layer_shapes_full_size():
    """3x3 layers of the two detectors at their real size: the whole tensor within 1e-4 of torch, sampled windows bit-exact vs the oracle"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(13)
    for cin, cout, h, w, dil in ((64, 64, 150, 497, 1), (256, 256, 38, 125, 1), (32, 32, 192, 624, 1), (128, 128, 96, 312, 2), (256, 512, 75, 249, 1)):
        x = torch.randn((2, cin, h, w), device=dev, generator=gen)
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=gen) * (1.0 / (9 * cin)) ** 0.5
        bias = torch.randn((cout,), device=dev, generator=gen)
        y = ops.conv2d(x, ops.Conv2dPrep(wt, 1, dil, dil), bias, relu=True)
        ref = F.relu(F.conv2d(x, wt, bias, 1, dil, dil))
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (cin, cout, h, w)
        # a window at the bottom-right corner (image edge, partial tile) and one in the interior: the oracle on the window + halo
        for (r0, c0) in ((h - 6, w - 9), (h // 2, w // 2 - 3)):
            hal = 2 * dil
            rs_, re_, cs_, ce_ = max(0, r0 - hal), min(h, r0 + 6 + hal), max(0, c0 - hal), min(w, c0 + 9 + hal)
            sub = x[:1, :, rs_:re_, cs_:ce_].cpu().numpy()
            want = C.conv2d(sub, wt.cpu().numpy(), bias.cpu().numpy(), padding=dil, dilation=dil, relu=True, chunk=8)
            # rows / columns of the window that do not touch the crop's artificial border (true image borders are real zero padding)
            a0 = 0 if rs_ == 0 else dil
            a1 = want.shape[2] if re_ == h else want.shape[2] - dil
            b0 = 0 if cs_ == 0 else dil
            b1 = want.shape[3] if ce_ == w else want.shape[3] - dil
            got = y[:1, :, rs_ + a0:rs_ + a1, cs_ + b0:cs_ + b1].cpu().numpy()
            assert got.tobytes() == np.ascontiguousarray(want[:, :, a0:a1, b0:b1]).tobytes(), (cin, cout, h, w, r0, c0)


def test_oracle_conv2d_wino_matches_torch():
    """the Winograd restatement (oracle conv2d_wino) is the same operator: within float32 rounding of torch, forward and backward"""
    for shape in ((1, 5, 7, 9, 11), (2, 8, 16, 8, 32), (1, 3, 4, 1, 1), (1, 17, 9, 5, 2), (1, 64, 20, 13, 41)):
        b, cin, cout, h, w = shape
        x, wt, rs = _case(b, cin, cout, h, w, k=3, seed=sum(shape))
        bias, res = rs.randn(cout).astype(np.float32), rs.randn(b, cout, h, w).astype(np.float32)
        got = C.conv2d_wino(x, wt, bias, res, relu=True)
        ref = F.relu(F.conv2d(torch.tensor(x), torch.tensor(wt), torch.tensor(bias), 1, 1) + torch.tensor(res)).numpy()
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        g = rs.randn(b, cout, h, w).astype(np.float32)
        refg = torch.nn.grad.conv2d_input(x.shape, torch.tensor(wt), torch.tensor(g), padding=1).numpy()
        assert np.abs(C.conv2d_wino(g, wt, transpose=True) - refg).max() <= 1e-5 * max(1.0, np.abs(refg).max())
    # the epilogue: + bias, + residual, ReLU, then the mask
    x, wt, rs = _case(1, 8, 5, 4, 6, k=3, seed=3)
    bias, res, mask = rs.randn(5).astype(np.float32), rs.randn(1, 5, 4, 6).astype(np.float32), rs.randn(1, 5, 4, 6).astype(np.float32)
    want = np.maximum(C.conv2d_wino(x, wt) + bias[None, :, None, None] + res, np.float32(0)) * (mask > 0)
    assert C.conv2d_wino(x, wt, bias, res, mask, relu=True).tobytes() == (want + np.float32(0)).astype(np.float32).tobytes()


# (B, Cin, Cout, H, W): channels around the 8-channel stage and the 16 / 64-channel blocks, odd heights and widths (half patches at
# the edge), maps smaller than a patch, more than one tile in every direction
WINO = [(1, 8, 64, 8, 32), (2, 3, 32, 19, 63), (1, 32, 32, 24, 40), (1, 64, 64, 13, 41), (2, 12, 70, 9, 33), (1, 5, 7, 3, 2), (1, 256, 18, 6, 10),
        (1, 9, 9, 1, 1), (1, 17, 130, 17, 65), (3, 16, 16, 2, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", WINO)
def test_hip_conv2d_wino_bit_exact_vs_oracle(shape):
    """csrc/wino2d.hip (Winograd F(2x2,3x3), products on v_mfma_f32_16x16x4_f32) == oracle conv2d_wino bit for bit, forward with the full
    epilogue and backward w.r.t. the input with skip gradient and ReLU mask; and within 1e-4 of torch"""
    from eval_driving_safety_amd import ops
    b, cin, cout, h, w = shape
    x, wt, rs = _case(b, cin, cout, h, w, k=3, seed=sum(shape) + 5)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = rs.randn(b, cout, h, w).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, wt, bias, res, mask))
    prep = ops.Conv2dPrep(tw, 1, 1, 1)
    assert prep.has_wino
    want_plain, want_full = C.conv2d_wino(x, wt), C.conv2d_wino(x, wt, bias, res, mask, relu=True)
    for tile in (-1, 0, 1, 2, 3, 4, 5, 6, 7):  # 8x32 / 16x16 outputs x 64 channels (512 threads), the same x 32 channels (256 threads): one result
        assert ops.conv2d(tx, prep, wino=True, tile=tile).cpu().numpy().tobytes() == want_plain.tobytes(), tile
        assert ops.conv2d(tx, prep, tb, tr, True, tm, wino=True, tile=tile).cpu().numpy().tobytes() == want_full.tobytes(), tile
    ref = F.conv2d(tx, tw, tb, 1, 1)
    assert float((ops.conv2d(tx, prep, tb, wino=True) - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gres = rs.randn(b, cin, h, w).astype(np.float32)
    tg, tgr = torch.tensor(g, device=dev), torch.tensor(gres, device=dev)
    assert ops.conv2d_dgrad(tg, prep, wino=True).cpu().numpy().tobytes() == C.conv2d_wino(g, wt, transpose=True).tobytes()
    want_b = C.conv2d_wino(g, wt, residual=gres, mask=x, transpose=True)
    for tile in (0, 1, 2, 3, 4, 5, 6, 7):
        assert ops.conv2d_dgrad(tg, prep, residual=tgr, mask=tx, wino=True, tile=tile).cpu().numpy().tobytes() == want_b.tobytes(), tile
    refg = torch.nn.grad.conv2d_input(x.shape, tw, tg, padding=1)
    assert float((ops.conv2d_dgrad(tg, prep, wino=True) - refg).abs().max()) <= 1e-4 * max(1.0, float(refg.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("deep", ["0", "1"])
def test_hip_wino_deep_staging_is_the_same_result(deep):
    """the 32-channel workgroups' two staging schedules (csrc/wino2d.hip DEEP: loads travel a stage longer through a second register set,
    uniform stage loop with clamped requests past the end) forced either way through the hooks build: the oracle's bytes, 2D and 3D, from
    two stages per tile up"""
    import os
    from eval_driving_safety_amd import _lib, ops
    dev = torch.device("cuda", 0)
    with _lib.using(_lib.HOOKS_LIB_PATH):
        os.environ["ADV_WINO_DEEP"] = deep
        try:
            for (b, cin, cout, h, w) in ((1, 5, 7, 9, 33), (2, 12, 40, 8, 32), (1, 40, 32, 13, 41), (1, 136, 18, 6, 10)):
                x, wt, rs = _case(b, cin, cout, h, w, k=3, seed=sum((b, cin, cout, h, w)) + 11)
                tx, tw = torch.tensor(x, device=dev), torch.tensor(wt, device=dev)
                prep = ops.Conv2dPrep(tw, 1, 1, 1)
                for tile in (2, 3, 5, 7):
                    assert ops.conv2d(tx, prep, wino=True, tile=tile).cpu().numpy().tobytes() == C.conv2d_wino(x, wt).tobytes(), (cin, tile)
            rs = np.random.RandomState(3)
            for (cin, cout, d, h, w) in ((4, 6, 3, 5, 9), (12, 32, 1, 8, 33), (32, 32, 4, 9, 20)):
                x = rs.randn(1, cin, d, h, w).astype(np.float32)
                wt = (rs.randn(cout, cin, 3, 3, 3) * 0.1).astype(np.float32)
                prep = ops.Conv3dWinoPrep(torch.tensor(wt, device=dev))
                want = C.conv3d_wino(x, wt)
                for tile in (2, 3, 5, 7):
                    assert ops.conv3d_wino(torch.tensor(x, device=dev), prep, None, relu=False, tile=tile).cpu().numpy().tobytes() == want.tobytes(), (cin, tile)
        finally:
            del os.environ["ADV_WINO_DEEP"]


@pytest.mark.gpu
def test_hip_conv2d_wino_layer_shapes_full_size():
    """the detectors' 3x3 layers at their real size through the Winograd kernel: the whole tensor within 1e-4 of torch, windows (on the
    even patch grid) bit-exact vs the oracle"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(17)
    for cin, cout, h, w in ((64, 64, 150, 497), (256, 256, 38, 125), (32, 32, 192, 624), (256, 512, 75, 249)):
        x = torch.randn((2, cin, h, w), device=dev, generator=gen)
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=gen) * (1.0 / (9 * cin)) ** 0.5
        bias = torch.randn((cout,), device=dev, generator=gen)
        y = ops.conv2d(x, ops.Conv2dPrep(wt, 1, 1, 1), bias, relu=True, wino=True)
        ref = F.relu(F.conv2d(x, wt, bias, 1, 1))
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (cin, cout, h, w)
        for (r0, c0) in (((h - 6) & ~1, (w - 9) & ~1), ((h // 2) & ~1, (w // 2 - 3) & ~1)):
            rs_, re_, cs_, ce_ = max(0, r0 - 2), min(h, r0 + 6 + 2), max(0, c0 - 2), min(w, c0 + 10 + 2)     # even origins: the same patch grid
            sub = x[:1, :, rs_:re_, cs_:ce_].cpu().numpy()
            want = C.conv2d_wino(sub, wt.cpu().numpy(), bias.cpu().numpy(), relu=True)
            a0 = 0 if rs_ == 0 else 2
            a1 = want.shape[2] if re_ == h else (want.shape[2] - 2) & ~1
            b0 = 0 if cs_ == 0 else 2
            b1 = want.shape[3] if ce_ == w else (want.shape[3] - 2) & ~1
            got = y[:1, :, rs_ + a0:rs_ + a1, cs_ + b0:cs_ + b1].cpu().numpy()
            assert got.tobytes() == np.ascontiguousarray(want[:, :, a0:a1, b0:b1]).tobytes(), (cin, cout, h, w, r0, c0)


@pytest.mark.gpu
def test_hip_bias_act_equals_the_three_torch_kernels():
    """adv_bias_act_f32: y <- relu(y + bias + residual) in one pass, bit-equal to torch's add, add, relu done apart"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(2)
    for shape in ((2, 5, 7, 9), (1, 64, 150, 497), (3, 7, 1, 1), (1, 3, 4, 5, 6)):
        y = torch.randn(shape, device=dev, generator=gen)
        bias = torch.randn((shape[1],), device=dev, generator=gen)
        res = torch.randn(shape, device=dev, generator=gen)
        bshape = (1, -1) + (1,) * (len(shape) - 2)
        assert torch.equal(ops.bias_act_(y.clone(), bias, res, True), torch.relu(y + bias.view(bshape) + res))
        assert torch.equal(ops.bias_act_(y.clone(), bias, None, False), y + bias.view(bshape))
        assert torch.equal(ops.bias_act_(y.clone(), None, res, True), torch.relu(y + res))
        assert torch.equal(ops.bias_act_(y.clone(), None, None, True), torch.relu(y))
