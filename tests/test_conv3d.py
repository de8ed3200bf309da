"""3x3x3 convolution on the float32 matrix cores: bit-exact against the C oracle's k-ordered fmaf chain, within
float32 rounding of torch's conv3d (the floating-point reference), forward and adjoint."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

C = pytest.importorskip("oracle.oracle_c", reason="make -C oracle first (build() does it)")


def _case(b, cin, cout, d, h, w, seed=0):
    rs = np.random.RandomState(seed)
    x = rs.randn(b, cin, d, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3, 3) * 0.1).astype(np.float32)
    return x, wt


def test_oracle_conv_matches_torch_and_is_self_adjoint():
    x, wt = _case(1, 8, 5, 3, 6, 7)
    y = C.conv3d_k3(x, wt)
    ref = F.conv3d(torch.tensor(x), torch.tensor(wt), padding=1).numpy()
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)
    g = np.random.RandomState(1).randn(*y.shape).astype(np.float32)
    gx = C.conv3d_k3(g, wt, transpose=True)
    tx = torch.tensor(x, requires_grad=True)
    F.conv3d(tx, torch.tensor(wt), padding=1).backward(torch.tensor(g))
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert np.array_equal(C.conv3d_k3(x, wt, relu=True), np.maximum(y, 0))


SHAPES = [(1, 8, 5, 3, 6, 7), (2, 16, 32, 2, 8, 32), (1, 8, 33, 3, 9, 40), (1, 24, 64, 5, 10, 45), (1, 8, 1, 2, 3, 4),
          (1, 4, 12, 3, 17, 36), (1, 12, 8, 4, 8, 64), (1, 8, 16, 5, 9, 78), (2, 4, 8, 3, 5, 33), (1, 4, 4, 2, 2, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_hip_conv3d_bit_exact_vs_oracle(shape):
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape))
    dev = torch.device("cuda", 0)
    tx, tw = torch.tensor(x, device=dev), torch.tensor(wt, device=dev)
    wp = ops.conv3d_k3_prep(tw)
    y = ops.conv3d_k3(tx, wp, cout)
    want = C.conv3d_k3(x, wt)
    assert y.cpu().numpy().tobytes() == want.tobytes(), "forward"
    assert ops.conv3d_k3(tx, wp, cout, relu=True).cpu().numpy().tobytes() == np.maximum(want, 0).tobytes(), "relu"
    if cout % 4 == 0 or cout < 4:                       # the adjoint needs Cin' = cout to be a multiple of 4, or 1..3
        g = np.random.RandomState(3).randn(*want.shape).astype(np.float32)
        wpt = ops.conv3d_k3_prep(tw, transpose=True)
        gx = ops.conv3d_k3(torch.tensor(g, device=dev), wpt, cin)
        assert gx.cpu().numpy().tobytes() == C.conv3d_k3(g, wt, transpose=True).tobytes(), "adjoint"


@pytest.mark.gpu
def test_hip_conv3d_autograd_vs_torch_at_cost_volume_scale():
    """a slab of the DSGN-sized volume: 64 -> 32 channels, 8 x 96 x 312 voxels, against torch's conv3d on the GPU"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((1, 64, 8, 96, 312), device=dev, generator=gen)
    wt = torch.randn((32, 64, 3, 3, 3), device=dev, generator=gen) * 0.05
    xr = x.clone().requires_grad_(True)
    ref = F.conv3d(xr, wt, padding=1)
    g = torch.randn(ref.shape, device=dev, generator=gen)
    ref.backward(g)
    xm = x.clone().requires_grad_(True)
    y = ops.Conv3dK3.apply(xm, ops.conv3d_k3_prep(wt), ops.conv3d_k3_prep(wt, transpose=True), 32)
    y.backward(g)
    scale = float(ref.abs().max())
    assert float((y - ref).abs().max()) <= 1e-4 * scale
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())


@pytest.mark.gpu
def test_hip_conv3d_single_output_channel_with_torch_adjoint():
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((1, 32, 6, 24, 40), device=dev, generator=gen)
    wt = torch.randn((1, 32, 3, 3, 3), device=dev, generator=gen) * 0.05
    xr = x.clone().requires_grad_(True)
    ref = F.conv3d(xr, wt, padding=1)
    g = torch.randn(ref.shape, device=dev, generator=gen)
    ref.backward(g)
    xm = x.clone().requires_grad_(True)
    y = ops.Conv3dK3.apply(xm, ops.conv3d_k3_prep(wt), None, 1, wt)
    y.backward(g)
    assert y.detach().cpu().numpy().tobytes() == C.conv3d_k3(x.cpu().numpy(), wt.cpu().numpy()).tobytes()
    assert float((y - ref).abs().max()) <= 1e-4 * float(ref.detach().abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
def test_hip_persistent_tile_walk_equals_one_workgroup_per_tile(masked, route):
    """more tiles than resident workgroups: every workgroup walks several tiles with the stage pipeline running across them
    (XCD-contiguous shares) - the same bits as one workgroup per tile, and as the oracle on a slab"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn((2, 8, 27, 60, 300), device=dev, generator=gen)          # 10 x 8 x 14 tiles x 2 batch x 2 blocks = 4480 workgroup-tiles
    wt = torch.randn((40, 8, 3, 3, 3), device=dev, generator=gen) * 0.1
    bias = torch.linspace(-0.2, 0.3, 40, device=dev)
    wp = ops.conv3d_k3_prep(wt)
    mask = 0b101010101010101010101010101 if masked else ops.ALL_TAPS

    def run():
        return ops._conv3d_ex(x, wp, 40, 1, True, bias, mask)

    y = run()
    with route(ADV_CONV_ONE_TILE_PER_WG="1"):
        assert torch.equal(run(), y)
    want = C.conv3d_k3_ex(x[1:, :, 20:27].cpu().numpy(), wt.cpu().numpy(), bias=bias.cpu().numpy(), relu=True, tap_mask=mask)
    got = ops._conv3d_ex(x[1:, :, 20:27].contiguous(), wp, 40, 1, True, bias, mask)
    assert got.cpu().numpy().tobytes() == want.tobytes()
    # the interior depth planes of the slab do not see its artificial borders: they must equal the full run
    assert torch.equal(got[:, :, 1:6], y[1:, :, 21:26])


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["ADV_CONV_NO_DMA=1", "ADV_CONV_TH=4", "ADV_CONV_TH=8", "ADV_CONV_TH=44", "ADV_CONV_GENERIC=1", "ADV_CONV_CLASS_LAUNCHES=1", "ADV_CONV_T_CLASS_TILES=1"])
def test_hip_alternative_code_paths_give_the_same_bits(switch, route):
    """register-staged vs LDS-DMA stages, both tile heights, the scalar-staging kernel, for the transposed convolution eight
    launches / the class as a tile index / all classes per tile: identical results on plain, masked, strided and transposed layers (W % 4 == 0 and != 0)"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(21)
    results = []

    def layers():
        out = []
        for w in (40, 38):
            g = torch.Generator(device=dev).manual_seed(100 + w)
            x = torch.randn((2, 8, 6, 20, w), device=dev, generator=g)
            wt = torch.randn((40, 8, 3, 3, 3), device=dev, generator=g) * 0.1
            bias = torch.linspace(-0.1, 0.2, 40, device=dev)
            out.append(ops.conv3d_k3(x, ops.conv3d_k3_prep(wt), 40, relu=True, bias=bias))
            out.append(ops._conv3d_ex(x, ops.conv3d_k3_prep(wt), 40, 1, False, None, 0b000111000101010000111000101))
            out.append(ops.conv3d_k3_s2(x, ops.conv3d_k3_s2_prep(wt), 40, relu=True, bias=bias, route="s2d"))
            wtt = torch.randn((8, 24, 3, 3, 3), device=dev, generator=g) * 0.1
            out.append(ops.conv_transpose3d_k3_s2(x, ops.conv_transpose3d_k3_s2_prep(wtt), 24, bias=bias[:24]))
        return out

    results.append(layers())                     # the shipped library
    with route(**dict([switch.split("=")])):     # the hooks build with the switch set
        results.append(layers())
    for a, b in zip(*results):
        assert torch.equal(a, b)


NARROW = [(1, 32, 1, 6, 24, 40), (2, 8, 1, 3, 9, 33), (1, 12, 3, 4, 8, 78), (1, 8, 8, 5, 10, 45), (1, 16, 5, 2, 17, 36), (1, 4, 2, 1, 1, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", NARROW)
def test_hip_narrow_layers_bit_exact_vs_oracle_and_vs_padded_matrix_kernel(shape, route):
    """Cout <= 8 runs on the vector ALUs (narrow_out), its adjoint with 1..3 input channels too (narrow_in): the same bits as the
    oracle's fmaf chain AND as the matrix kernel that pads the channels to 32 rows (ADV_CONV_NO_NARROW=1)."""
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape) + 7)
    dev = torch.device("cuda", 0)
    tx, tw = torch.tensor(x, device=dev), torch.tensor(wt, device=dev)
    wp = ops.conv3d_k3_prep(tw)
    bias = torch.linspace(-0.3, 0.4, cout, device=dev)
    y = ops.conv3d_k3(tx, wp, cout)
    yb = ops.conv3d_k3(tx, wp, cout, relu=True, bias=bias)
    want = C.conv3d_k3(x, wt)
    assert y.cpu().numpy().tobytes() == want.tobytes(), "narrow forward vs oracle"
    assert yb.cpu().numpy().tobytes() == C.conv3d_k3_ex(x, wt, bias=bias.cpu().numpy(), relu=True).tobytes(), "bias + relu"
    with route(ADV_CONV_NO_NARROW="1"):
        assert torch.equal(ops.conv3d_k3(tx, wp, cout), y), "narrow forward vs the padded matrix kernel"
    if cout < 4:
        g = np.random.RandomState(5).randn(*want.shape).astype(np.float32)
        wpt = ops.conv3d_k3_prep(tw, transpose=True)
        gx = ops.conv3d_k3(torch.tensor(g, device=dev), wpt, cin)
        assert gx.cpu().numpy().tobytes() == C.conv3d_k3(g, wt, transpose=True).tobytes(), "narrow adjoint vs oracle"


@pytest.mark.gpu
def test_hip_single_output_layer_autograd_on_the_narrow_kernels():
    """32 -> 1 (the per-plane score layer of a cost-volume network) forward AND backward on libadvengine, against torch"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((2, 32, 6, 24, 78), device=dev, generator=gen)
    wt = torch.randn((1, 32, 3, 3, 3), device=dev, generator=gen) * 0.05
    xr = x.clone().requires_grad_(True)
    ref = F.conv3d(xr, wt, padding=1)
    g = torch.randn(ref.shape, device=dev, generator=gen)
    ref.backward(g)
    xm = x.clone().requires_grad_(True)
    y = ops.Conv3dK3.apply(xm, ops.conv3d_k3_prep(wt), ops.conv3d_k3_prep(wt, transpose=True), 1)
    y.backward(g)
    assert float((y - ref).abs().max()) <= 1e-4 * float(ref.detach().abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    assert xm.grad.cpu().numpy().tobytes() == C.conv3d_k3(g.cpu().numpy(), wt.cpu().numpy(), transpose=True).tobytes()


# ------------------------------------------------------------------------------------------ hourglass layers
def test_oracle_strided_and_transposed_conv_match_torch():
    """the oracle's stride-2 convolution, the transposed convolution as eight masked parity classes, bias and ReLU,
    against torch's conv3d / conv_transpose3d (CPU)"""
    rs = np.random.RandomState(5)
    x = rs.randn(2, 8, 5, 6, 7).astype(np.float32)                     # odd and even dims
    wt = (rs.randn(12, 8, 3, 3, 3) * 0.1).astype(np.float32)
    bias = rs.randn(12).astype(np.float32)
    y = C.conv3d_k3_ex(x, wt, bias=bias, stride=2)
    ref = F.conv3d(torch.tensor(x), torch.tensor(wt), torch.tensor(bias), stride=2, padding=1).numpy()
    assert y.shape == ref.shape == (2, 12, 3, 3, 4)
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(C.conv3d_k3_ex(x, wt, bias=bias, stride=2, relu=True), np.maximum(y, 0))
    assert np.array_equal(C.conv3d_k3_ex(x, wt), C.conv3d_k3(x, wt))   # the plain convolution is the default
    np.testing.assert_allclose(C.conv3d_k3_s2(x, wt, bias=bias), ref, rtol=1e-5, atol=1e-5)   # space-to-depth formulation
    w_t = (rs.randn(8, 5, 3, 3, 3) * 0.1).astype(np.float32)           # ConvTranspose3d layout [Cin, Cout, 3,3,3]
    up = C.conv_transpose3d_k3_s2(x, w_t)
    ref = F.conv_transpose3d(torch.tensor(x), torch.tensor(w_t), stride=2, padding=1, output_padding=1).numpy()
    assert up.shape == ref.shape == (2, 5, 10, 12, 14)
    np.testing.assert_allclose(up, ref, rtol=1e-5, atol=1e-5)
    # and it IS the adjoint of the strided convolution with the same weights
    xs = rs.randn(1, 5, 10, 12, 14).astype(np.float32)
    down = C.conv3d_k3_ex(xs, w_t, stride=2)                           # w_t read as [out = 8, in = 5]
    lhs = float((down.astype(np.float64) * x[:1]).sum())
    rhs = float((xs.astype(np.float64) * C.conv_transpose3d_k3_s2(x[:1], w_t)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


HG_SHAPES = [(1, 8, 12, 5, 6, 7), (2, 16, 32, 4, 8, 32), (1, 8, 33, 3, 9, 40), (1, 32, 64, 6, 16, 44), (1, 16, 32, 7, 5, 36), (1, 8, 40, 6, 10, 40)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", HG_SHAPES)
def test_hip_hourglass_layers_bit_exact_vs_oracle(shape):
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape))
    rs = np.random.RandomState(9)
    bias = rs.randn(cout).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb = torch.tensor(x, device=dev), torch.tensor(wt, device=dev), torch.tensor(bias, device=dev)
    wp = ops.conv3d_k3_prep(tw)
    # bias + ReLU epilogue of the plain convolution
    got = ops.conv3d_k3(tx, wp, cout, relu=True, bias=tb)
    assert got.cpu().numpy().tobytes() == C.conv3d_k3_ex(x, wt, bias=bias, relu=True).tobytes(), "bias + relu"
    # stride 2: the direct strided kernel, and space-to-depth + the stride-1 kernel with per-class tap masks
    got = ops.conv3d_k3_s2(tx, wp, cout, bias=tb)
    want = C.conv3d_k3_ex(x, wt, bias=bias, stride=2, chunk=ops.conv3d_k3_s2_stage_channels(tx, cout))
    assert got.cpu().numpy().tobytes() == want.tobytes(), "stride 2 (direct)"
    ref = F.conv3d(tx, tw, tb, stride=2, padding=1)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert ops.space_to_depth2(tx).cpu().numpy().tobytes() == C.space_to_depth2(x).tobytes(), "space to depth"
    s2 = ops.conv3d_k3_s2_prep(tw)
    assert sorted(bin(m).count("1") for m in s2[1]) == [1, 2, 2, 2, 4, 4, 4, 8]
    got = ops.conv3d_k3_s2(tx, s2, cout, bias=tb, relu=True, route="s2d")
    assert got.cpu().numpy().tobytes() == C.conv3d_k3_s2(x, wt, bias=bias, relu=True).tobytes(), "stride 2 (space-to-depth)"
    auto = ops.conv3d_k3_s2(tx, s2, cout, bias=tb, relu=True)           # direct where the strided matrix kernel takes the shape
    assert torch.equal(auto, ops.conv3d_k3_s2(tx, wp, cout, bias=tb, relu=True) if ops.conv3d_k3_s2_stage_channels(tx, cout) == 2 else got)
    assert float((got - F.relu(ref)).abs().max()) <= 1e-4 * float(ref.abs().max())
    # transposed convolution: eight masked-tap launches of the stride-1 kernel
    w_t = (rs.randn(cin, cout, 3, 3, 3) * 0.1).astype(np.float32)
    classes = ops.conv_transpose3d_k3_s2_prep(torch.tensor(w_t, device=dev))
    assert sorted(bin(m).count("1") for _, m, _ in classes) == [1, 2, 2, 2, 4, 4, 4, 8]     # 27 taps in all
    got = ops.conv_transpose3d_k3_s2(tx, classes, cout, bias=tb, relu=True)
    want = C.conv_transpose3d_k3_s2(x, w_t, bias=bias, relu=True)
    assert got.cpu().numpy().tobytes() == want.tobytes(), "transposed"
    ref = F.relu(F.conv_transpose3d(tx, torch.tensor(w_t, device=dev), tb, stride=2, padding=1, output_padding=1))
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_hip_hourglass_autograd_vs_torch():
    """down (stride 2) and up (transposed) layers as autograd functions: gradients w.r.t. the input within 1e-4 of torch's"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((1, 32, 12, 24, 80), device=dev, generator=gen)
    wd = torch.randn((64, 32, 3, 3, 3), device=dev, generator=gen) * 0.05           # down: 32 -> 64
    wu = torch.randn((64, 32, 3, 3, 3), device=dev, generator=gen) * 0.05           # up (ConvTranspose layout [in = 64, out = 32])
    xr = x.clone().requires_grad_(True)
    ref = F.conv_transpose3d(F.relu(F.conv3d(xr, wd, stride=2, padding=1)), wu, stride=2, padding=1, output_padding=1)
    g = torch.randn(ref.shape, device=dev, generator=gen)
    ref.backward(g)
    xm = x.clone().requires_grad_(True)
    down = ops.Conv3dK3S2.apply(xm, ops.conv3d_k3_s2_prep(wd), ops.conv_transpose3d_k3_s2_prep(wd), 64)
    up = ops.ConvTranspose3dK3S2.apply(F.relu(down), ops.conv_transpose3d_k3_s2_prep(wu), ops.conv3d_k3_s2_prep(wu), 32)
    up.backward(g)
    assert tuple(up.shape) == tuple(ref.shape) == (1, 32, 12, 24, 80)
    assert float((up - ref).abs().max()) <= 1e-4 * float(ref.detach().abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("w", [40, 38])
def test_hip_hourglass_skip_and_mask_in_the_down_layers_backward(w):
    """<round 4> an hourglass whose down-sampling layer's input also feeds the up-sampling layer's skip connection, with the producer's ReLU
    left to the down-sampling layer's backward (ops.Conv3dK3S2 skip_out / mask_input): the fused backward launch
    (adv_conv_transpose3d_k3_s2_dgrad_f32: transposed convolution + skip gradient + mask) equals the oracle's transposed convolution, numpy's
    add and where bit for bit; the graph's values and input gradient equal torch's (1e-4).  w = 38: the gradient volume is 19 wide - the
    register-staged variant of the all-classes kernel (any width), same contract."""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn((1, 8, 6, 10, w), device=dev, generator=gen)
    w0 = torch.randn((16, 8, 3, 3, 3), device=dev, generator=gen) * 0.1              # producer: stride 1, ReLU
    wd = torch.randn((32, 16, 3, 3, 3), device=dev, generator=gen) * 0.05            # down: 16 -> 32
    wu = torch.randn((32, 16, 3, 3, 3), device=dev, generator=gen) * 0.05            # up (ConvTranspose layout [in = 32, out = 16]) + skip
    xr = x.clone().requires_grad_(True)
    a = F.relu(F.conv3d(xr, w0, padding=1))
    ref = F.relu(F.conv_transpose3d(F.relu(F.conv3d(a, wd, stride=2, padding=1)), wu, stride=2, padding=1, output_padding=1) + a)
    g = torch.randn(ref.shape, device=dev, generator=gen)
    ref.backward(g)
    xm = x.clone().requires_grad_(True)
    am = ops.Conv3dK3.apply(xm, ops.conv3d_k3_prep(w0), ops.conv3d_k3_prep(w0, transpose=True), 16, None, None, "consumer")
    down, skip = ops.Conv3dK3S2.apply(am, ops.conv3d_k3_s2_prep(wd), ops.conv_transpose3d_k3_s2_prep(wd), 32, None, True, True, True)
    up = ops.ConvTranspose3dK3S2.apply(down, ops.conv_transpose3d_k3_s2_prep(wu), ops.conv3d_k3_s2_prep(wu), 16, None, True, skip)
    up.backward(g)
    assert float((up - ref).abs().max()) <= 1e-4 * float(ref.detach().abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    # the fused launch alone against the oracle's pieces
    gy = torch.randn(down.shape, device=dev, generator=gen)
    gs = torch.randn(am.shape, device=dev, generator=gen)
    classes = ops.conv_transpose3d_k3_s2_prep(wd)
    fused = ops.conv_transpose3d_k3_s2_dgrad(gy, classes, 16, residual=gs, mask=am.detach())
    wt_t = wd.cpu().numpy()                                                           # conv weight [out = 32, in = 16] read as ConvTranspose [in = 32, out = 16]
    want = C.conv_transpose3d_k3_s2(gy.cpu().numpy(), wt_t) + gs.cpu().numpy()
    want = np.where(am.detach().cpu().numpy() > 0, want, np.float32(0)).astype(np.float32)
    assert fused is not None and fused.cpu().numpy().tobytes() == want.tobytes()
    assert torch.equal(ops.conv_transpose3d_k3_s2_dgrad(gy, classes, 16, mask=am.detach()),
                       ops.relu_backward(ops.conv_transpose3d_k3_s2(gy, classes, 16), am.detach()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", HG_SHAPES + [(1, 32, 1, 3, 8, 36), (1, 2, 32, 3, 8, 36)])
def test_hip_residual_epilogue_equals_conv_then_add_then_relu(shape, route):
    """y = relu(conv(x) + bias + skip) in the epilogue (an hourglass's skip connection): the same float operations in the same
    order as the convolution followed by a separate add and max - on the matrix kernel (all-taps, masked / transposed, every
    staging variant) and on the narrow vector-ALU kernels - and equal to the oracle's convolution plus a numpy add."""
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape) + 1)
    rs = np.random.RandomState(17)
    bias = rs.randn(cout).astype(np.float32)
    skip = rs.randn(b, cout, d, h, w).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb, ts = (torch.tensor(a, device=dev) for a in (x, wt, bias, skip))
    wp = ops.conv3d_k3_prep(tw)
    want = np.maximum(C.conv3d_k3_ex(x, wt, bias=bias) + skip, np.float32(0))
    import contextlib
    for env in ({}, {"ADV_CONV_NO_DMA": "1"}, {"ADV_CONV_GENERIC": "1"}, {"ADV_CONV_NO_NARROW": "1"}):
        with (route(**env) if env else contextlib.nullcontext()):        # {}: the shipped library
            got = ops.conv3d_k3(tx, wp, cout, relu=True, bias=tb, residual=ts)
            assert got.cpu().numpy().tobytes() == want.tobytes(), env
            assert torch.equal(got, F.relu(ops.conv3d_k3(tx, wp, cout, bias=tb) + ts)), env
            assert torch.equal(ops.conv3d_k3(tx, wp, cout, residual=ts), ops.conv3d_k3(tx, wp, cout) + ts), env      # no bias, no relu
    if cin % 4 == 0:
        w_t = (rs.randn(cin, cout, 3, 3, 3) * 0.1).astype(np.float32)
        classes = ops.conv_transpose3d_k3_s2_prep(torch.tensor(w_t, device=dev))
        skip2 = torch.tensor(rs.randn(b, cout, 2 * d, 2 * h, 2 * w).astype(np.float32), device=dev)
        # (<round 4> ADV_CONV_T_TD: the all-classes kernel's tile as 1 x 4, 2 x 2 or 4 x 1 input planes x rows - the host picks by padding)
        # (ADV_CONV_T_NO_DMA: the register-staged variant - what widths that are not multiples of 4 take - on every shape)
        for env in ({}, {"ADV_CONV_T_CLASS_TILES": "1"}, {"ADV_CONV_CLASS_LAUNCHES": "1"}, {"ADV_CONV_T_TD": "1"}, {"ADV_CONV_T_TD": "2"}, {"ADV_CONV_T_TD": "4"},
                    {"ADV_CONV_T_NO_DMA": "1"}, {"ADV_CONV_T_NO_DMA": "1", "ADV_CONV_T_TD": "2"}, {"ADV_CONV_T_NO_DMA": "1", "ADV_CONV_T_TD": "4"}):
            with (route(**env) if env else contextlib.nullcontext()):
                got = ops.conv_transpose3d_k3_s2(tx, classes, cout, bias=tb, relu=True, residual=skip2)
                want_t = np.maximum(C.conv_transpose3d_k3_s2(x, w_t, bias=bias) + skip2.cpu().numpy(), np.float32(0))
                assert got.cpu().numpy().tobytes() == want_t.tobytes(), ("transposed", env)
    with pytest.raises(ValueError):
        ops.conv3d_k3(tx, wp, cout, residual=ts[:, :, :-1].contiguous())


@pytest.mark.gpu
def test_hip_skip_connection_autograd_vs_torch():
    """relu(up(x) + bias + skip) as ONE autograd function: the value equals the separate add / relu bit for bit, and both x and the
    skip tensor receive torch's gradients (the skip's gradient is the masked incoming gradient itself)"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((1, 64, 6, 12, 40), device=dev, generator=gen)
    skip = torch.randn((1, 32, 12, 24, 80), device=dev, generator=gen)
    wu = torch.randn((64, 32, 3, 3, 3), device=dev, generator=gen) * 0.05
    bu = torch.randn((32,), device=dev, generator=gen)
    wc = torch.randn((32, 32, 3, 3, 3), device=dev, generator=gen) * 0.05
    g = torch.randn(skip.shape, device=dev, generator=gen)
    xr, sr = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    ref = F.relu(F.conv_transpose3d(xr, wu, bu, stride=2, padding=1, output_padding=1) + sr)
    ref2 = F.relu(F.conv3d(ref, wc, padding=1) + sr)
    ref2.backward(g)
    xm, sm = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    cls, fwd = ops.conv_transpose3d_k3_s2_prep(wu), ops.conv3d_k3_s2_prep(wu)
    up = ops.ConvTranspose3dK3S2.apply(xm, cls, fwd, 32, bu, True, sm)
    assert torch.equal(up, F.relu(ops.conv_transpose3d_k3_s2(x, cls, 32, bias=bu) + skip))
    out = ops.Conv3dK3.apply(up, ops.conv3d_k3_prep(wc), ops.conv3d_k3_prep(wc, transpose=True), 32, None, None, True, sm)  # weight, bias, relu, residual
    out.backward(g)
    assert float((out - ref2).abs().max()) <= 1e-4 * float(ref2.detach().abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    assert float((sm.grad - sr.grad).abs().max()) <= 1e-4 * float(sr.grad.abs().max())


@pytest.mark.gpu
def test_hip_conv3d_batch_beyond_2_31_elements_equals_per_item_launches():
    """maximum sizes: a batch whose input has 2.2e9 elements (8.8 GB; every item fits 32-bit offsets, the batch does not) in one
    launch equals the items convolved one by one - forward with bias + ReLU, and the masked / transposed kernel."""
    from eval_driving_safety_amd import ops
    if torch.cuda.mem_get_info(0)[0] < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free device memory")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(12)
    b, cin, cout, d, h, w = 6, 64, 32, 48, 96, 1248
    assert b * cin * d * h * w > 2 ** 31
    x = torch.empty((b, cin, d, h, w), device=dev)
    for i in range(b):
        x[i].normal_(generator=gen)
    wt = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=gen) * 0.05
    bias = torch.randn((cout,), device=dev, generator=gen)
    wp = ops.conv3d_k3_prep(wt)
    y = ops.conv3d_k3(x, wp, cout, relu=True, bias=bias)
    for i in range(b):
        assert torch.equal(y[i:i + 1], ops.conv3d_k3(x[i:i + 1], wp, cout, relu=True, bias=bias)), i
    del y
    xs = x[:, :, :24, :48, :624].contiguous()                       # transposed: output [6,32,48,96,1248] = 1.1e9, eight classes
    w_t = torch.randn((cin, cout, 3, 3, 3), device=dev, generator=gen) * 0.05
    classes = ops.conv_transpose3d_k3_s2_prep(w_t)
    up = ops.conv_transpose3d_k3_s2(xs, classes, cout, bias=bias)
    for i in (0, b - 1):
        assert torch.equal(up[i:i + 1], ops.conv_transpose3d_k3_s2(xs[i:i + 1], classes, cout, bias=bias)), i
    del up, xs
    wd = torch.randn((64, cin, 3, 3, 3), device=dev, generator=gen) * 0.05            # strided, direct matrix kernel: 64 -> 64
    wpd = ops.conv3d_k3_prep(wd)
    assert ops.conv3d_k3_s2_stage_channels(x, 64) == 2
    down = ops.conv3d_k3_s2(x, wpd, 64, relu=True)
    for i in (0, b - 1):
        assert torch.equal(down[i:i + 1], ops.conv3d_k3_s2(x[i:i + 1], wpd, 64, relu=True)), i


S2_DIRECT = [(1, 8, 33, 3, 9, 40), (2, 8, 32, 5, 19, 36), (1, 16, 5, 4, 8, 72), (1, 4, 24, 3, 3, 4), (2, 12, 64, 5, 7, 36), (1, 32, 64, 6, 16, 44), (1, 4, 96, 4, 10, 132), (1, 12, 40, 1, 1, 4), (1, 16, 72, 7, 33, 64),
             (1, 8, 64, 13, 10, 40), (1, 8, 40, 9, 20, 36)]     # flat volumes: 5 and 10 output rows (the 4 x 1 and 2 x 2 tile shapes by themselves)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", S2_DIRECT)
def test_hip_direct_strided_kernel_bit_exact_vs_oracle(shape, route):
    """the direct strided matrix kernel (W % 4 == 0; stages of TWO input channels, operands read from LDS at stride 2; more than 32
    output channels: two channel blocks per workgroup, else two rows per wave) against the oracle run with the same stage size - odd
    and even dims, 1 to 3 blocks of output channels, bias / skip connection / ReLU - and within 1e-4 of torch; ADV_CONV_S2_GENERIC=1 sends the same call to the
    scalar-staging kernel, whose stage is 4 channels"""
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape) + 3)
    rs = np.random.RandomState(23)
    bias = rs.randn(cout).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb = torch.tensor(x, device=dev), torch.tensor(wt, device=dev), torch.tensor(bias, device=dev)
    wp = ops.conv3d_k3_prep(tw)
    assert ops.conv3d_k3_s2_stage_channels(tx, cout) == 2
    assert ops.conv3d_k3_s2_stage_channels(tx, 32) == 2 and ops.conv3d_k3_s2_stage_channels(tx[..., :w - 1].contiguous(), cout) == 4
    got = ops.conv3d_k3_s2(tx, wp, cout, bias=tb, relu=True)
    assert got.cpu().numpy().tobytes() == C.conv3d_k3_ex(x, wt, bias=bias, stride=2, relu=True, chunk=2).tobytes(), "direct strided, bias + relu"
    ref = F.relu(F.conv3d(tx, tw, tb, stride=2, padding=1))
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    plain = ops.conv3d_k3_s2(tx, wp, cout)
    assert plain.cpu().numpy().tobytes() == C.conv3d_k3_ex(x, wt, stride=2, chunk=2).tobytes(), "direct strided, plain"
    skip = torch.tensor(rs.randn(*plain.shape).astype(np.float32), device=dev)
    assert torch.equal(ops._conv3d_ex(tx, wp, cout, 2, True, tb, residual=skip), F.relu(ops.conv3d_k3_s2(tx, wp, cout, bias=tb) + skip))
    if cout > 32:     # one or two output depth planes per workgroup (the host picks by the tile count): the same bits
        for pd in ("1", "2"):
            with route(ADV_CONV_S2_PD=pd):
                assert torch.equal(ops.conv3d_k3_s2(tx, wp, cout, bias=tb, relu=True), got), "planes per workgroup: " + pd
        # <round 4> the tile's four waves over 1, 2 or 4 output planes (1 x 4, 2 x 2, 4 x 1 planes x rows; the host picks the shape
        # that pads the volume least), alone and with two planes per wave: the same bits
        for wd in ("1", "2", "4"):
            for pd in ("1", "2"):
                with route(ADV_CONV_S2_WD=wd, ADV_CONV_S2_PD=pd):
                    assert torch.equal(ops.conv3d_k3_s2(tx, wp, cout, bias=tb, relu=True), got), "tile %s planes x %d rows, %s planes per wave" % (wd, 4 // int(wd), pd)
                    assert torch.equal(ops._conv3d_ex(tx, wp, cout, 2, True, tb, residual=skip), F.relu(ops.conv3d_k3_s2(tx, wp, cout, bias=tb) + skip))
    with route(ADV_CONV_S2_GENERIC="1"):
        assert ops.conv3d_k3_s2_stage_channels(tx, cout) == 4
        slow = ops.conv3d_k3_s2(tx, wp, cout, bias=tb, relu=True)
        assert slow.cpu().numpy().tobytes() == C.conv3d_k3_ex(x, wt, bias=bias, stride=2, relu=True, chunk=4).tobytes(), "scalar-staging strided kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 8, 40, 6, 20, 40), (2, 32, 32, 5, 12, 36), (1, 12, 36, 3, 9, 38), (1, 64, 32, 4, 8, 76),
                                   (1, 32, 1, 4, 12, 40), (2, 16, 3, 3, 9, 38)])       # (the last two: the adjoint of a 32 -> 1 / 16 -> 3 layer - the narrow-input kernel)
def test_hip_masked_dgrad_equals_conv_then_relu_backward(shape):
    """adv_conv3d_k3_masked_f32: the backward w.r.t. the input with the ReLU mask of that input in the epilogue - bit-equal to the
    unmasked kernel followed by relu_backward, on every staging route; and ops.Conv3dK3's chain flags give the same gradient as the
    unchained formulation"""
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    x, wt = _case(*shape, seed=sum(shape) + 9)
    dev = torch.device("cuda", 0)
    tx = torch.relu(torch.tensor(x, device=dev))                                   # a ReLU output: zeros and positives
    tw = torch.tensor(wt, device=dev)
    wp, wpt = ops.conv3d_k3_prep(tw), ops.conv3d_k3_prep(tw, transpose=True)
    g = torch.tensor(np.random.RandomState(3).randn(b, cout, d, h, w).astype(np.float32), device=dev)
    want = ops.relu_backward(ops.conv3d_k3(g, wpt, cin), tx)
    got = ops.conv3d_k3_masked(g, wpt, cin, tx)
    if cin <= 8:      # the backward of a layer with 8 input channels or fewer runs on the narrow vector-ALU kernel, which has no mask epilogue
        assert got is None
    else:
        assert got is not None and torch.equal(got, want)
    # chained pair of layers: y1 = relu(conv(x0)) consumed only by conv2
    x0 = torch.tensor(np.random.RandomState(4).randn(b, cin, d, h, w).astype(np.float32), device=dev, requires_grad=True)
    w1 = torch.tensor((np.random.RandomState(5).randn(cin, cin, 3, 3, 3) * 0.1).astype(np.float32), device=dev)
    p1, p1t = ops.conv3d_k3_prep(w1), ops.conv3d_k3_prep(w1, transpose=True)
    outs = []
    for chained in (False, True):
        y1 = ops.Conv3dK3.apply(x0, p1, p1t, cin, None, None, "consumer" if chained else True)
        y2 = ops.Conv3dK3.apply(y1, wp, wpt, cout, None, None, False, None, chained)
        (gx,) = torch.autograd.grad(y2, x0, g)
        outs.append(gx)
    assert torch.equal(outs[0], outs[1])


# ---- the Winograd route (csrc/wino2d.hip on 3x3x3 layers: transform in the (H, W) plane, depth taps inside the contraction) ----
def test_oracle_conv3d_wino_matches_torch():
    import torch.nn.functional as F
    from oracle import oracle_c as C
    rs = np.random.RandomState(5)
    for (b, cin, cout, d, h, w) in ((1, 5, 7, 3, 9, 11), (2, 8, 16, 4, 8, 32), (1, 3, 4, 1, 1, 1), (1, 6, 9, 2, 5, 2)):
        x = rs.randn(b, cin, d, h, w).astype(np.float32)
        wt = (rs.randn(cout, cin, 3, 3, 3) * (1.0 / (27 * cin)) ** 0.5).astype(np.float32)
        bias, res = rs.randn(cout).astype(np.float32), rs.randn(b, cout, d, h, w).astype(np.float32)
        ref = F.relu(F.conv3d(torch.tensor(x), torch.tensor(wt), torch.tensor(bias), padding=1) + torch.tensor(res)).numpy()
        assert np.abs(C.conv3d_wino(x, wt, bias, res, relu=True) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        g = rs.randn(b, cout, d, h, w).astype(np.float32)
        refg = torch.nn.grad.conv3d_input(x.shape, torch.tensor(wt), torch.tensor(g), padding=1).numpy()
        assert np.abs(C.conv3d_wino(g, wt, transpose=True) - refg).max() <= 1e-5 * max(1.0, np.abs(refg).max())
    x = rs.randn(1, 4, 2, 4, 6).astype(np.float32)
    wt = (rs.randn(5, 4, 3, 3, 3) * 0.1).astype(np.float32)
    bias, res, mask = rs.randn(5).astype(np.float32), rs.randn(1, 5, 2, 4, 6).astype(np.float32), rs.randn(1, 5, 2, 4, 6).astype(np.float32)
    want = np.maximum(C.conv3d_wino(x, wt) + bias[None, :, None, None, None] + res, np.float32(0)) * (mask > 0)
    assert C.conv3d_wino(x, wt, bias, res, mask, relu=True).tobytes() == (want + np.float32(0)).astype(np.float32).tobytes()


# (B, Cin, Cout, D, H, W): channels around the 4 / 8-channel stages and the 16 / 32 / 64-channel blocks, one plane (no depth neighbours),
# two planes (every plane at a border), odd heights and widths, more than one tile in every direction
WINO3 = [(1, 8, 32, 3, 8, 32), (2, 3, 32, 2, 19, 63), (1, 32, 32, 4, 9, 40), (1, 64, 64, 3, 13, 41), (1, 12, 70, 1, 9, 33), (1, 5, 7, 5, 3, 2),
         (1, 9, 9, 2, 1, 1), (1, 17, 40, 3, 17, 65)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", WINO3)
def test_hip_conv3d_wino_bit_exact_vs_oracle(shape):
    from oracle import oracle_c as C
    from eval_driving_safety_amd import ops
    b, cin, cout, d, h, w = shape
    rs = np.random.RandomState(sum(shape) + 9)
    x = rs.randn(b, cin, d, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3, 3) * (1.0 / (27 * cin)) ** 0.5).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    res, mask = rs.randn(b, cout, d, h, w).astype(np.float32), rs.randn(b, cout, d, h, w).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx, tw, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, wt, bias, res, mask))
    prep = ops.Conv3dWinoPrep(tw)
    want_plain, want_full = C.conv3d_wino(x, wt), C.conv3d_wino(x, wt, bias, res, mask, relu=True)
    for tile in (-1, 0, 1, 2, 3, 4, 5, 6, 7):
        assert ops.conv3d_wino(tx, prep, tile=tile).cpu().numpy().tobytes() == want_plain.tobytes(), tile
        assert ops.conv3d_wino(tx, prep, tb, tr, True, tm, tile=tile).cpu().numpy().tobytes() == want_full.tobytes(), tile
    ref = torch.nn.functional.conv3d(tx, tw, tb, padding=1)
    assert float((ops.conv3d_wino(tx, prep, tb) - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    g, gres = rs.randn(b, cout, d, h, w).astype(np.float32), rs.randn(b, cin, d, h, w).astype(np.float32)
    tg, tgr = torch.tensor(g, device=dev), torch.tensor(gres, device=dev)
    assert ops.conv3d_wino_dgrad(tg, prep).cpu().numpy().tobytes() == C.conv3d_wino(g, wt, transpose=True).tobytes()
    want_b = C.conv3d_wino(g, wt, residual=gres, mask=x, transpose=True)
    for tile in (0, 1, 2, 3, 4, 5, 6, 7):
        assert ops.conv3d_wino_dgrad(tg, prep, residual=tgr, mask=tx, tile=tile).cpu().numpy().tobytes() == want_b.tobytes(), tile
    refg = torch.nn.grad.conv3d_input(x.shape, tw, tg, padding=1)
    assert float((ops.conv3d_wino_dgrad(tg, prep) - refg).abs().max()) <= 1e-4 * max(1.0, float(refg.abs().max()))


@pytest.mark.gpu
def test_hip_conv3d_wino_layer_shapes_full_size_agree_with_the_direct_kernel():
    """the cost-volume layers at their real size: Winograd route vs the direct float32-MFMA kernel (different order of operations, same
    operator): within 1e-4 of each other, forward and backward"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(21)
    for cin, cout, dims in ((32, 32, (48, 96, 312)), (64, 32, (24, 96, 312)), (64, 64, (24, 48, 156)), (128, 128, (24, 10, 152))):
        x = torch.randn((1, cin) + dims, device=dev, generator=gen)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=gen) * (1.0 / (27 * cin)) ** 0.5
        prep = ops.Conv3dWinoPrep(wt)
        y, ref = ops.conv3d_wino(x, prep, relu=True), ops.conv3d_k3(x, ops.conv3d_k3_prep(wt), cout, relu=True)
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (cin, cout, dims)
        g = torch.randn_like(ref)
        gx, refg = ops.conv3d_wino_dgrad(g, prep), ops.conv3d_k3(g, ops.conv3d_k3_prep(wt, transpose=True), cin)
        assert float((gx - refg).abs().max()) <= 1e-4 * float(refg.abs().max()), (cin, cout, dims)
