"""csrc/wino4.hip - Winograd F(4x4,3x3) on the float32 matrix cores - against the oracle's restatement of its order of operations
(oracle/oracle.c orc_conv_wino4: bit for bit) and against torch's operator (1e-4 of the output's magnitude), 2D and 3x3x3 layers,
forward and the backward w.r.t. the input, both workgroup shapes, ragged sizes, every epilogue."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


def test_oracle_wino4_matches_torch():
    from oracle import oracle_c
    rs = np.random.RandomState(0)
    for shape, cout in (((2, 5, 9, 13), 6), ((1, 3, 4, 4), 2), ((1, 7, 17, 6), 3)):
        x = rs.randn(*shape).astype(np.float32)
        w = (rs.randn(cout, shape[1], 3, 3) * 0.2).astype(np.float32)
        bias = rs.randn(cout).astype(np.float32)
        ref = F.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bias), padding=1).numpy()
        assert np.abs(oracle_c.conv_wino4(x, w, bias) - ref).max() <= 1e-5 * np.abs(ref).max()
        g = rs.randn(shape[0], cout, shape[2], shape[3]).astype(np.float32)
        xr = torch.tensor(x, requires_grad=True)
        F.conv2d(xr, torch.tensor(w), padding=1).backward(torch.tensor(g))
        assert np.abs(oracle_c.conv_wino4(g, w, transpose=True) - xr.grad.numpy()).max() <= 1e-5 * np.abs(xr.grad.numpy()).max()
    x = rs.randn(1, 4, 5, 6, 9).astype(np.float32)
    w = (rs.randn(3, 4, 3, 3, 3) * 0.2).astype(np.float32)
    ref = F.relu(F.conv3d(torch.tensor(x), torch.tensor(w), padding=1)).numpy()
    assert np.abs(oracle_c.conv_wino4(x, w, relu=True) - ref).max() <= 1e-5 * np.abs(ref).max()
    g = rs.randn(1, 3, 5, 6, 9).astype(np.float32)
    xr = torch.tensor(x, requires_grad=True)
    F.conv3d(xr, torch.tensor(w), padding=1).backward(torch.tensor(g))
    assert np.abs(oracle_c.conv_wino4(g, w, transpose=True) - xr.grad.numpy()).max() <= 1e-5 * np.abs(xr.grad.numpy()).max()


CASES_2D = [  # b, cin, cout, h, w
    (1, 4, 64, 16, 32), (2, 8, 64, 16, 32), (1, 5, 7, 9, 13), (2, 12, 70, 21, 45), (1, 64, 64, 38, 125), (1, 16, 130, 5, 76), (3, 3, 3, 4, 4), (1, 32, 32, 33, 65),
]
CASES_3D = [  # b, cin, cout, d, h, w
    (1, 4, 8, 3, 10, 20), (1, 8, 64, 1, 16, 32), (2, 6, 5, 4, 7, 11), (1, 32, 32, 5, 20, 44),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES_2D)
def test_hip_conv2d_wino4_bit_exact_vs_oracle(case):
    from oracle import oracle_c
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    b, cin, cout, h, w = case
    rs = np.random.RandomState(cin * 7 + w)
    x = rs.randn(b, cin, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3) * 0.2).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = (rs.rand(b, cout, h, w) > 0.3).astype(np.float32)
    prep = ops.ConvWino4Prep(torch.tensor(wt, device=dev))
    tx, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, bias, res, mask))
    for tile in (0, 1, 2, 3, -1):
        y = ops.conv_wino4(tx, prep, tile=tile).cpu().numpy()
        want = oracle_c.conv_wino4(x, wt)
        assert y.tobytes() == want.tobytes(), (case, tile, float(np.abs(y - want).max()))
    y = ops.conv_wino4(tx, prep, tb, tr, True, tm).cpu().numpy()
    assert y.tobytes() == oracle_c.conv_wino4(x, wt, bias, res, mask, relu=True).tobytes()
    ref = F.conv2d(tx, torch.tensor(wt, device=dev), padding=1)
    assert float((ops.conv_wino4(tx, prep) - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gx = ops.conv_wino4_dgrad(torch.tensor(g, device=dev), prep).cpu().numpy()
    assert gx.tobytes() == oracle_c.conv_wino4(g, wt, transpose=True).tobytes()
    assert torch.equal(ops.conv_wino4(tx, prep, tb, tr, True, tm), ops.conv_wino4(tx, prep, tb, tr, True, tm))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES_3D)
def test_hip_conv3d_wino4_bit_exact_vs_oracle(case):
    from oracle import oracle_c
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    b, cin, cout, d, h, w = case
    rs = np.random.RandomState(cin * 5 + w)
    x = rs.randn(b, cin, d, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3, 3) * 0.15).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, d, h, w).astype(np.float32)
    prep = ops.ConvWino4Prep(torch.tensor(wt, device=dev))
    tx = torch.tensor(x, device=dev)
    for tile in (0, 1, 2, 3):
        y = ops.conv_wino4(tx, prep, torch.tensor(bias, device=dev), torch.tensor(res, device=dev), True, tile=tile).cpu().numpy()
        want = oracle_c.conv_wino4(x, wt, bias, res, relu=True)
        assert y.tobytes() == want.tobytes(), (case, tile, float(np.abs(y - want).max()))
    ref = F.conv3d(tx, torch.tensor(wt, device=dev), padding=1)
    assert float((ops.conv_wino4(tx, prep) - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    g = rs.randn(b, cout, d, h, w).astype(np.float32)
    gx = ops.conv_wino4_dgrad(torch.tensor(g, device=dev), prep).cpu().numpy()
    assert gx.tobytes() == oracle_c.conv_wino4(g, wt, transpose=True).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(5, 8, 70, 14, 14), (2, 8, 64, 7, 15), (1, 16, 64, 14, 14), (6, 32, 128, 20, 9)])
def test_hip_conv2d_wino4_image_pairs_bit_exact_vs_oracle(case):
    """tile 4: two images side by side in one 16 x 32 tile (maps of at most 15 columns - the RoI heads' 14 x 14 maps), odd batch sizes too;
    the same bytes as the one-image shapes and as the oracle"""
    from oracle import oracle_c
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    b, cin, cout, h, w = case
    rs = np.random.RandomState(cin * 3 + b)
    x = rs.randn(b, cin, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3) * 0.2).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = (rs.rand(b, cout, h, w) > 0.3).astype(np.float32)
    prep = ops.ConvWino4Prep(torch.tensor(wt, device=dev))
    tx, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, bias, res, mask))
    want = oracle_c.conv_wino4(x, wt, bias, res, mask, relu=True)
    for tile in (4, 0, -1):
        y = ops.conv_wino4(tx, prep, tb, tr, True, tm, tile=tile).cpu().numpy()
        assert y.tobytes() == want.tobytes(), (case, tile, float(np.abs(y - want).max()))
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gx = ops.conv_wino4_dgrad(torch.tensor(g, device=dev), prep, tile=4 if cout % 8 == 0 else -1).cpu().numpy()      # (pairs need a contraction of whole stages: 8 channels)
    assert gx.tobytes() == oracle_c.conv_wino4(g, wt, transpose=True).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 64, 19, 63), (1, 40, 70, 9, 13), (2, 20, 33, 21, 45), (1, 128, 32, 12, 30)])
def test_hip_conv2d_wino4_ksplit_bit_exact_vs_oracle(case):
    """the K-split launch (small maps: the contraction dealt to several workgroups per tile, the parts added in order by a second kernel):
    every tile shape and 2 / 3 / 4 parts against the oracle's restatement of that order, epilogue and backward included; one part too many
    for the stages (more parts than stages) degenerates to fewer parts, not to empty ones"""
    from oracle import oracle_c
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    b, cin, cout, h, w = case
    rs = np.random.RandomState(cin + 5 * w)
    x = rs.randn(b, cin, h, w).astype(np.float32)
    wt = (rs.randn(cout, cin, 3, 3) * 0.2).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    res = rs.randn(b, cout, h, w).astype(np.float32)
    mask = (rs.rand(b, cout, h, w) > 0.3).astype(np.float32)
    prep = ops.ConvWino4Prep(torch.tensor(wt, device=dev))
    tx, tb, tr, tm = (torch.tensor(a, device=dev) for a in (x, bias, res, mask))
    ref = F.conv2d(tx, torch.tensor(wt, device=dev), padding=1)
    for tile in (0, 1, 2, 3):
        for splits in (2, 3, 4, 40):
            chunk = ops.conv_wino4_ksplit_chunk(cin, tile, splits)
            assert chunk % (8 if tile < 2 else 4) == 0 and chunk * splits >= cin
            y = ops.conv_wino4(tx, prep, tile=tile, splits=splits)
            want = oracle_c.conv_wino4(x, wt, chunk=chunk)
            assert y.cpu().numpy().tobytes() == want.tobytes(), (case, tile, splits, chunk)
            assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        chunk = ops.conv_wino4_ksplit_chunk(cin, tile, 3)
        y = ops.conv_wino4(tx, prep, tb, tr, True, tm, tile=tile, splits=3).cpu().numpy()
        assert y.tobytes() == oracle_c.conv_wino4(x, wt, bias, res, mask, relu=True, chunk=chunk).tobytes(), (case, tile)
    g = rs.randn(b, cout, h, w).astype(np.float32)
    gres = rs.randn(b, cin, h, w).astype(np.float32)
    chunk = ops.conv_wino4_ksplit_chunk(cout, 0, 2)
    gx = ops.conv_wino4_dgrad(torch.tensor(g, device=dev), prep, residual=torch.tensor(gres, device=dev), tile=0, splits=2).cpu().numpy()
    assert gx.tobytes() == oracle_c.conv_wino4(g, wt, residual=gres, transpose=True, chunk=chunk).tobytes()
    # the library's own rule (splits=0) gives the bytes of SOME explicit number of parts with the tile it picks; and the same bytes twice
    a, b2 = ops.conv_wino4(tx, prep, tb, splits=0), ops.conv_wino4(tx, prep, tb, splits=0)
    assert torch.equal(a, b2) and float((a - ref - tb[None, :, None, None]).abs().max()) <= 1e-4 * float(ref.abs().max())
